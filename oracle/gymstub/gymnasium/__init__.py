"""Minimal stand-in for the `gymnasium` package (TEST INFRASTRUCTURE ONLY).

The reference (facebookresearch/Pearl, mounted read-only at /root/reference in
the build container) imports `gymnasium` at module scope in its spaces and
environment adapters (pearl/utils/instantiations/spaces/discrete.py:18-27,
box.py:19-28, pearl/user_envs/__init__.py:9-50).  gymnasium is not installed and
there is no network, so the golden-vector generator (oracle/make_golden.py) puts
this directory on sys.path to be able to import and *run* the real reference on
CPU.  Nothing in pearl_amd imports this.
"""
import numpy as np


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
        if isinstance(seed, np.random.Generator):
            self._np_random = seed
        else:
            self._np_random = np.random.default_rng(seed)

    @property
    def shape(self):
        return self._shape

    @property
    def np_random(self):
        return self._np_random

    def seed(self, seed=None):
        self._np_random = np.random.default_rng(seed)
        return [seed]

    def sample(self, mask=None):
        raise NotImplementedError

    def contains(self, x):
        raise NotImplementedError

    def __contains__(self, x):
        return self.contains(x)


class Env:
    metadata = {}
    observation_space = None
    action_space = None

    def reset(self, *, seed=None, options=None):
        raise NotImplementedError

    def step(self, action):
        raise NotImplementedError

    def close(self):
        pass

    @property
    def unwrapped(self):
        return self


class Wrapper(Env):
    def __init__(self, env):
        self.env = env

    def __getattr__(self, name):
        return getattr(self.env, name)

    def reset(self, **kw):
        return self.env.reset(**kw)

    def step(self, action):
        return self.env.step(action)

    @property
    def unwrapped(self):
        return self.env.unwrapped


class ObservationWrapper(Wrapper):
    pass


class RewardWrapper(Wrapper):
    pass


class ActionWrapper(Wrapper):
    pass


def make(*args, **kwargs):
    raise RuntimeError("gymnasium stub: no environments are available")


from . import spaces, envs  # noqa: E402,F401
