import numpy as np

from .. import Space


class Discrete(Space):
    def __init__(self, n, seed=None, start=0):
        self.n = int(n)
        self.start = int(start)
        super().__init__((), np.int64, seed)

    def sample(self, mask=None):
        if mask is not None:
            valid = np.flatnonzero(np.asarray(mask) == 1)
            if len(valid) > 0:
                return int(self.start + self._np_random.choice(valid))
            return self.start
        return int(self.start + self._np_random.integers(self.n))

    def contains(self, x):
        try:
            x = int(x)
        except Exception:
            return False
        return self.start <= x < self.start + self.n


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        low = np.asarray(low, dtype=dtype)
        high = np.asarray(high, dtype=dtype)
        if shape is None:
            shape = np.broadcast(low, high).shape
        self.low = np.broadcast_to(low, shape).astype(dtype).copy()
        self.high = np.broadcast_to(high, shape).astype(dtype).copy()
        super().__init__(shape, dtype, seed)

    def sample(self, mask=None):
        u = self._np_random.uniform(size=self._shape)
        return (self.low + u * (self.high - self.low)).astype(self.dtype)

    def contains(self, x):
        x = np.asarray(x)
        return x.shape == self._shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        self.spaces = dict(spaces or {}, **kw)
        super().__init__(None, None, seed)


class Tuple(Space):
    def __init__(self, spaces=(), seed=None):
        self.spaces = tuple(spaces)
        super().__init__(None, None, seed)


class MultiDiscrete(Space):
    def __init__(self, nvec, dtype=np.int64, seed=None):
        self.nvec = np.asarray(nvec, dtype=dtype)
        super().__init__(self.nvec.shape, dtype, seed)


class MultiBinary(Space):
    def __init__(self, n, seed=None):
        self.n = n
        super().__init__((n,) if np.isscalar(n) else tuple(n), np.int8, seed)
