"""the INTEGRATION.md §2 binding, executed.  Needs the reference itself
(/root/reference + oracle/gymstub), so it runs in the build container and is skipped on the GPU
box; without a HIP device the hot path must fail loudly (no CPU fallback), with one the real
PearlAgent learns through libpearl_amd."""
import os
import random
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PEARL_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pearl")),
                                reason="the reference (facebookresearch/Pearl) is not on this box")


@pytest.fixture()
def reference():
    added = [os.path.join(REPO, "oracle", "gymstub"), REF]
    sys.path[:0] = added
    yield
    for p in added:
        sys.path.remove(p)


def test_pearl_amd_components_under_the_real_pearl_agent(reference):
    from pearl.action_representation_modules.one_hot_action_representation_module import (
        OneHotActionTensorRepresentationModule)
    from pearl.api.action_result import ActionResult
    from pearl.pearl_agent import PearlAgent
    from pearl.policy_learners.policy_learner import PolicyLearner
    from pearl.replay_buffers.replay_buffer import ReplayBuffer
    from pearl.utils.instantiations.spaces.discrete_action import DiscreteActionSpace

    from pearl_amd import _native as N
    from reference_binding import HipDeepQLearning, HipReplayBuffer

    A, S = 3, 4
    space = DiscreteActionSpace([torch.tensor([k]) for k in range(A)])
    pl = HipDeepQLearning(state_dim=S, action_space=space, hidden_dims=[16, 16], training_rounds=4,
                          batch_size=8, action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = HipReplayBuffer(64)
    assert isinstance(pl, PolicyLearner) and isinstance(rb, ReplayBuffer)      # the reference's ABCs
    has_gpu = torch.cuda.is_available()
    agent = PearlAgent(policy_learner=pl, replay_buffer=rb, device_id=0 if has_gpu else -1)
    assert agent.replay_buffer is rb and rb._is_action_continuous is False
    agent.reset(torch.zeros(S), space)
    agent._latest_action = torch.tensor([1])
    result = ActionResult(observation=torch.ones(S), reward=1.0, terminated=False, truncated=False,
                          available_action_space=space)
    if not has_gpu:
        # no HIP device: the arena refuses, loudly — the real agent never falls back to a CPU path
        with pytest.raises(N.NativeError, match="no HIP device|no CPU"):
            agent.observe(result)
        assert len(rb) == 0
        return
    gen = torch.Generator().manual_seed(0)
    for i in range(40):
        agent._latest_action = torch.tensor([i % A])
        agent.observe(ActionResult(observation=torch.randn(S, generator=gen), reward=float(i % 3),
                                   terminated=(i % 10 == 9), truncated=False,
                                   available_action_space=space))
    assert len(rb) == 40
    random.seed(0)
    report = agent.learn()
    assert len(report["loss"]) == 4 and all(x == x for x in report["loss"])
    batch = rb.sample(8)
    assert type(batch).__module__.startswith("pearl.")       # the reference's TransitionBatch
    assert agent.act(exploit=True) is not None
