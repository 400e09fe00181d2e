"""The INTEGRATION.md §2 binding, executed: the REAL ``pearl.pearl_agent.PearlAgent`` driving the
HIP replay arena and the HIP DQN learner through tests/reference_binding.py.

The reference comes from /root/reference in the build container and from oracle/_ref/ (staged by
oracle/stage_ref.sh, git-ignored, travels with the working tree) on the GPU box.

* ``-m "not gpu"``: without a HIP device the hot path fails loudly — the real agent never falls
  back to a CPU path;
* ``-m gpu``: observe -> learn -> act of the real agent on libpearl_amd.so,
  and the same learn() against the reference's OWN DeepQLearning on identical data.
"""
import os
import random
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reference_root():
    for cand in (os.environ.get("PEARL_REFERENCE"), "/root/reference",
                 os.path.join(REPO, "oracle", "_ref")):
        if cand and os.path.isdir(os.path.join(cand, "pearl")):
            return cand
    return None


REF = _reference_root()
pytestmark = pytest.mark.skipif(REF is None, reason="the reference (facebookresearch/Pearl) is "
                                "neither at /root/reference nor staged under oracle/_ref")


@pytest.fixture()
def reference():
    added = [os.path.join(REPO, "oracle", "gymstub"), REF]
    sys.path[:0] = added
    yield
    for p in added:
        sys.path.remove(p)


def _agent(A=3, S=4, rounds=4, batch=8, cap=64):
    from pearl.action_representation_modules.one_hot_action_representation_module import (
        OneHotActionTensorRepresentationModule)
    from pearl.pearl_agent import PearlAgent
    from pearl.policy_learners.policy_learner import PolicyLearner
    from pearl.replay_buffers.replay_buffer import ReplayBuffer
    from pearl.utils.instantiations.spaces.discrete_action import DiscreteActionSpace

    from reference_binding import HipDeepQLearning, HipReplayBuffer

    space = DiscreteActionSpace([torch.tensor([k]) for k in range(A)])
    pl = HipDeepQLearning(state_dim=S, action_space=space, hidden_dims=[16, 16],
                          training_rounds=rounds, batch_size=batch,
                          action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = HipReplayBuffer(cap)
    assert isinstance(pl, PolicyLearner) and isinstance(rb, ReplayBuffer)      # the reference's ABCs
    has_gpu = torch.cuda.is_available()
    agent = PearlAgent(policy_learner=pl, replay_buffer=rb, device_id=0 if has_gpu else -1)
    assert agent.replay_buffer is rb and rb._is_action_continuous is False
    return agent, pl, rb, space


def test_real_pearl_agent_refuses_without_a_device(reference):
    if torch.cuda.is_available():
        pytest.skip("a HIP device is visible: the refusal branch is the CPU container's")
    from pearl.api.action_result import ActionResult

    from pearl_amd import _native as N
    agent, pl, rb, space = _agent()
    agent.reset(torch.zeros(4), space)
    agent._latest_action = torch.tensor([1])
    result = ActionResult(observation=torch.ones(4), reward=1.0, terminated=False, truncated=False,
                          available_action_space=space)
    # no HIP device: the arena refuses, loudly — the real agent never falls back to a CPU path
    with pytest.raises(N.NativeError, match="no HIP device|no CPU"):
        agent.observe(result)
    assert len(rb) == 0


@pytest.mark.gpu
def test_real_pearl_agent_observes_learns_and_acts_on_libpearl_amd(reference):
    """pearl/pearl_agent.py:169-231 (observe / learn / act) of the reference, unmodified, with the
    two HIP-backed components plugged in through the reference's own ABCs."""
    from pearl.api.action_result import ActionResult

    from pearl_amd import _native as N
    A, S = 3, 4
    agent, pl, rb, space = _agent(A, S)
    agent.reset(torch.zeros(S), space)
    gen = torch.Generator().manual_seed(0)
    for i in range(40):
        agent._latest_action = torch.tensor([i % A])
        agent.observe(ActionResult(observation=torch.randn(S, generator=gen), reward=float(i % 3),
                                   terminated=(i % 10 == 9), truncated=False,
                                   available_action_space=space))
    assert len(rb) == 40
    before = {k: v.clone() for k, v in pl.impl._Q.state_dict().items()}
    random.seed(0)
    report = agent.learn()
    assert len(report["loss"]) == 4 and all(x == x for x in report["loss"])
    assert pl._training_steps == 4
    assert any(not torch.equal(before[k], v) for k, v in pl.impl._Q.state_dict().items())
    # the arithmetic ran in the HIP library: its handle exists and is bound to these parameters
    assert pl.impl._native.handle is not None
    N.check(N.lib().pa_dqn_check(pl.impl._native.handle))
    batch = rb.sample(8)
    assert type(batch).__module__.startswith("pearl.")       # the reference's TransitionBatch
    assert batch.state.is_cuda and batch.state.shape == (8, S)
    assert agent.act(exploit=True) is not None


@pytest.mark.gpu
def test_hip_learner_tracks_the_reference_learner_on_identical_data(reference):
    """The reference's OWN DeepQLearning (CPU, torch autograd + AdamW) and the HIP learner, same
    initial parameters, same transitions, same `random.sample` index stream: the per-round
    mean |Q - target| reports and the parameters after a learn() call agree."""
    from pearl.action_representation_modules.one_hot_action_representation_module import (
        OneHotActionTensorRepresentationModule)
    from pearl.policy_learners.sequential_decision_making.deep_q_learning import DeepQLearning
    from pearl.replay_buffers.basic_replay_buffer import BasicReplayBuffer
    from pearl.utils.instantiations.spaces.discrete_action import DiscreteActionSpace

    from reference_binding import HipDeepQLearning, HipReplayBuffer

    A, S, N_, B, R = 4, 12, 300, 32, 12
    space = DiscreteActionSpace([torch.tensor([k]) for k in range(A)])
    torch.manual_seed(3)
    ref = DeepQLearning(state_dim=S, action_space=space, hidden_dims=[32, 32], training_rounds=R,
                        batch_size=B, action_representation_module=OneHotActionTensorRepresentationModule(A))
    hip = HipDeepQLearning(state_dim=S, action_space=space, hidden_dims=[32, 32], training_rounds=R,
                           batch_size=B, action_representation_module=OneHotActionTensorRepresentationModule(A))
    hip.impl._Q.load_state_dict(ref._Q.state_dict())
    hip.impl._Q_target.load_state_dict(ref._Q_target.state_dict())
    hip.impl.to(torch.device("cuda", 0))
    rb_ref, rb_hip = BasicReplayBuffer(N_), HipReplayBuffer(N_)
    rb_hip.device_for_batches = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(1)
    st = torch.randn(N_ + 1, S, generator=gen)
    for i in range(N_):
        kw = dict(state=st[i], action=torch.tensor([i % A]), reward=float(i % 5),
                  terminated=(i % 17 == 0), truncated=False, curr_available_actions=space,
                  next_state=st[i + 1], next_available_actions=space, max_number_actions=A)
        rb_ref.push(**kw)
        rb_hip.push(**kw)
    random.seed(11)
    want = ref.learn(rb_ref)["loss"]
    random.seed(11)
    got = hip.learn(rb_hip)["loss"]
    torch.testing.assert_close(torch.tensor(got), torch.tensor(want), rtol=2e-4, atol=1e-5)
    for k, v in ref._Q.state_dict().items():
        torch.testing.assert_close(hip.impl._Q.state_dict()[k].cpu(), v, rtol=1e-3, atol=2e-5, msg=k)
    for k, v in ref._Q_target.state_dict().items():
        torch.testing.assert_close(hip.impl._Q_target.state_dict()[k].cpu(), v, rtol=1e-3, atol=2e-5,
                                   msg=k)
