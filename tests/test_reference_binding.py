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


# ---------------------------------------------------------------------------------------------
# PPO and ContinuousSoftActorCritic behind the reference's ABCs (VERDICT r5 missing-2)
# ---------------------------------------------------------------------------------------------
def _stand_alone(ref_learner):
    """What PearlAgent.__init__ would give the reference's learner (pearl_agent.py:95), without
    the agent: on the GPU box the agent would move it to cuda:0 (utils/device.py:48-59), and the
    comparison wants the reference's own CPU arithmetic."""
    from pearl.history_summarization_modules.identity_history_summarization_module import (
        IdentityHistorySummarizationModule)
    from pearl.safety_modules.identity_safety_module import IdentitySafetyModule
    ref_learner.safety_module = IdentitySafetyModule()
    ref_learner.set_history_summarization_module(IdentityHistorySummarizationModule())


def _ppo_pair(S=6, A=4, hidden=(24, 24), rounds=5, batch=32, eps=0.1):
    from pearl.action_representation_modules.one_hot_action_representation_module import (
        OneHotActionTensorRepresentationModule)
    from pearl.policy_learners.sequential_decision_making.ppo import (PPOReplayBuffer,
                                                                       ProximalPolicyOptimization)
    from pearl.utils.instantiations.spaces.discrete_action import DiscreteActionSpace

    from reference_binding import HipPPO, HipPPOReplayBuffer
    space = DiscreteActionSpace([torch.tensor([k]) for k in range(A)])
    torch.manual_seed(4)
    ref = ProximalPolicyOptimization(
        state_dim=S, action_space=space, actor_hidden_dims=list(hidden), critic_hidden_dims=list(hidden),
        training_rounds=rounds, batch_size=batch, epsilon=eps,
        action_representation_module=OneHotActionTensorRepresentationModule(A))
    hip = HipPPO(state_dim=S, action_space=space, actor_hidden_dims=list(hidden),
                 critic_hidden_dims=list(hidden), training_rounds=rounds, batch_size=batch, epsilon=eps,
                 action_representation_module=OneHotActionTensorRepresentationModule(A))
    hip.impl._actor.load_state_dict(ref._actor.state_dict())
    hip.impl._critic.load_state_dict(ref._critic.state_dict())
    _stand_alone(ref)
    return ref, hip, space, PPOReplayBuffer, HipPPOReplayBuffer


@pytest.mark.gpu
def test_real_pearl_agent_drives_hip_ppo_on_policy(reference):
    """The reference's unmodified PearlAgent with HipPPO + HipPPOReplayBuffer: observe -> learn
    (rollout pass + pa_ppo_learn; then the agent's own on-policy `replay_buffer.clear()`,
    pearl_agent.py:217-218) -> act with and without exploitation."""
    from pearl.api.action_result import ActionResult
    from pearl.pearl_agent import PearlAgent
    from pearl.policy_learners.policy_learner import PolicyLearner
    from pearl.replay_buffers.replay_buffer import ReplayBuffer
    S, A = 6, 4
    _, pl, space, _, HipPPOReplayBuffer = _ppo_pair(S, A, rounds=3, batch=16)
    rb = HipPPOReplayBuffer(128)
    assert isinstance(pl, PolicyLearner) and isinstance(rb, ReplayBuffer) and pl.on_policy
    agent = PearlAgent(policy_learner=pl, replay_buffer=rb, device_id=0)
    agent.reset(torch.zeros(S), space)
    gen = torch.Generator().manual_seed(0)
    for i in range(50):
        agent._latest_action = torch.tensor([i % A])
        agent.observe(ActionResult(observation=torch.randn(S, generator=gen), reward=float(i % 3) - 1.0,
                                   terminated=(i % 13 == 12), truncated=(i % 17 == 16),
                                   available_action_space=space))
    assert len(rb) == 50
    before = {k: v.clone() for k, v in pl.impl._actor.state_dict().items()}
    random.seed(0)
    report = agent.learn()
    assert set(report) == {"actor_loss", "critic_loss"}          # the reference's report keys
    assert all(len(v) == 3 and all(x == x for x in v) for v in report.values())
    assert pl._training_steps == 3 and len(rb) == 0              # on-policy: cleared by the agent
    assert any(not torch.equal(before[k], v) for k, v in pl.impl._actor.state_dict().items())
    assert pl.impl._flat, "the HIP engine never bound the networks"
    acts = {int(agent.act(exploit=False)) for _ in range(30)}
    assert acts <= set(range(A)) and int(agent.act(exploit=True)) in range(A)


@pytest.mark.gpu
def test_hip_ppo_tracks_the_reference_ppo_on_identical_data(reference):
    """One learn() of the reference's own ProximalPolicyOptimization (CPU) and of HipPPO: same
    initial networks, same rollout, same `random.sample` stream — GAE / lambda-returns / old action
    probabilities, the per-round reports and the parameters agree."""
    ref, hip, space, PPOReplayBuffer, HipPPOReplayBuffer = _ppo_pair()
    S, A, N_ = 6, 4, 96
    hip.impl.to(torch.device("cuda", 0))
    rb_ref, rb_hip = PPOReplayBuffer(N_), HipPPOReplayBuffer(N_)
    rb_hip.device_for_batches = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(2)
    st = torch.randn(N_ + 1, S, generator=gen)
    rw = torch.randn(N_, generator=gen)
    for i in range(N_):
        kw = dict(state=st[i], action=torch.tensor([(i * 7) % A]), reward=float(rw[i]),
                  terminated=(i % 19 == 18), truncated=(i % 23 == 11), curr_available_actions=space,
                  next_state=st[i + 1], next_available_actions=space, max_number_actions=A)
        rb_ref.push(**kw)
        rb_hip.push(**kw)
    random.seed(21)
    want = ref.learn(rb_ref)
    random.seed(21)
    got = hip.learn(rb_hip)
    extra = rb_hip.impl.extra
    torch.testing.assert_close(extra["gae"].cpu(), torch.cat([t.gae for t in rb_ref.memory]).view(-1),
                               rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(extra["action_probs"].cpu(),
                               torch.cat([t.action_probs for t in rb_ref.memory]).view(-1), rtol=1e-5, atol=1e-7)
    for k in ("actor_loss", "critic_loss"):
        torch.testing.assert_close(torch.tensor(got[k]), torch.tensor([float(x.detach()) if torch.is_tensor(x) else float(x) for x in want[k]]),
                                   rtol=2e-4, atol=2e-4, msg=k)
    for name in ("_actor", "_critic"):
        for k, v in getattr(ref, name).state_dict().items():
            torch.testing.assert_close(getattr(hip.impl, name).state_dict()[k].cpu(), v, rtol=1e-3,
                                       atol=2e-5, msg=f"{name}.{k}")


def _sac_pair(S=7, A=3, hidden=(32, 32), rounds=4, batch=24):
    from pearl.policy_learners.sequential_decision_making.soft_actor_critic_continuous import (
        ContinuousSoftActorCritic)
    from pearl.utils.instantiations.spaces.box_action import BoxActionSpace

    from reference_binding import HipContinuousSAC
    space = BoxActionSpace(low=-torch.ones(A) * torch.tensor([1.0, 2.0, 0.5]),
                           high=torch.ones(A) * torch.tensor([1.5, 1.0, 0.5]))
    torch.manual_seed(9)
    ref = ContinuousSoftActorCritic(state_dim=S, action_space=space, actor_hidden_dims=list(hidden),
                                    critic_hidden_dims=list(hidden), training_rounds=rounds,
                                    batch_size=batch)
    hip = HipContinuousSAC(state_dim=S, action_space=space, actor_hidden_dims=list(hidden),
                           critic_hidden_dims=list(hidden), training_rounds=rounds, batch_size=batch)
    hip.impl._actor.load_state_dict(ref._actor.state_dict())
    hip.impl._critic.load_state_dict(ref._critic.state_dict())
    hip.impl._critic_target.load_state_dict(ref._critic_target.state_dict())
    _stand_alone(ref)
    return ref, hip, space


@pytest.mark.gpu
def test_real_pearl_agent_drives_hip_continuous_sac(reference):
    """The reference's unmodified PearlAgent with HipContinuousSAC + HipReplayBuffer: continuous
    actions through observe (no max_number_actions, pearl_agent.py:199-203), learn() = pa_sac_learn,
    act inside the action box."""
    from pearl.api.action_result import ActionResult
    from pearl.pearl_agent import PearlAgent

    from reference_binding import HipReplayBuffer
    S, A = 7, 3
    _, pl, space = _sac_pair(S, A)
    rb = HipReplayBuffer(256, sampler="device")
    agent = PearlAgent(policy_learner=pl, replay_buffer=rb, device_id=0)
    assert rb._is_action_continuous is True and not pl.on_policy
    agent.reset(torch.zeros(S), space)
    gen = torch.Generator().manual_seed(5)
    for i in range(60):
        agent._latest_action = space.low + (space.high - space.low) * torch.rand(A, generator=gen)
        agent.observe(ActionResult(observation=torch.randn(S, generator=gen), reward=float(i % 4),
                                   terminated=(i % 11 == 10), truncated=False,
                                   available_action_space=space))
    assert len(rb) == 60
    before = {k: v.clone() for k, v in pl.impl._critic.state_dict().items()}
    random.seed(3)
    torch.manual_seed(3)
    report = agent.learn()
    assert set(report) == {"actor_loss", "critic_loss", "entropy_coef"}
    assert all(len(v) == 4 and all(x == x for x in v) for v in report.values())
    assert pl._training_steps == 4 and len(rb) == 60             # off-policy: the buffer stays
    assert any(not torch.equal(before[k], v) for k, v in pl.impl._critic.state_dict().items())
    a = torch.as_tensor(agent.act(exploit=True)).cpu().view(-1)
    assert a.shape == (A,) and bool((a >= space.low - 1e-6).all()) and bool((a <= space.high + 1e-6).all())


@pytest.mark.gpu
def test_hip_sac_tracks_the_reference_sac_on_identical_data_and_noise(reference):
    """One learn() of the reference's own ContinuousSoftActorCritic (CPU; its reparameterisation
    noise comes from torch's seeded global generator) and of HipContinuousSAC fed the same draws:
    reports and parameters agree."""
    from pearl.replay_buffers.basic_replay_buffer import BasicReplayBuffer

    from reference_binding import HipReplayBuffer
    S, A, N_, B, R = 7, 3, 120, 24, 4
    ref, hip, space = _sac_pair(S, A, rounds=R, batch=B)
    hip.impl.to(torch.device("cuda", 0))
    rb_ref, rb_hip = BasicReplayBuffer(N_), HipReplayBuffer(N_)
    for rb in (rb_ref, rb_hip):
        rb._is_action_continuous = True
    rb_hip.device_for_batches = torch.device("cuda", 0)
    gen = torch.Generator().manual_seed(6)
    st = torch.randn(N_ + 1, S, generator=gen)
    for i in range(N_):
        kw = dict(state=st[i], action=space.low + (space.high - space.low) * torch.rand(A, generator=gen),
                  reward=float(torch.randn((), generator=gen)), terminated=(i % 9 == 8), truncated=False,
                  curr_available_actions=space, next_state=st[i + 1], next_available_actions=space)
        rb_ref.push(**kw)
        rb_hip.push(**kw)
    torch.manual_seed(77)
    noise = torch.stack([torch.stack([torch.normal(torch.zeros(B, A), torch.ones(B, A)) for _ in range(2)])
                         for _ in range(R)])

    class Replay:
        calls = 0

        def __call__(self, b, a, dev):
            r, j = divmod(self.calls, 2)
            Replay.calls += 1
            return noise[r, j]

        def rounds(self, first, n, b, a, dev):
            Replay.calls += 2 * n
            return noise[first:first + n].to(dev)

    hip.impl.noise_source = Replay()
    random.seed(31)
    torch.manual_seed(77)
    want = ref.learn(rb_ref)
    random.seed(31)
    got = hip.learn(rb_hip)
    assert Replay.calls == 2 * R
    for k in ("actor_loss", "critic_loss", "entropy_coef"):
        torch.testing.assert_close(torch.tensor(got[k]), torch.tensor([float(x.detach()) if torch.is_tensor(x) else float(x) for x in want[k]]),
                                   rtol=5e-4, atol=5e-5, msg=k)
    for name in ("_actor", "_critic", "_critic_target"):
        for k, v in getattr(ref, name).state_dict().items():
            torch.testing.assert_close(getattr(hip.impl, name).state_dict()[k].cpu(), v, rtol=2e-3,
                                       atol=3e-5, msg=f"{name}.{k}")
