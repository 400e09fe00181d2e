"""GPU: the opt-in performance fields of learn()'s report (second half;
pearl_amd/policy_learners/policy_learner.py::perf_reported).  Off by default: the report then has
exactly the reference's keys (policy_learner.py:181-195 aggregates whatever learn_batch returns:
{"loss"} for DeepQLearning, {"actor_loss", "critic_loss"} for PPO, + "entropy_coef" for SAC) and
the call is the untouched loop.  On: the same losses, bit for bit, plus perf/* one-element lists."""
import random

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dqn(rounds=25, B=256, n=4096, S=128, A=16):
    from pearl_amd import (BasicReplayBuffer, DeepQLearning, DiscreteActionSpace,
                           OneHotActionTensorRepresentationModule, PearlAgent)
    sp = DiscreteActionSpace([torch.tensor([k]) for k in range(A)])
    torch.manual_seed(0)
    pl = DeepQLearning(state_dim=S, action_space=sp, hidden_dims=[256, 256], training_rounds=rounds,
                       batch_size=B, action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = BasicReplayBuffer(n, sampler="device")
    PearlAgent(pl, replay_buffer=rb, device_id=0)
    g = torch.Generator(device=DEV).manual_seed(1)
    st = torch.randn(n + 1, S, device=DEV, generator=g)
    ids = torch.arange(n, device=DEV)
    rb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 7).float(),
                 terminated=(ids % 50 == 0), truncated=torch.zeros(n, dtype=torch.bool, device=DEV),
                 next_state=st[1:], curr_available_actions=sp, next_available_actions=sp,
                 max_number_actions=A)
    return pl, rb


def test_dqn_report_is_the_references_by_default_and_gains_perf_fields_on_request():
    runs = {}
    for on in (False, True):
        pl, rb = _dqn()
        assert pl.performance_report is False
        pl.performance_report = on
        out = []
        for call in range(2):          # (the first call binds the library handle; the second is timed)
            random.seed(5 + call)
            out.append(pl.learn(rb))
        runs[on] = out
    for rep in runs[False]:
        assert set(rep) == {"loss"} and len(rep["loss"]) == 25
    assert runs[True][0]["loss"] == runs[False][0]["loss"] and runs[True][1]["loss"] == runs[False][1]["loss"]
    rep = runs[True][1]
    perf = {k: v for k, v in rep.items() if k.startswith("perf/")}
    assert set(rep) - set(perf) == {"loss"}
    assert all(isinstance(v, list) and len(v) == 1 for v in perf.values())
    assert perf["perf/rounds"] == [25]
    tps, wall = perf["perf/transitions_per_s"][0], perf["perf/learn_wall_us"][0]
    # (a consistency check, not a speed claim: a cold box has taken 84 ms for this call — lazy
    #  allocations of the first timed learn() — where a warm one takes 1.5 ms)
    assert abs(tps - 256 * 25 / (wall * 1e-6)) <= 1e-6 * tps and 1e3 < tps < 1e9
    stages = {k.rsplit("/", 1)[1]: v[0] for k, v in perf.items() if k.startswith("perf/kernel_us/")}
    assert {"target_pass", "row_pass", "weight_grad_adamw"} <= set(stages), stages
    assert all(0.5 < v < 5000 for v in stages.values()), stages
    print("\nDQN perf report:", {k: round(v[0], 2) for k, v in perf.items()})


def test_sac_and_ppo_reports_gain_perf_fields_on_request():
    from pearl_amd import (BasicReplayBuffer, BoxActionSpace, ContinuousSoftActorCritic,
                           DiscreteActionSpace, OneHotActionTensorRepresentationModule, PearlAgent,
                           PPOReplayBuffer, ProximalPolicyOptimization)
    n, S, A = 4096, 32, 4
    torch.manual_seed(0)
    st = torch.randn(n + 1, S, device=DEV)
    ids = torch.arange(n, device=DEV)
    # SAC
    pl = ContinuousSoftActorCritic(action_space=BoxActionSpace(-torch.ones(A), torch.ones(A)), state_dim=S,
                                   actor_hidden_dims=[64, 64], critic_hidden_dims=[64, 64], batch_size=256,
                                   training_rounds=8)
    rb = BasicReplayBuffer(n, sampler="device")
    PearlAgent(pl, replay_buffer=rb, device_id=0)
    rb.push_many(state=st[:-1], action=torch.rand(n, A, device=DEV) * 2 - 1, reward=(ids % 5).float(),
                 terminated=(ids % 11 == 0), truncated=torch.zeros(n, dtype=torch.bool, device=DEV),
                 next_state=st[1:])
    random.seed(1)
    assert set(pl.learn(rb)) == {"actor_loss", "critic_loss", "entropy_coef"}
    pl.performance_report = True
    rep = pl.learn(rb)
    assert {"actor_loss", "critic_loss", "entropy_coef", "perf/transitions_per_s", "perf/learn_wall_us",
            "perf/rounds"} <= set(rep)
    assert any(k.startswith("perf/kernel_us/") for k in rep), sorted(rep)
    print("\nSAC perf report:", {k: round(v[0], 2) for k, v in rep.items() if k.startswith("perf/")})
    # PPO
    sp = DiscreteActionSpace([torch.tensor([k]) for k in range(A)])
    ppo = ProximalPolicyOptimization(action_space=sp, state_dim=S, actor_hidden_dims=[64, 64],
                                     critic_hidden_dims=[64, 64], training_rounds=6, batch_size=512,
                                     epsilon=0.1, action_representation_module=OneHotActionTensorRepresentationModule(A))
    prb = PPOReplayBuffer(n, sampler="device")
    PearlAgent(ppo, replay_buffer=prb, device_id=0)
    prb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 5).float(),
                  terminated=(ids % 50 == 49), truncated=torch.zeros(n, dtype=torch.bool, device=DEV),
                  next_state=st[1:], curr_available_actions=sp, next_available_actions=sp, max_number_actions=A)
    random.seed(2)
    assert set(ppo.learn(prb)) == {"actor_loss", "critic_loss"}
    ppo.performance_report = True
    rep = ppo.learn(prb)
    assert {"actor_loss", "critic_loss", "perf/transitions_per_s", "perf/rounds"} <= set(rep)
    assert rep["perf/rounds"] == [6]
    print("\nPPO perf report:", {k: round(v[0], 2) for k, v in rep.items() if k.startswith("perf/")})
