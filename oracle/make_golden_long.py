#!/usr/bin/env python3
"""Multi-round, full-size golden runs of the REAL reference (/root/reference, CPU) for BASELINE
configs 3 / 4 / 5, each with its own yardstick (VERDICT r5 "next" 1a / 1b).

TEST INFRASTRUCTURE ONLY (build container; the reference does not travel to the GPU box):

    python oracle/make_golden_long.py [sac] [ppo] [ppo_eps0] [bandit]

Every fixture holds TWO runs of the reference from identical initial parameters on identical
transitions and index lists:

  ref    `learn(replay_buffer)` as the reference runs it
         (policy_learner.py:162-195 -> soft_actor_critic_continuous.py:131-231 / ppo.py:152-293 /
         neural_linear_bandit.py:159-225);
  twin   the same call with the rows of every minibatch in a different ORDER (and, for SAC, the
         reparameterisation noise permuted with them): identical mathematics, different fp32
         summation order inside MKL.  Two fp32 implementations of one AdamW trajectory drift apart
         chaotically; the twin measures how far the reference drifts from ITSELF, which is the bar
         the HIP loops are held to (tests/test_gpu_long_runs.py), block of rounds by block of rounds.

  fp64   the same call with the reference's own modules converted to float64 (`.double()`, the
         batches cast up by the history-summarisation hook, the SAME float32 initial parameters,
         index lists and noise): the trajectory both fp32 implementations approximate.  The bar for
         the HIP loops is "as close to fp64 as the reference's fp32 run is, within a small factor".

Stored: the initial parameters, the index lists, (SAC) the noise of every round, the per-round
reports of all three runs, the final parameters of `ref`, `twin_divergence` — per-tensor statistics
of twin vs ref (oracle/fixture_inputs.py::divergence) instead of a second parameter snapshot — and
`fp64_minus_ref`, the float64 run's final parameters as float32 DIFFERENCES from `ref`'s (the
differences are ~1e-7: stored this way they lose nothing that matters).
The transitions are regenerated from `input_seed` where the fixture is used (fixture_inputs.py).
"""
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("PEARL_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "gymstub"))
sys.path.insert(0, REF)

import torch  # noqa: E402

import fixture_inputs as FI  # noqa: E402

import pearl.replay_buffers.tensor_based_replay_buffer as TBRB  # noqa: E402
import torch.distributions.normal as TDN  # noqa: E402
from pearl.action_representation_modules.one_hot_action_representation_module import (  # noqa: E402
    OneHotActionTensorRepresentationModule,
)
from pearl.pearl_agent import PearlAgent  # noqa: E402
from pearl.policy_learners.sequential_decision_making.ppo import (  # noqa: E402
    PPOReplayBuffer,
    ProximalPolicyOptimization,
)
from pearl.policy_learners.sequential_decision_making.soft_actor_critic_continuous import (  # noqa: E402
    ContinuousSoftActorCritic,
)
from pearl.replay_buffers import BasicReplayBuffer  # noqa: E402
from pearl.replay_buffers.transition import TransitionBatch  # noqa: E402
from pearl.utils.instantiations.spaces.box_action import BoxActionSpace  # noqa: E402
from pearl.utils.instantiations.spaces.discrete_action import DiscreteActionSpace  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")

SAC_LONG = {
    # BASELINE config 3 at its own batch size, 20 rounds of learn() (two target-free windows: SAC's
    # critic target moves every step)
    "cfg3_learn20": dict(S=64, A=8, hidden=[256, 256], N=4096, B=1024, rounds=20, input_seed=31,
                         learn_seed=41, noise_seed=2024, perm_seed=7),
}
PPO_LONG = {
    # BASELINE config 4: the 65 536-transition rollout, 2 epochs' worth of 4096-row minibatches
    "cfg4_learn32": dict(S=256, A=16, hidden=[256, 256], N=65536, B=4096, rounds=32, epsilon=0.1,
                         input_seed=4, learn_seed=43, perm_seed=8),
    # the reference's DEFAULT epsilon = 0.0 (ppo.py:105) at the benchmark's minibatch size: the
    # clipped surrogate passes a gradient only where ratio * gae <= gae (torch.min's tie rule)
    "cfg4_eps0": dict(S=256, A=16, hidden=[256, 256], N=8192, B=4096, rounds=6, epsilon=0.0,
                      input_seed=14, learn_seed=44, perm_seed=9),
}
BANDIT_LONG = {
    "cfg5_steps20": dict(F=512, hidden=[256, 64], B=4096, steps=20, input_seed=15, perm_seed=10),
}


def clone_sd(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def space(n):
    return DiscreteActionSpace([torch.tensor([k]) for k in range(n)])


class _ListSampler:
    """Stands in for the `random` module inside tensor_based_replay_buffer (:276
    `random.sample(self.memory, batch_size)`): hands out the recorded index lists."""

    def __init__(self, lists):
        self.lists, self.k = lists, 0

    def sample(self, population, k):
        idx = self.lists[self.k]
        self.k += 1
        assert len(idx) == k
        items = list(population)          # (deque random access is O(n); one pass instead)
        return [items[i] for i in idx]


def with_lists(lists, fn):
    real = TBRB.random
    TBRB.random = _ListSampler(lists)
    try:
        return fn()
    finally:
        TBRB.random = real


from pearl.history_summarization_modules.identity_history_summarization_module import (  # noqa: E402
    IdentityHistorySummarizationModule,
)


class Cast64(IdentityHistorySummarizationModule):
    """The hook through which every state reaches the networks (policy_learner.py:203-205,
    ppo.py:226, :268): the float64 run casts there."""

    def forward(self, x):
        return x.double()


def to_fp64(pl):
    pl.double()
    pl._history_summarization_module = Cast64()
    return pl


def delta32(sd64, sd32):
    return {k: (sd64[k].double() - v.double()).float() for k, v in sd32.items()
            if v.is_floating_point()}


def permute_lists(lists, seed):
    g = torch.Generator().manual_seed(seed)
    perms = [torch.randperm(len(row), generator=g) for row in lists]
    return [[row[int(j)] for j in p] for row, p in zip(lists, perms)], perms


def reports_tensor(report, keys):
    return {k: torch.tensor([float(v) for v in report[k]], dtype=torch.float64) for k in keys}


# ------------------------------------------------------------------------------------------ SAC
def sac_learner(cfg, sp, init=None):
    torch.manual_seed(6)
    pl = ContinuousSoftActorCritic(action_space=sp, state_dim=cfg["S"],
                                   actor_hidden_dims=cfg["hidden"], critic_hidden_dims=cfg["hidden"],
                                   batch_size=cfg["B"], training_rounds=cfg["rounds"])
    if init is not None:
        pl._actor.load_state_dict(init["actor0"])
        pl._critic.load_state_dict(init["critic0"])
        pl._critic_target.load_state_dict(init["critic_target0"])
    return pl


def sac_buffer(cfg, pl, sp, inputs):
    states, actions, rewards, term = inputs
    rb = BasicReplayBuffer(cfg["N"])
    PearlAgent(policy_learner=pl, replay_buffer=rb)
    for i in range(cfg["N"]):
        rb.push(state=states[i], action=actions[i], reward=float(rewards[i]),
                terminated=bool(term[i]), truncated=False, curr_available_actions=sp,
                next_state=states[i + 1], next_available_actions=sp)
    return rb


def make_sac_long(name, cfg):
    t0 = time.time()
    N, B, A, R = cfg["N"], cfg["B"], cfg["A"], cfg["rounds"]
    low, high = FI.sac_box(cfg)
    sp = BoxActionSpace(low=low, high=high)
    inputs = FI.sac_transitions(cfg)
    pl = sac_learner(cfg, sp)
    fx = {"config": dict(cfg), "low": low, "high": high,
          "actor0": clone_sd(pl._actor), "critic0": clone_sd(pl._critic),
          "critic_target0": clone_sd(pl._critic_target),
          "checksums": dict(states=FI.checksum(inputs[0]), actions=FI.checksum(inputs[1]))}
    rb = sac_buffer(cfg, pl, sp, inputs)
    random.seed(cfg["learn_seed"])
    lists = [random.sample(range(N), B) for _ in range(R)]
    # the noise learn() will draw (Normal.rsample -> _standard_normal -> torch.normal(zeros, ones)):
    # per round the actor update's draw, then the critic target's (:131-231)
    torch.manual_seed(cfg["noise_seed"])
    noise = torch.stack([torch.stack([torch.normal(torch.zeros(B, A), torch.ones(B, A)) for _ in range(2)])
                         for _ in range(R)])                       # [R, 2, B, A]
    random.seed(cfg["learn_seed"])
    torch.manual_seed(cfg["noise_seed"])
    rep = pl.learn(rb)                                             # the reference, untouched
    keys = ("actor_loss", "critic_loss", "entropy_coef")
    fx["lists"], fx["noise"] = torch.tensor(lists), noise
    fx["reports"] = reports_tensor(rep, keys)
    after = {"actor": clone_sd(pl._actor), "critic": clone_sd(pl._critic),
             "critic_target": clone_sd(pl._critic_target)}
    fx["after"] = after
    fx["log_entropy_after"] = pl._log_entropy.detach().clone()

    def replay(lists_, noise_, fp64=False):
        """learn() of a fresh learner on recorded index lists and recorded noise."""
        p2 = sac_learner(cfg, sp, fx)
        rb2 = sac_buffer(cfg, p2, sp, inputs)
        if fp64:
            to_fp64(p2)
        q = [noise_[r, j] for r in range(R) for j in range(2)]
        real = TDN._standard_normal
        TDN._standard_normal = lambda shape, dtype, device: q.pop(0).reshape(shape)
        try:
            rep2 = with_lists(lists_, lambda: p2.learn(rb2))
        finally:
            TDN._standard_normal = real
        assert not q
        return p2, rep2

    # self-check of the recording: replaying lists + noise IS the seeded run, bit for bit
    p_same, rep_same = replay(lists, noise)
    assert rep_same == rep, "recorded lists / noise do not reproduce learn()"
    for k, v in clone_sd(p_same._actor).items():
        assert torch.equal(v, after["actor"][k]), k
    # the twin: rows of every batch (and their noise) in another order
    plists, perms = permute_lists(lists, cfg["perm_seed"])
    pnoise = torch.stack([noise[r][:, perms[r]] for r in range(R)])
    p_twin, rep_twin = replay(plists, pnoise)
    fx["twin_reports"] = reports_tensor(rep_twin, keys)
    fx["twin_divergence"] = {
        "actor": FI.divergence(clone_sd(p_twin._actor), after["actor"]),
        "critic": FI.divergence(clone_sd(p_twin._critic), after["critic"]),
        "critic_target": FI.divergence(clone_sd(p_twin._critic_target), after["critic_target"])}
    fx["twin_log_entropy_after"] = p_twin._log_entropy.detach().clone()
    p64, rep64 = replay(lists, noise, fp64=True)
    assert p64._actor.fc_mu.weight.dtype == torch.float64
    fx["fp64_reports"] = reports_tensor(rep64, keys)
    fx["fp64_minus_ref"] = {"actor": delta32(clone_sd(p64._actor), after["actor"]),
                            "critic": delta32(clone_sd(p64._critic), after["critic"]),
                            "critic_target": delta32(clone_sd(p64._critic_target), after["critic_target"])}
    fx["fp64_log_entropy_after"] = p64._log_entropy.detach().clone()
    path = os.path.join(OUT, f"sac_{name}.pt")
    torch.save(fx, path)
    r64 = ((fx["fp64_reports"]["critic_loss"] - fx["reports"]["critic_loss"]).abs()
           / fx["reports"]["critic_loss"].abs())
    print(f"  sac {name}: ref vs fp64 critic_loss rel first/last {float(r64[0]):.2e} / {float(r64[-1]):.2e}")
    rel = ((fx["twin_reports"]["critic_loss"] - fx["reports"]["critic_loss"]).abs()
           / fx["reports"]["critic_loss"].abs())
    print(f"sac {name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s); "
          f"critic_loss {float(fx['reports']['critic_loss'][0]):.5f} -> "
          f"{float(fx['reports']['critic_loss'][-1]):.5f}; twin rel diff first/last "
          f"{float(rel[0]):.2e} / {float(rel[-1]):.2e}")


# ------------------------------------------------------------------------------------------ PPO
def ppo_learner(cfg, init=None):
    torch.manual_seed(5)
    A = cfg["A"]
    pl = ProximalPolicyOptimization(
        action_space=space(A), state_dim=cfg["S"], actor_hidden_dims=cfg["hidden"],
        critic_hidden_dims=cfg["hidden"], training_rounds=cfg["rounds"], batch_size=cfg["B"],
        epsilon=cfg["epsilon"], action_representation_module=OneHotActionTensorRepresentationModule(A))
    if init is not None:
        pl._actor.load_state_dict(init["actor0"])
        pl._critic.load_state_dict(init["critic0"])
    return pl


def ppo_buffer(cfg, pl, rollout):
    states, actions, rewards, term, trunc = rollout
    N, A = cfg["N"], cfg["A"]
    rb = PPOReplayBuffer(N + 5)
    PearlAgent(policy_learner=pl, replay_buffer=rb)
    sp = space(A)
    acts, rews, te, tr = actions.tolist(), rewards.tolist(), term.tolist(), trunc.tolist()
    for i in range(N):
        rb.push(state=states[i], action=torch.tensor([acts[i]]), reward=rews[i], terminated=te[i],
                truncated=tr[i], curr_available_actions=sp, next_state=states[i + 1],
                next_available_actions=sp, max_number_actions=A)
    return rb


def make_ppo_long(name, cfg):
    t0 = time.time()
    N, B, R = cfg["N"], cfg["B"], cfg["rounds"]
    rollout = FI.ppo_rollout(cfg)
    pl = ppo_learner(cfg)
    fx = {"config": dict(cfg), "actor0": clone_sd(pl._actor), "critic0": clone_sd(pl._critic),
          "checksums": dict(states=FI.checksum(rollout[0]), actions=FI.checksum(rollout[1]),
                            rewards=FI.checksum(rollout[2]))}
    rb = ppo_buffer(cfg, pl, rollout)
    random.seed(cfg["learn_seed"])
    lists = [random.sample(range(N), B) for _ in range(R)]
    # what the first round's surrogate sees (ppo.py:160-176): ratio of the minibatch forward to the
    # rollout forward under UNCHANGED parameters
    probe = {}
    real_loss = pl._actor_loss

    def spy(batch):
        if "ratio0" not in probe:
            with torch.no_grad():
                p = pl._actor.get_action_prob(
                    state_batch=batch.state, action_batch=batch.action,
                    available_actions=batch.curr_available_actions,
                    unavailable_actions_mask=batch.curr_unavailable_actions_mask)
                probe["ratio0"] = torch.div(p, batch.action_probs).clone()
                probe["gae0"] = batch.gae.clone()
        return real_loss(batch)

    pl._actor_loss = spy
    random.seed(cfg["learn_seed"])
    rep = pl.learn(rb)
    keys = ("actor_loss", "critic_loss")
    fx["lists"] = torch.tensor(lists)
    fx["reports"] = reports_tensor(rep, keys)
    fx["ratio0"], fx["gae0"] = probe["ratio0"].view(-1), probe["gae0"].view(-1)
    after = {"actor": clone_sd(pl._actor), "critic": clone_sd(pl._critic)}
    fx["after"] = after
    print(f"  ppo {name}: reference run {time.time() - t0:.0f} s; round-0 ratio == 1 exactly for "
          f"{int((fx['ratio0'] == 1).sum())} of {B} rows, max |ratio - 1| "
          f"{float((fx['ratio0'] - 1).abs().max()):.2e}")
    plists, _ = permute_lists(lists, cfg["perm_seed"])
    p2 = ppo_learner(cfg, fx)
    rb2 = ppo_buffer(cfg, p2, rollout)
    rep_twin = with_lists(plists, lambda: p2.learn(rb2))
    fx["twin_reports"] = reports_tensor(rep_twin, keys)
    fx["twin_divergence"] = {"actor": FI.divergence(clone_sd(p2._actor), after["actor"]),
                             "critic": FI.divergence(clone_sd(p2._critic), after["critic"])}
    p3 = ppo_learner(cfg, fx)
    rb3 = ppo_buffer(cfg, p3, rollout)
    to_fp64(p3)
    rep64 = with_lists(lists, lambda: p3.learn(rb3))
    assert p3._actor._model[0][0].weight.dtype == torch.float64
    fx["fp64_reports"] = reports_tensor(rep64, keys)
    fx["fp64_minus_ref"] = {"actor": delta32(clone_sd(p3._actor), after["actor"]),
                            "critic": delta32(clone_sd(p3._critic), after["critic"])}
    r64 = ((fx["fp64_reports"]["actor_loss"] - fx["reports"]["actor_loss"]).abs()
           / fx["reports"]["actor_loss"].abs().clamp_min(1e-3))
    print(f"  ppo {name}: ref vs fp64 actor_loss rel first/last {float(r64[0]):.2e} / {float(r64[-1]):.2e}")
    path = os.path.join(OUT, f"ppo_{name}.pt")
    torch.save(fx, path)
    rel = ((fx["twin_reports"]["actor_loss"] - fx["reports"]["actor_loss"]).abs()
           / fx["reports"]["actor_loss"].abs().clamp_min(1e-3))
    print(f"ppo {name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s); "
          f"actor_loss {float(fx['reports']['actor_loss'][0]):.4f} -> "
          f"{float(fx['reports']['actor_loss'][-1]):.4f}; twin rel diff first/last "
          f"{float(rel[0]):.2e} / {float(rel[-1]):.2e}")


# --------------------------------------------------------------------------------------- bandit
def make_bandit_long(name, cfg):
    from pearl.policy_learners.contextual_bandits.neural_linear_bandit import NeuralLinearBandit
    t0 = time.time()
    F, B, K = cfg["F"], cfg["B"], cfg["steps"]

    def learner(init=None):
        torch.manual_seed(8)
        pl = NeuralLinearBandit(feature_dim=F, hidden_dims=cfg["hidden"], batch_size=B,
                                learning_rate=1e-3)
        if init is not None:
            pl.model.load_state_dict(init)
        return pl

    pl = learner()
    fx = {"config": dict(cfg), "model0": clone_sd(pl.model), "batches": [], "checksums": []}
    gen = torch.Generator().manual_seed(77)
    wtrue = torch.randn(F, generator=gen) / F ** 0.5
    twin = learner(fx["model0"])
    p64 = learner(fx["model0"])
    p64.model.double()
    pg = torch.Generator().manual_seed(cfg["perm_seed"])
    reports, twin_reports, reports64 = [], [], []
    for k in range(K):
        x = FI.bandit_contexts(cfg, k)
        r = torch.sigmoid(x @ wtrue) + 0.05 * torch.randn(B, generator=gen)
        w = None if k % 2 == 0 else torch.rand(B, generator=gen) + 0.5
        rep = pl.learn_batch(TransitionBatch(state=x, action=torch.zeros(B, 1), reward=r, weight=w))
        p = torch.randperm(B, generator=pg)
        rep2 = twin.learn_batch(TransitionBatch(state=x[p], action=torch.zeros(B, 1), reward=r[p],
                                                weight=None if w is None else w[p]))
        rep3 = p64.learn_batch(TransitionBatch(state=x.double(), action=torch.zeros(B, 1),
                                               reward=r.double(),
                                               weight=None if w is None else w.double()))
        reports64.append((float(rep3["loss"]), float(rep3["mu_scores"])))
        fx["batches"].append(dict(reward=r, weight=w))
        fx["checksums"].append(FI.checksum(x))
        reports.append((float(rep["loss"]), float(rep["mu_scores"])))
        twin_reports.append((float(rep2["loss"]), float(rep2["mu_scores"])))
    fx["reports"] = torch.tensor(reports, dtype=torch.float64)
    fx["twin_reports"] = torch.tensor(twin_reports, dtype=torch.float64)
    fx["model_after"] = clone_sd(pl.model)
    fx["twin_divergence"] = FI.divergence(clone_sd(twin.model), fx["model_after"])
    fx["fp64_reports"] = torch.tensor(reports64, dtype=torch.float64)
    fx["fp64_minus_ref"] = delta32(clone_sd(p64.model), fx["model_after"])
    xq = FI.normalish((64, F), cfg["input_seed"] * 100 + 99)
    with torch.no_grad():
        fx["query"] = dict(sigma=pl.model.calculate_sigma(xq).clone(), mu=pl.model(xq).clone(),
                           twin_sigma=twin.model.calculate_sigma(xq).clone(),
                           twin_mu=twin.model(xq).clone(),
                           fp64_sigma=p64.model.calculate_sigma(xq.double()).clone(),
                           fp64_mu=p64.model(xq.double()).clone())
    path = os.path.join(OUT, f"bandit_{name}.pt")
    torch.save(fx, path)
    rel = (fx["twin_reports"][:, 0] - fx["reports"][:, 0]).abs() / fx["reports"][:, 0].abs()
    print(f"bandit {name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB, {time.time() - t0:.0f} s); "
          f"loss {reports[0][0]:.5f} -> {reports[-1][0]:.5f}; twin rel diff first/last "
          f"{float(rel[0]):.2e} / {float(rel[-1]):.2e}")


def main():
    os.makedirs(OUT, exist_ok=True)
    want = set(sys.argv[1:]) or {"sac", "ppo", "ppo_eps0", "bandit"}
    if "sac" in want:
        for name, cfg in SAC_LONG.items():
            make_sac_long(name, cfg)
    if "bandit" in want:
        for name, cfg in BANDIT_LONG.items():
            make_bandit_long(name, cfg)
    if "ppo_eps0" in want:
        make_ppo_long("cfg4_eps0", PPO_LONG["cfg4_eps0"])
    if "ppo" in want:
        make_ppo_long("cfg4_learn32", PPO_LONG["cfg4_learn32"])


if __name__ == "__main__":
    main()
