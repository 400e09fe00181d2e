"""GPU: the benchmarked native loops of BASELINE configs 3 / 4 / 5 — pa_sac_learn, pa_ppo_learn,
pa_bandit_step x N — over MANY rounds at FULL size against runs of the real reference
(oracle/make_golden_long.py -> tests/golden/{sac_cfg3_learn20, ppo_cfg4_learn32, ppo_cfg4_eps0,
bandit_cfg5_steps20}.pt), on the reference's own index lists and reparameterisation noise.

Two fp32 implementations of one AdamW trajectory do not stay bit-close, so each fixture carries its
yardsticks, produced by the reference itself:

  twin   the reference on the same minibatches with their rows in another order (summation order);
  fp64   the reference's modules in float64 on the same inputs — what both fp32 runs approximate.

The criterion, stated once.  Reports, per block of rounds:

    rel_err(HIP, fp64)  <=  4 * rel_err(reference fp32, fp64)  +  floor      (floor: a few fp32 ulps
                                                                              of the reported sum)
Final parameters, per tensor (rms and mean |difference| to the fp64 run):

    dist(HIP, fp64)  <=  max( 4 * dist(reference fp32, fp64),  1e-3 * lr * rounds )

i.e. as close to the exact trajectory as the reference is within a factor of four — or, where the
reference is still at rounding level (short runs: MKL's blocked sums leave it 2 ulps from float64
after 6 rounds), within 0.1 % of the distance AdamW can move a parameter in that many rounds; and
no single element further than 2.5 lr per round (what sign flips of noise-level gradients can
accumulate).  The same form as DQN's two-oracle test
(test_gpu_dqn.py::test_full_size_200_round_loss_curve_against_the_oracle).

Measured on MI355X (profiles/r06_a_long_runs.txt; worst tensor, rms, HIP vs reference):
SAC cfg3 20 rounds 1.26e-4 vs 1.26e-4 (ratio 1.01 — both carry log pi's cancellation);
PPO cfg4 32 rounds 5.4e-6 vs 3.6e-6 (ratio 1.5, last-layer bias 7.3; with the fp32-MFMA row step —
PEARL_AMD_TEST_ROWSTEP_SPLIT=0 — 1.02: the difference is the bf16x3 row step, not the order of sums);
PPO epsilon = 0, 6 rounds 3.9e-7 vs 1.1e-8 and bandit cfg5 20 steps 7.5e-6 vs 4.1e-7: the reference
is at rounding level there and the HIP loop is 20-35x further — 0.07 % / 0.04 % of the AdamW travel,
heavy-tailed (max / rms ~ 19: elements whose gradient is at the level of AdamW's eps).  That ratio
measures WHEN the first noise-level event happens, not how sums are formed
(profiles/r06_e_long_run_distance_is_event_driven.txt): the HIP gradients of step 1 are closer to
float64 than MKL's in every tensor, for four steps the HIP run is twice as close as a CPU fp32 run,
both jump by the same 1.2e-7 at one step and grow with the dynamics from there — and two CPU runs of
the same torch code on two hosts end 6.6x apart from each other."""
import os
import random

import pytest
import torch

from conftest import GOLDEN_DIR
from oracle import fixture_inputs as FI

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
K_REPORT = 4.0
K_PARAM = 4.0


@pytest.fixture(autouse=True)
def _arithmetic_override():
    """Diagnostic switch (not used by the suite): PEARL_AMD_TEST_DW_SPLIT=0 runs these tests with the
    fp32-MFMA weight-gradient loop, PEARL_AMD_TEST_ROWSTEP_SPLIT=0 with the fp32-MFMA row step —
    to tell the arithmetic's share of a distance from the summation order's."""
    from pearl_amd import _native as N
    dw, rs = os.environ.get("PEARL_AMD_TEST_DW_SPLIT"), os.environ.get("PEARL_AMD_TEST_ROWSTEP_SPLIT")
    if dw is not None:
        N.check(N.lib().pa_debug_set_dw_split(int(dw)))
    if rs is not None:
        N.check(N.lib().pa_debug_set_rowstep_split(int(rs)))
    yield
    if dw is not None:
        N.check(N.lib().pa_debug_set_dw_split(-1))
    if rs is not None:
        N.check(N.lib().pa_debug_set_rowstep_split(-1))


def load(name):
    return torch.load(os.path.join(GOLDEN_DIR, f"{name}.pt"), map_location="cpu", weights_only=False)


def dspace(n):
    from pearl_amd import DiscreteActionSpace
    return DiscreteActionSpace([torch.tensor([k]) for k in range(n)])


def fp64_params(fx, net, key=None):
    """The float64 run's final parameters: reference fp32 + stored difference."""
    ref = fx["after"][net] if key is None else fx[key]
    delta = fx["fp64_minus_ref"][net] if key is None else fx["fp64_minus_ref"]
    return {k: ref[k].double() + delta[k].double() for k in delta}


def check_reports(got, fx, keys, blocks, floor, label):
    """rel. error of the per-round reports against fp64, HIP next to the reference, per block."""
    print(f"\n{label}: rel. error of the reported values against the float64 run (max per block)")
    print("  key           rounds     HIP        reference   twin-vs-ref")
    for j, k in enumerate(keys):
        ref = fx["reports"][k] if isinstance(fx["reports"], dict) else fx["reports"][:, j]
        f64 = fx["fp64_reports"][k] if isinstance(fx["fp64_reports"], dict) else fx["fp64_reports"][:, j]
        twin = fx["twin_reports"][k] if isinstance(fx["twin_reports"], dict) else fx["twin_reports"][:, j]
        g = torch.tensor(got[k], dtype=torch.float64)
        scale = f64.abs().clamp_min(1e-3 * float(f64.abs().max()))
        e_hip, e_ref, e_twin = (g - f64).abs() / scale, (ref - f64).abs() / scale, (twin - ref).abs() / scale
        for lo, hi in blocks:
            a, b, c = float(e_hip[lo:hi].max()), float(e_ref[lo:hi].max()), float(e_twin[lo:hi].max())
            print(f"  {k:12s}  {lo:3d}-{hi:3d}   {a:.2e}   {b:.2e}    {c:.2e}")
            assert a <= K_REPORT * b + floor, (k, lo, hi, a, b)


def check_params(hip_sd, ref_sd, f64_sd, lr, rounds, label):
    """Final parameters: rms and mean |difference| to fp64 within K of the reference's own or
    0.1 % of the AdamW travel lr * rounds; no element further than sign flips can carry it (2.5 lr
    per round)."""
    d_hip = FI.divergence(hip_sd, f64_sd)
    d_ref = FI.divergence(ref_sd, f64_sd)
    print(f"\n{label}: distance of the final parameters to the float64 run (rms | mean | max)")
    worst = 0.0
    for k in f64_sd:
        if k not in d_hip or "_critic_networks_combined" in k or d_hip[k]["ref_rms"] == 0:
            continue
        h, r = d_hip[k], d_ref[k]
        floor = 2.0 ** -24 * max(r["ref_rms"], 1e-30)       # fp32 resolution of the tensor's scale
        print(f"  {k:44s} HIP {h['rms']:.2e} {h['mean_abs']:.2e} {h['max_abs']:.2e} | "
              f"ref {r['rms']:.2e} {r['mean_abs']:.2e} {r['max_abs']:.2e}")
        travel = 1e-3 * lr * rounds
        assert h["rms"] <= max(K_PARAM * r["rms"], travel) + floor, (k, h, r)
        assert h["mean_abs"] <= max(K_PARAM * r["mean_abs"], travel) + floor, (k, h, r)
        assert h["max_abs"] <= max(K_PARAM * r["max_abs"], 2.5 * lr * rounds), (k, h, r)
        worst = max(worst, h["rms"] / max(r["rms"], floor))
    print(f"  worst rms ratio HIP / reference: {worst:.2f}")


# ------------------------------------------------------------------------------------------ SAC
class RecordedNoise:
    """The reparameterisation noise the reference drew (torch's global generator: per round the
    actor update's draw, then the critic target's) — per call for learn_batch, per block of rounds
    for the native loop (ContinuousSoftActorCritic.noise_source)."""

    def __init__(self, noise):
        self.noise, self.calls = noise, 0          # [R, 2, B, A]

    def __call__(self, B, A, dev):
        r, j = divmod(self.calls, 2)
        self.calls += 1
        return self.noise[r, j]

    def rounds(self, first, n, B, A, dev):
        self.calls += 2 * n
        return self.noise[first:first + n].to(dev)


def test_sac_cfg3_learn_20_rounds_native_loop_against_the_reference():
    """BASELINE config 3 (S = 64, 8-dim actions, [256, 256], B = 1024): learn() = 20 rounds through
    pa_sac_learn on the reference's index lists and noise (soft_actor_critic_continuous.py:131-231
    under policy_learner.py:162-195)."""
    from pearl_amd import BasicReplayBuffer, BoxActionSpace, ContinuousSoftActorCritic, PearlAgent
    fx = load("sac_cfg3_learn20")
    cfg = fx["config"]
    N, B, R = cfg["N"], cfg["B"], cfg["rounds"]
    states, actions, rewards, term = FI.sac_transitions(cfg)
    assert FI.checksum(states) == fx["checksums"]["states"]
    assert FI.checksum(actions) == fx["checksums"]["actions"]
    pl = ContinuousSoftActorCritic(action_space=BoxActionSpace(fx["low"], fx["high"]), state_dim=cfg["S"],
                                   actor_hidden_dims=cfg["hidden"], critic_hidden_dims=cfg["hidden"],
                                   batch_size=B, training_rounds=R)
    pl._actor.load_state_dict(fx["actor0"])
    pl._critic.load_state_dict(fx["critic0"])
    pl._critic_target.load_state_dict(fx["critic_target0"])
    rb = BasicReplayBuffer(N, sampler="python")
    PearlAgent(pl, replay_buffer=rb, device_id=0)
    rb.push_many(state=states[:-1].to(DEV), action=actions.to(DEV), reward=rewards.to(DEV),
                 terminated=term.to(DEV), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
                 next_state=states[1:].to(DEV))
    pl.noise_source = RecordedNoise(fx["noise"])
    random.seed(cfg["learn_seed"])
    got = pl.learn(rb)
    assert pl._flat.get("loop_ws") is not None, "learn() did not take the native loop (pa_sac_learn)"
    assert pl.noise_source.calls == 2 * R and pl._training_steps == R
    assert torch.equal(rb.last_indices.cpu(), fx["lists"][-1])        # the reference's index stream
    keys = ("actor_loss", "critic_loss", "entropy_coef")
    assert all(len(got[k]) == R for k in keys)
    # first round, before any drift: the reference's own values (log pi is ill-conditioned where a
    # component saturates — the reference is 1.8e-6 / 2.9e-4 from float64 here itself)
    for k, tol in (("actor_loss", 2e-5), ("critic_loss", 1e-3), ("entropy_coef", 2e-5)):
        want = float(fx["reports"][k][0])
        assert abs(got[k][0] - want) <= tol * max(1.0, abs(want)), (k, got[k][0], want)
    check_reports(got, fx, keys, ((0, 1), (1, 5), (5, 10), (10, 20)), 2e-6, "SAC cfg3")
    for net, mod in (("actor", pl._actor), ("critic", pl._critic), ("critic_target", pl._critic_target)):
        hip = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
        check_params(hip, fx["after"][net], fp64_params(fx, net), 1e-3, R, f"SAC cfg3 {net}")
    le = float(pl._log_entropy.detach().cpu())
    e_hip = abs(le - float(fx["fp64_log_entropy_after"]))
    e_ref = abs(float(fx["log_entropy_after"]) - float(fx["fp64_log_entropy_after"]))
    print(f"log_entropy: HIP {e_hip:.2e} from float64, reference {e_ref:.2e}")
    assert e_hip <= K_PARAM * e_ref + 1e-7


# ------------------------------------------------------------------------------------------ PPO
def make_ppo_full(fx):
    from pearl_amd import (OneHotActionTensorRepresentationModule, PearlAgent, PPOReplayBuffer,
                           ProximalPolicyOptimization)
    cfg = fx["config"]
    A, N = cfg["A"], cfg["N"]
    states, actions, rewards, term, trunc = FI.ppo_rollout(cfg)
    for k, v in (("states", states), ("actions", actions), ("rewards", rewards)):
        assert FI.checksum(v) == fx["checksums"][k], k
    pl = ProximalPolicyOptimization(
        action_space=dspace(A), state_dim=cfg["S"], actor_hidden_dims=cfg["hidden"],
        critic_hidden_dims=cfg["hidden"], training_rounds=cfg["rounds"], batch_size=cfg["B"],
        epsilon=cfg["epsilon"], action_representation_module=OneHotActionTensorRepresentationModule(A))
    pl._actor.load_state_dict(fx["actor0"])
    pl._critic.load_state_dict(fx["critic0"])
    rb = PPOReplayBuffer(N + 5, sampler="python")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    rb.push_many(state=states[:-1].to(DEV), action=actions.view(-1, 1).to(DEV), reward=rewards.to(DEV),
                 terminated=term.to(DEV), truncated=trunc.to(DEV), next_state=states[1:].to(DEV),
                 curr_available_actions=dspace(A), next_available_actions=dspace(A),
                 max_number_actions=A)
    return pl, rb, agent


def test_ppo_cfg4_learn_32_rounds_native_loop_against_the_reference():
    """BASELINE config 4 (S = 256, 16 actions, [256, 256]): the 65 536-transition rollout, learn() =
    preprocess_replay_buffer + 32 minibatches of 4096 (two epochs' worth) through pa_ppo_learn on
    the reference's index lists (ppo.py:152-293)."""
    fx = load("ppo_cfg4_learn32")
    cfg = fx["config"]
    R = cfg["rounds"]
    pl, rb, agent = make_ppo_full(fx)
    random.seed(cfg["learn_seed"])
    got = pl.learn(rb)
    assert pl._flat.get("loop_ws") is not None, "learn() did not take the native loop (pa_ppo_learn)"
    assert torch.equal(rb.last_indices.cpu(), fx["lists"][-1])
    keys = ("actor_loss", "critic_loss")
    assert all(len(got[k]) == R for k in keys) and pl._training_steps == R
    # first round: parameters untouched, the rollout's own probabilities -> the reference's values
    for k in keys:
        want = float(fx["reports"][k][0])
        assert abs(got[k][0] - want) <= 2e-5 * max(1.0, abs(want)), (k, got[k][0], want)
    # the actor loss is a SUM over 4096 rows of terms that cancel (sum of -ratio * gae ~ 1e3 with
    # terms ~ 1): its fp32 resolution relative to its own magnitude is the floor
    check_reports(got, fx, keys, ((0, 1), (1, 8), (8, 16), (16, 32)), 5e-6, "PPO cfg4")
    for net, mod in (("actor", pl._actor), ("critic", pl._critic)):
        hip = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
        check_params(hip, fx["after"][net], fp64_params(fx, net), 1e-4, R, f"PPO cfg4 {net}")
    random.seed(1)
    agent.learn()                       # on-policy: PearlAgent.learn clears the rollout (:217-218)
    assert len(rb) == 0


def test_ppo_epsilon_zero_at_the_benchmark_minibatch():
    """The reference's default epsilon = 0.0 (ppo.py:105) at B = 4096.  clamp(ratio, 1, 1) = 1, so a
    row passes a gradient only where ratio * gae <= gae — and exactly where ratio == 1, through
    torch.min's tie rule and clamp's closed interval.  In the first round the reference's ratio IS
    exactly 1 in every row (`ratio0`, recorded: its minibatch forward reproduces its rollout
    forward), so every row with gae != 0 trains.

    Criterion.  (1) The HIP step must make the same decision in that round: no row with gae != 0
    may have an all-zero logit gradient — it would, in about half the rows, if the minibatch
    forward differed from the rollout forward in the last bits (a step with epsilon = 0 therefore
    keeps the fp32-MFMA forward, bitwise the rollout's; run_rowstep in mlp.hip).  (2) From round 2
    on |ratio - 1| is ~1e-3, far from an ulp, and the run is held to the reference like any other:
    reports and final parameters within K of the reference's own distance to float64."""
    from pearl_amd import _native as N
    fx = load("ppo_cfg4_eps0")
    cfg = fx["config"]
    R, B, A = cfg["rounds"], cfg["B"], cfg["A"]
    assert cfg["epsilon"] == 0.0 and bool((fx["ratio0"] == 1).all())
    # (1) one round on the reference's first index list
    pl, rb, _ = make_ppo_full(fx)
    pl._training_rounds = 1
    random.seed(cfg["learn_seed"])
    first = pl.learn(rb)
    assert int(N.lib().pa_rowstep_last_split()) == 0, "epsilon = 0 took the bf16x3 forward"
    ws = pl._flat.get("loop_ws")
    assert ws is not None and torch.equal(rb.last_indices.cpu(), fx["lists"][0])
    d_logits = ws["d_logits"].cpu()
    gae = fx["gae0"]
    dead = (d_logits == 0).all(dim=1)
    assert int((dead & (gae != 0)).sum()) == 0, f"{int(dead.sum())} rows of {B} got no gradient"
    # with ratio == 1: d loss / d logit_j = -gae * (onehot_j - p_j) * p_a / p_old = -gae (onehot_j - p_j) p_a / p_a
    assert bool(((d_logits.sum(dim=1)).abs() <= 1e-5 * gae.abs() + 1e-7).all())   # sums to 0 per row
    want0 = float(fx["reports"]["actor_loss"][0])
    assert abs(first["actor_loss"][0] - want0) <= 2e-5 * abs(want0)
    # (2) the whole run
    pl, rb, _ = make_ppo_full(fx)
    random.seed(cfg["learn_seed"])
    got = pl.learn(rb)
    keys = ("actor_loss", "critic_loss")
    check_reports(got, fx, keys, ((0, 1), (1, 3), (3, R)), 5e-6, "PPO eps=0")
    for net, mod in (("actor", pl._actor), ("critic", pl._critic)):
        hip = {k: v.detach().cpu() for k, v in mod.state_dict().items()}
        check_params(hip, fx["after"][net], fp64_params(fx, net), 1e-4, R, f"PPO eps=0 {net}")


# --------------------------------------------------------------------------------------- bandit
def test_bandit_cfg5_20_steps_against_the_reference():
    """BASELINE config 5 (512-dim contexts, hidden [256, 64], B = 4096): 20 learn_batch calls
    (pa_bandit_step: network step + LinUCB moments + fp64 solve on a side stream) against the
    reference's trajectory (neural_linear_bandit.py:159-225, linear_regression.py:111-121, :192-270)."""
    from pearl_amd import NeuralLinearBandit, TransitionBatch
    fx = load("bandit_cfg5_steps20")
    cfg = fx["config"]
    F, B, K = cfg["F"], cfg["B"], cfg["steps"]
    pl = NeuralLinearBandit(feature_dim=F, hidden_dims=cfg["hidden"], batch_size=B, learning_rate=1e-3)
    pl.model.load_state_dict(fx["model0"])
    pl.to(DEV)
    got = {"loss": [], "mu": []}
    for k in range(K):
        x = FI.bandit_contexts(cfg, k)
        assert FI.checksum(x) == fx["checksums"][k]
        b = fx["batches"][k]
        rep = pl.learn_batch(TransitionBatch(
            state=x.to(DEV), action=torch.zeros(B, 1, device=DEV), reward=b["reward"].to(DEV),
            weight=None if b["weight"] is None else b["weight"].to(DEV)))
        got["loss"].append(float(rep["loss"]))
        got["mu"].append(float(rep["mu_scores"]))
    want0 = float(fx["reports"][0, 0])
    assert abs(got["loss"][0] - want0) <= 1e-5 * max(1.0, abs(want0))
    check_reports(got, fx, ("loss", "mu"), ((0, 1), (1, 5), (5, 10), (10, 20)), 5e-6, "bandit cfg5")
    sd = {k: v.detach().cpu() for k, v in pl.model.state_dict().items()}
    f64 = fp64_params(fx, None, key="model_after")
    nn_keys = [k for k in f64 if k.startswith("_nn_layers") or k.startswith("linear_layer_e2e")]
    check_params({k: sd[k] for k in nn_keys}, {k: fx["model_after"][k] for k in nn_keys},
                 {k: f64[k] for k in nn_keys}, 1e-3, K, "bandit cfg5 network")
    # LinUCB moments: sums of 81 920 rank-1 terms; A and b against float64, relative to their scale.
    # A chained fp32 sum of n terms carries ~sqrt(n) 2^-24 = 1.7e-5 of the terms' scale; measured
    # 1.7e-6 of max |A| here, MKL's blocked sums 1.2e-7: held to 4e-6 (or 4x the reference's).
    for key in ("_linear_regression_layer._A", "_linear_regression_layer._b"):
        scale = float(f64[key].abs().max())
        e_hip = float((sd[key].double() - f64[key]).abs().max()) / scale
        e_ref = float((fx["model_after"][key].double() - f64[key]).abs().max()) / scale
        print(f"  {key}: max error / max |.|  HIP {e_hip:.2e}  reference {e_ref:.2e}")
        assert e_hip <= max(K_PARAM * e_ref, 4e-6)
    # what the regression is for: sigma and mu of fresh contexts.  Both go through inv(A + lambda I):
    # a relative perturbation e_A of the matrix moves the solution by up to cond * e_A, so the bar is
    # the reference's own error, or what the conditioning gives the measured error of A
    A64 = f64["_linear_regression_layer._A"]
    cond = float(torch.linalg.cond(A64 + torch.eye(A64.shape[0], dtype=torch.float64)))
    e_A = float((sd["_linear_regression_layer._A"].double() - A64).abs().max() / A64.abs().max())
    print(f"  cond(A + I) = {cond:.3g}, e_A = {e_A:.2e}: solution-level bound cond * e_A = {cond * e_A:.2e}")
    xq = FI.normalish((64, F), cfg["input_seed"] * 100 + 99).to(DEV)
    with torch.no_grad():
        mu = pl.model(xq).view(-1).cpu().double()
        sigma = pl.model.calculate_sigma(xq).view(-1).cpu().double()
    q = fx["query"]
    for name, g, ref, f in (("mu", mu, q["mu"], q["fp64_mu"]), ("sigma", sigma, q["sigma"], q["fp64_sigma"])):
        f = f.view(-1).double()
        scale = float(f.abs().max())
        e_hip = float((g - f).abs().max()) / scale
        e_ref = float((ref.view(-1).double() - f).abs().max()) / scale
        print(f"  {name} of 64 fresh contexts: HIP {e_hip:.2e}  reference {e_ref:.2e}  (of max |.|)")
        assert e_hip <= max(K_PARAM * e_ref + 1e-5, cond * e_A), (name, e_hip, e_ref)
