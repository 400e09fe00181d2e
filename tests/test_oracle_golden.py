"""CPU: pin the oracle (oracle/pearl_oracle.py) against fixtures minted by the REAL reference
(oracle/make_golden.py).  Replay contract bit-exact; learner numerics to fp32 round-off."""
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN_NAMES
from helpers import fill_oracle_replay, oracle_learner
from oracle import pearl_oracle as O


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_replay_contract_bit_exact(golden, name):
    fx = golden(name)
    rb = fill_oracle_replay(fx)
    cfg = fx["config"]
    assert len(rb) == cfg["N"]
    # same seed -> random.sample on the deque picks the positions the fixture recorded
    random.seed(fx["sample_seed"])
    got = rb.sample(cfg["B"])
    also = rb.sample_at(fx["sample_idx"].tolist())
    for k, want in fx["batch_raw"].items():
        assert got[k].dtype == want.dtype and got[k].shape == want.shape, k
        assert torch.equal(got[k], want), k
        assert torch.equal(also[k], want), k
    pre = O.preprocess(got, cfg["A"])
    for k, want in fx["batch_pre"].items():
        assert torch.equal(pre[k], want), k


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_random_sample_range_equals_deque(golden, name):
    """random.sample(range(n), k) and random.sample(deque_of_n, k) pick the same
    positions and leave the global RNG in the same state."""
    fx = golden(name)
    n, B = fx["config"]["N"], fx["config"]["B"]
    random.seed(fx["sample_seed"])
    idx = random.sample(range(n), B)
    state_after = random.getstate()
    assert idx == fx["sample_idx"].tolist()
    random.seed(fx["sample_seed"])
    rb = fill_oracle_replay(fx)
    random.sample(rb.memory, B)
    assert random.getstate() == state_after


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_one_batch_numerics(golden, name):
    fx = golden(name)
    pl = oracle_learner(fx)
    b = fx["batch_pre"]
    q = pl.q_values(b["state"], b["action"])
    nv = pl.next_state_values(b["next_state"], b["next_available_actions"],
                              b["next_unavailable_actions_mask"])
    y = pl.bellman_target(b)
    torch.testing.assert_close(q, fx["q"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(nv, fx["next_v"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(y, fx["target"], rtol=1e-5, atol=1e-6)
    q2, g = pl.gradients(b, fx["target"])
    torch.testing.assert_close(((q2 - fx["target"]) ** 2).mean(), fx["mse"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close((q2 - fx["target"]).abs().mean(), fx["mean_abs_td"], rtol=1e-5,
                               atol=1e-6)
    for k, want in fx["grads"].items():
        torch.testing.assert_close(g[k].reshape(want.shape), want, rtol=2e-4, atol=2e-6, msg=k)


@pytest.mark.parametrize("name", GOLDEN_NAMES)
def test_learn_trajectory(golden, name):
    """`rounds` steps of learn(): losses, parameters, target network and AdamW state."""
    fx = golden(name)
    cfg = fx["config"]
    pl = oracle_learner(fx)
    rb = fill_oracle_replay(fx)
    random.seed(fx["learn_seed"])
    losses = pl.learn(rb, cfg["rounds"], cfg["B"], cfg["A"])
    torch.testing.assert_close(torch.tensor(losses), fx["learn_losses"], rtol=2e-4, atol=1e-5)
    assert pl.training_steps == fx["training_steps_after"]
    for k in O.PARAM_KEYS:
        torch.testing.assert_close(pl.p[k], fx["params_after"][k], rtol=1e-3, atol=2e-5, msg=k)
        torch.testing.assert_close(pl.t[k], fx["target_after"][k], rtol=1e-3, atol=2e-5, msg=k)
        st = fx["opt_after"][k]
        assert float(st["step"]) == cfg["rounds"]
        torch.testing.assert_close(pl.m[k], st["exp_avg"], rtol=1e-3, atol=1e-6, msg=k)
        torch.testing.assert_close(pl.v[k], st["exp_avg_sq"], rtol=1e-3, atol=1e-8, msg=k)
        torch.testing.assert_close(pl.vmax[k], st["max_exp_avg_sq"], rtol=1e-3, atol=1e-8, msg=k)
    # the index lists recorded by the generator are the ones this run drew
    random.seed(fx["learn_seed"])
    again = [random.sample(range(cfg["N"]), cfg["B"]) for _ in range(cfg["rounds"])]
    assert again == fx["learn_idx"].tolist()


def test_philox_known_answer():
    """Philox4x32-10 KAT from the Random123 distribution (kat_vectors): counter/key all-ones."""
    assert O.philox4x32_10(0, 0, 0, 0, 0, 0) == (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)
    assert O.philox4x32_10(0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff, 0xffffffff,
                           0xffffffff) == (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)
    assert O.philox4x32_10(0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344, 0xa4093822,
                           0x299f31d0) == (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)


def test_philox_sampler_is_a_uniform_subset_sampler():
    idx = O.philox_sample_indices(1000, seed=5, offset=0, B=256)
    assert len(set(idx.tolist())) == 256 and idx.min() >= 0 and idx.max() < 1000
    # dense case forces many collision rounds
    idx = O.philox_sample_indices(64, seed=9, offset=3, B=64)
    assert sorted(idx.tolist()) == list(range(64))
    # marginal uniformity: chi-square over many draws of a small population
    counts = np.zeros(20)
    for off in range(400):
        counts[O.philox_sample_indices(20, seed=1, offset=off, B=5)] += 1
    expected = 400 * 5 / 20
    chi2 = ((counts - expected) ** 2 / expected).sum()
    assert chi2 < 43.8  # 99.9th percentile of chi2(19)


SARSA = ["sarsa_tiny", "sarsa_wrap"]


def load_sarsa(name):
    from conftest import GOLDEN_DIR
    import os
    return torch.load(os.path.join(GOLDEN_DIR, f"{name}.pt"), map_location="cpu", weights_only=False)


def fill_sarsa_oracle(fx):
    rb = O.SarsaReplayOracle(fx["config"]["capacity"])
    A = fx["config"]["A"]
    for p in fx["pushes"]:
        rb.push(p["state"], torch.tensor([p["action"]]), p["reward"], p["terminated"], p["truncated"],
                A, p["next_state"], A, A)
    return rb


@pytest.mark.parametrize("name", SARSA)
def test_sarsa_replay_and_learner_oracle(name):
    """SarsaReplayOracle + DqnOracle(sarsa=True) against the reference's SARSAReplayBuffer +
    DeepSARSA: which pushes end up stored (delayed completion, dropped chain, FIFO wrap), the
    sampled rows incl. next_action bit-exact, Q / next values / targets, the learn() trajectory."""
    fx = load_sarsa(name)
    cfg = fx["config"]
    rb = fill_sarsa_oracle(fx)
    assert len(rb) == fx["stored"]
    raw = rb.sample_at(fx["sample_idx"].tolist())
    for k, want in fx["batch_raw"].items():
        assert torch.equal(raw[k], want), k
    pre = O.preprocess(raw, cfg["A"])
    for k, want in fx["batch_pre"].items():
        assert torch.equal(pre[k], want), k
    pl = O.DqnOracle(fx["params0"], fx["target0"], sarsa=True, tau=0.1)   # deep_td_learning.py:61
    torch.testing.assert_close(pl.q_values(pre["state"], pre["action"]), fx["q"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pl.bellman_target(pre), fx["target"], rtol=1e-5, atol=1e-6)
    random.seed(fx["learn_seed"])
    losses = pl.learn(rb, cfg["rounds"], cfg["B"], cfg["A"])
    torch.testing.assert_close(torch.tensor(losses), fx["learn_losses"], rtol=2e-4, atol=1e-5)
    for k in O.PARAM_KEYS:
        torch.testing.assert_close(pl.p[k], fx["params_after"][k], rtol=1e-3, atol=2e-5, msg=k)
        torch.testing.assert_close(pl.t[k], fx["target_after"][k], rtol=1e-3, atol=2e-5, msg=k)


@pytest.mark.parametrize("variant", ["reward_only", "with_terminated_fn"])
def test_her_replay_oracle(variant):
    """HerReplayOracle against the reference's HindsightExperienceReplayBuffer: every stored row
    (original pushes + goal-relabelled copies at episode ends), bit-exact, oldest first."""
    from helpers import her_reward, her_terminated
    fx = load_sarsa("her_tiny")
    v, cfg = fx["variants"][variant], fx["config"]
    rb = O.HerReplayOracle(cfg["capacity"], cfg["G"], her_reward,
                           her_terminated if variant == "with_terminated_fn" else None)
    for p in v["pushes"]:
        rb.push(p["state"].clone(), torch.tensor([p["action"]]), p["reward"], p["terminated"],
                p["truncated"], cfg["A"], p["next_state"].clone(), cfg["A"], cfg["A"])
    assert len(rb) == v["stored"]
    got = rb.sample_at(range(len(rb)))
    for k, want in v["contents"].items():
        assert torch.equal(got[k].reshape(want.shape), want), k


@pytest.mark.parametrize("name", ["cql_tiny_dynamic", "cql_small"])
def test_conservative_q_learning_oracle(name):
    """DqnOracle(conservative_alpha=2) against the reference's DeepQLearning(is_conservative=True):
    total loss (Bellman MSE + alpha * compute_cql_loss), its gradients and the learn() trajectory.
    Pins the oracle ahead of the HIP implementation (the learner still refuses is_conservative)."""
    fx = load_sarsa(name)
    cfg = fx["config"]
    pl = O.DqnOracle(fx["params0"], fx["target0"], conservative_alpha=2.0)
    b = fx["batch_pre"]
    target = pl.bellman_target(b)
    torch.testing.assert_close(target, fx["target"], rtol=1e-5, atol=1e-6)
    q, g, loss = pl.conservative_gradients(b, fx["target"])
    torch.testing.assert_close(q, fx["q"], rtol=1e-5, atol=1e-6)
    assert abs(loss - float(fx["mse"])) <= 1e-5 * max(1.0, abs(float(fx["mse"])))
    for k, want in fx["grads"].items():
        torch.testing.assert_close(g[k].reshape(want.shape), want, rtol=2e-4, atol=2e-6, msg=k)
    rb = fill_oracle_replay(fx)
    random.seed(fx["learn_seed"])
    losses = pl.learn(rb, cfg["rounds"], cfg["B"], cfg["A"])
    torch.testing.assert_close(torch.tensor(losses), fx["learn_losses"], rtol=2e-4, atol=1e-5)
    for k in O.PARAM_KEYS:
        torch.testing.assert_close(pl.p[k], fx["params_after"][k], rtol=1e-3, atol=2e-5, msg=k)
        torch.testing.assert_close(pl.t[k], fx["target_after"][k], rtol=1e-3, atol=2e-5, msg=k)


@pytest.mark.parametrize("variant", ["plain", "wrap"])
def test_bootstrap_replay_oracle(variant):
    """BootstrapReplayOracle against the reference's BootstrapReplayBuffer: the Bernoulli masks of
    every stored row (same draws from torch's global generator), the sampled batch incl.
    bootstrap_mask, and filter_batch_by_bootstrap_mask — all bit-exact."""
    fx = load_sarsa("bootstrap_tiny")
    cfg, v = fx["config"], fx["variants"][variant]
    rb = O.BootstrapReplayOracle(v["capacity"], cfg["p"], cfg["K"])
    torch.manual_seed(v["mask_seed"])
    for i in range(v["N"]):
        rb.push(v["states"][i], torch.tensor([i % cfg["A"]]), float(i % 7), i % 10 == 9, False,
                cfg["A"], v["states"][i + 1], cfg["A"], cfg["A"])
    assert len(rb) == v["stored"]
    assert torch.equal(torch.cat([r["bootstrap_mask"] for r in rb.memory]), v["masks"])
    random.seed(v["sample_seed"])
    got = rb.sample(cfg["B"])
    for k, want in v["batch"].items():
        if want is None:
            continue
        assert got[k].dtype == want.dtype and torch.equal(got[k], want), k
    filt = O.filter_by_bootstrap_mask(got, 2)
    for k, want in v["filtered_z2"].items():
        assert torch.equal(filt[k], want), k


@pytest.mark.parametrize("learner", ["dqn", "ddqn"])
def test_fullbatch_one_batch_numerics(learner):
    """BASELINE config 2 at its own batch size (B = 1024): the oracle against the reference's
    Q-values, next-state values, Bellman targets (rtol 1e-5) and gradients on the same batch."""
    fx = load_sarsa("dqn_cfg2_fullbatch")
    want = fx["learners"][learner]
    pl = O.DqnOracle(fx["params0"], fx["target0"], double_q=learner == "ddqn")
    b = O.preprocess(fx["batch_raw"], fx["config"]["A"])
    torch.testing.assert_close(pl.q_values(b["state"], b["action"]), want["q"], rtol=1e-5, atol=1e-6)
    nv = pl.next_state_values(b["next_state"], b["next_available_actions"],
                              b["next_unavailable_actions_mask"])
    torch.testing.assert_close(nv, want["next_v"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pl.bellman_target(b), want["target"], rtol=1e-5, atol=1e-6)
    q, g = pl.gradients(b, want["target"])
    torch.testing.assert_close(((q - want["target"]) ** 2).mean(), want["mse"], rtol=1e-5, atol=1e-6)
    for k, w in want["grads"].items():
        torch.testing.assert_close(g[k].reshape(w.shape), w, rtol=2e-4, atol=2e-6, msg=k)


QNETS = ["deep3_tiny", "wide_small", "multihead_tiny", "multihead_double_tiny", "multihead_cfg2_shape",
         "dueling_tiny", "dueling_double_small",
         # mlp_block's other forms (round 5): LayerNorm, leaky_relu / tanh / softplus / sigmoid
         "layernorm_tiny", "layernorm_small", "layernorm_multihead_tiny", "leaky_tiny",
         "tanh_layernorm_small", "softplus_tiny", "sigmoid_tiny",
         # the CQL term beyond the fused shape
         "cql_deep3_tiny", "cql_layernorm_small",
         # ... and on the other QValueNetwork types (round 6)
         "cql_multihead_tiny", "cql_multihead_small", "cql_dueling_tiny", "cql_dueling_small",
         # skip connections / batch norm as network instances (round 6)
         "skip_deep_tiny", "bn_skip_multihead_small"]


def qnet_well_conditioned(fx, key) -> torch.Tensor:
    """Elements of parameter `key` whose trajectory is a function of the algorithm.  In a dueling
    network Q = V + A - mean(A), anything that shifts A(s, .) by the same amount for every action of
    a state cancels exactly: the advantage tower's output bias, and the bias of every hidden unit
    that is active on all rows of a state.  Their true gradient is zero, autograd returns rounding
    noise (~1e-9) of either sign, and AdamW normalises that noise to steps of +-lr — the
    reference itself does not reproduce those elements across BLAS builds.  They are recognised by
    a noise-level gradient on the fixture's first batch and left out of trajectory comparisons."""
    g = fx["grads"][key]
    if fx["config"]["network"] != "dueling":
        return torch.ones_like(g, dtype=torch.bool)
    return g.abs() > 1e-6


@pytest.mark.parametrize("name", QNETS)
def test_qnet_architectures_oracle(name):
    """QNetOracle (other depths / multi-head / dueling, DQN and DoubleDQN rules) against the
    reference: one-batch Q-values, next-state values, targets, gradients, and the learn()
    trajectory with the reference's own index stream."""
    fx = load_sarsa(f"qnet_{name}")
    cfg = fx["config"]
    pl = O.QNetOracle(fx["params0"], fx["target0"], cfg["network"],
                      double_q=cfg.get("learner") == "double",
                      hidden_activation=cfg.get("hidden_activation", "relu"),
                      cql_alpha=2.0 if cfg.get("learner") == "cql" else None)
    b = fx["batch_pre"]
    torch.testing.assert_close(pl.q(pl.p, b["state"], b["action"], b["curr_available_actions"]),
                               fx["q"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pl.next_state_values(b), fx["next_v"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(pl.bellman_target(b), fx["target"], rtol=1e-5, atol=1e-6)
    _, g = pl.gradients(b, fx["target"])
    for k, want in fx["grads"].items():
        torch.testing.assert_close(g[k], want, rtol=2e-4, atol=2e-6, msg=k)
    rb = fill_oracle_replay(fx)
    random.seed(fx["learn_seed"])
    losses = pl.learn(rb, cfg["rounds"], cfg["B"], cfg["A"])
    torch.testing.assert_close(torch.tensor(losses), fx["learn_losses"], rtol=2e-4, atol=1e-5)
    for k in pl.keys:
        ok = qnet_well_conditioned(fx, k)
        torch.testing.assert_close(pl.p[k][ok], fx["params_after"][k][ok], rtol=1e-3, atol=2e-5, msg=k)
        torch.testing.assert_close(pl.t[k][ok], fx["target_after"][k][ok], rtol=1e-3, atol=2e-5, msg=k)
