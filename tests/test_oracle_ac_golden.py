"""CPU: the actor-critic oracle (oracle/actor_critic_oracle.py) against fixtures minted by the real
reference (oracle/make_golden_ac.py).  This is what "parity pinned" means for PPO and SAC."""
import os

import pytest
import torch

from conftest import GOLDEN_DIR
from oracle.actor_critic_oracle import PpoOracle, SacOracle

PPO = ["tiny", "eps0", "cfg4_shape_small", "cfg4_fullbatch"]
SAC = ["tiny", "cfg3_shape_small", "cfg3_fullbatch"]


def load(kind, name):
    return torch.load(os.path.join(GOLDEN_DIR, f"{kind}_{name}.pt"), map_location="cpu",
                      weights_only=False)


def onehot(actions, A):
    return torch.nn.functional.one_hot(actions.long(), A).float()


@pytest.mark.parametrize("name", PPO)
def test_ppo_preprocess_gae_lambda_return_action_probs(name):
    fx = load("ppo", name)
    cfg = fx["config"]
    orc = PpoOracle(fx["actor0"], fx["critic0"], cfg["A"], epsilon=cfg["epsilon"])
    N = cfg["N"]
    gae, ret, ap = orc.preprocess(fx["states"][:N], onehot(fx["actions"], cfg["A"]), fx["rewards"],
                                  fx["terminated"], fx["truncated"], fx["states"][N])
    torch.testing.assert_close(gae, fx["gae"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ret, fx["lam_return"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ap, fx["action_probs"].view(-1), rtol=1e-6, atol=1e-7)


def test_ppo_gae_known_answer():
    """The reference's own KAT (test/unit/with_pytorch/test_ppo.py:48-115): gamma 0.6, lambda 0.5,
    rewards 4, 6, 5, no terminal: gae_2 = td_2, gae_1 = td_1 + 0.3 gae_2, gae_0 = td_0 + 0.3 gae_1."""
    fx = load("ppo", "tiny")
    cfg = fx["config"]
    orc = PpoOracle(fx["actor0"], fx["critic0"], cfg["A"], gamma=0.6, lam=0.5)
    states = fx["states"][:4]
    with torch.no_grad():
        from oracle.actor_critic_oracle import mlp
        v = mlp(orc.critic, states).view(-1)
    r = torch.tensor([4.0, 6.0, 5.0])
    f = torch.zeros(3, dtype=torch.bool)
    gae, ret, _ = orc.preprocess(states[:3], onehot(torch.tensor([0, 1, 2]), cfg["A"]), r, f, f,
                                 states[3])
    td = [r[i] + 0.6 * v[i + 1] - v[i] for i in range(3)]
    want2 = td[2]
    want1 = td[1] + 0.3 * want2
    want0 = td[0] + 0.3 * want1
    torch.testing.assert_close(gae, torch.stack([want0, want1, want2]), rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ret, gae + v[:3], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("name", PPO)
def test_ppo_learn_trajectory(name):
    fx = load("ppo", name)
    cfg = fx["config"]
    A, N = cfg["A"], cfg["N"]
    orc = PpoOracle(fx["actor0"], fx["critic0"], A, epsilon=cfg["epsilon"])
    oh = onehot(fx["actions"], A)
    gae, ret, ap = orc.preprocess(fx["states"][:N], oh, fx["rewards"], fx["terminated"],
                                  fx["truncated"], fx["states"][N])
    la, lc = [], []
    for idx in fx["learn_idx"]:
        a, c = orc.learn_batch(fx["states"][:N][idx], oh[idx], ap[idx], gae[idx], ret[idx])
        la.append(a)
        lc.append(c)
    torch.testing.assert_close(torch.tensor(la), fx["actor_losses"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(torch.tensor(lc), fx["critic_losses"], rtol=1e-5, atol=1e-6)
    for i, (w, b) in enumerate(orc.actor):
        torch.testing.assert_close(w.detach(), fx["actor_after"][f"_model.{i}.0.weight"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(b.detach(), fx["actor_after"][f"_model.{i}.0.bias"], rtol=1e-5, atol=1e-6)
    for i, (w, b) in enumerate(orc.critic):
        torch.testing.assert_close(w.detach(), fx["critic_after"][f"_model.{i}.0.weight"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", SAC)
def test_sac_probe_and_trajectory(name):
    fx = load("sac", name)
    orc = SacOracle(fx["actor0"], fx["critic0"], fx["critic_target0"], fx["low"], fx["high"])
    b = fx["batch"]
    with torch.no_grad():
        act, logp = orc.sample_action(b["state"], fx["probe"]["noise"])
        q1 = orc.q(orc.c[0], b["state"], act)
        q2 = orc.q(orc.c[1], b["state"], act)
    torch.testing.assert_close(act, fx["probe"]["action"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(logp.view(-1), fx["probe"]["log_prob"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(q1, fx["probe"]["q1"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(q2, fx["probe"]["q2"], rtol=1e-5, atol=1e-6)
    for (na, nc), want in zip(fx["noises"], fx["reports"]):
        got = orc.learn_batch(b, na, nc)
        for k in want:
            assert abs(got[k] - want[k]) <= 2e-5 * max(1.0, abs(want[k])), (k, got[k], want[k])
    torch.testing.assert_close(orc.log_alpha.detach(), fx["log_entropy_after"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(orc.head[0].detach(), fx["actor_after"]["fc_mu.weight"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(orc.c[0][0][0].detach(),
                               fx["critic_after"]["_critic_1._model.0.0.weight"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(orc.ct[1][0][0],
                               fx["critic_target_after"]["_critic_2._model.0.0.weight"], rtol=1e-5, atol=1e-7)


BANDIT = ["tiny", "cfg5_shape_small", "cfg5_fullbatch", "mae_tiny", "bce_tiny", "mse_sigmoid_tiny",
          "mae_cfg5_shape_small", "bce_cfg5_shape_small",
          # mlp_block's other forms in the trunk (round 5): LayerNorm, leaky_relu, tanh
          "layernorm_tiny", "leaky_layernorm_small", "tanh_tiny",
          # force_pinv=True (the pseudo-inverse of the regularised, SPD matrix is its inverse)
          "pinv_tiny",
          # nn_e2e=False: mu from the regression's coefficients, the trunk learns through them
          "lin_head_tiny", "lin_head_small", "lin_head_sigmoid_tiny",
          # force_pinv on an unregularised regression with fewer contexts than coefficients (singular A)
          "pinv_singular_tiny", "pinv_singular_small",
          # mlp_block's remaining options in the trunk (round 6): batch norm, dropout (recorded masks),
          # skip connections
          "bn_tiny", "dropout_tiny", "skip_tiny", "bn_ln_dropout_skip_small", "bn_cfg5_shape"]


def bandit_batches(fx):
    """The (state, reward, weight) batches of a bandit fixture; seeded fixtures store a checksum of
    the contexts instead of the contexts (oracle/fixture_inputs.py)."""
    from oracle import fixture_inputs as FI
    for k, b in enumerate(fx["batches"]):
        if "state" in b:
            yield b["state"], b["reward"], b["weight"]
        else:
            x = FI.bandit_contexts(fx["config"], k)
            assert FI.checksum(x) == b["state_checksum"], "regenerated contexts differ from the minted ones"
            yield x, b["reward"], b["weight"]


@pytest.mark.parametrize("name", BANDIT)
def test_neural_linear_bandit_trajectory(name):
    from oracle.actor_critic_oracle import NeuralLinearOracle
    fx = load("bandit", name)
    cfg = fx["config"]
    orc = NeuralLinearOracle(fx["model0"], lr=1e-3, loss_type=cfg.get("loss", "mse"),
                             output_activation=cfg.get("out", "linear"),
                             hidden_activation=cfg.get("mlp", {}).get("hidden_activation", "relu"),
                             nn_e2e=cfg.get("mlp", {}).get("nn_e2e", True),
                             l2_reg_lambda=cfg.get("mlp", {}).get("l2_reg_lambda_linear", 1.0),
                             force_pinv=cfg.get("mlp", {}).get("force_pinv", False),
                             use_batch_norm=cfg.get("mlp", {}).get("use_batch_norm", False),
                             dropout_ratio=cfg.get("mlp", {}).get("dropout_ratio", 0.0),
                             use_skip_connections=cfg.get("mlp", {}).get("use_skip_connections", False))
    for step, ((x, r, w), want) in enumerate(zip(bandit_batches(fx), fx["reports"])):
        if "drop_masks" in fx:
            orc.masks = [m.clone() for m in fx["drop_masks"][step]]      # the reference's draws
        got = orc.learn_batch(x, r, w)
        assert abs(float(got["loss"]) - want["loss"]) <= 1e-5 * max(1.0, abs(want["loss"]))
        torch.testing.assert_close(got["prediction"], want["prediction"], rtol=1e-5, atol=1e-6)
    after = fx["model_after"]
    torch.testing.assert_close(orc.A, after["_linear_regression_layer._A"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(orc.b, after["_linear_regression_layer._b"], rtol=1e-5, atol=1e-5)
    # (the oracle inverts in fp32 like the reference: its own backward error is the method's)
    from helpers import assert_linear_solve_close
    if orc.lam > 0:
        assert_linear_solve_close(orc.coefs, orc.A, orc.b, orc.lam, after["_linear_regression_layer._coefs"],
                                  max_backward=5e-5, msg=name)
    else:
        # a singular A has no backward-error yardstick, and an fp32 eigh-based pseudo-inverse is only
        # as stable as the gap between the kept eigenvalues and torch's cut-off (the reference's own
        # result is 0.5 % / 15 % from the exact pseudo-inverse of its matrix on these two fixtures).
        # The oracle performs the reference's operations on the same numbers: it must reproduce them.
        torch.testing.assert_close(orc.inv_A, after["_linear_regression_layer._inv_A"], rtol=1e-4,
                                   atol=1e-4 * float(after["_linear_regression_layer._inv_A"].abs().max()))
        torch.testing.assert_close(orc.coefs, after["_linear_regression_layer._coefs"], rtol=1e-4,
                                   atol=1e-4 * float(after["_linear_regression_layer._coefs"].abs().max()))
    train_query = not fx.get("query_eval", False)     # (batch-norm / dropout fixtures query in eval mode)
    torch.testing.assert_close(orc.sigma(fx["query"]["x"], train_query), fx["query"]["sigma"].view(-1),
                               rtol=2e-4, atol=1e-6)
    bases = getattr(orc, "_bases", None) or [f"_nn_layers._model.{i}." for i in range(len(orc.trunk))]
    for i, (w_, b_) in enumerate(orc.trunk):
        torch.testing.assert_close(w_.detach(), after[f"{bases[i]}0.weight"], rtol=1e-4, atol=1e-6)
    for i, n in enumerate(orc.norms):
        if n is not None:
            # (the oracle's LayerNorm is the written-out formula, the reference's F.layer_norm: one
            #  noise-level element in 40 ends 1.6e-4 apart after three AdamW steps on the combined fixture)
            tol = dict(rtol=1e-3, atol=1e-5) if name == "bn_ln_dropout_skip_small" else dict(rtol=1e-4, atol=1e-6)
            torch.testing.assert_close(n[0].detach(), after[f"{bases[i]}1.weight"], **tol)
            torch.testing.assert_close(n[1].detach(), after[f"{bases[i]}1.bias"], **tol)
    for bn in (orc.bn or []):
        if bn is not None:       # weight / bias AND the running statistics every training forward moved
            for name, key in (("weight", "weight"), ("bias", "bias"), ("running_mean", "running_mean"),
                              ("running_var", "running_var"), ("nbt", "num_batches_tracked")):
                torch.testing.assert_close(bn[name].detach(), after[bn["key"] + key], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(orc.e2e.detach(), after["linear_layer_e2e.weight"], rtol=1e-4, atol=1e-6)


def test_ppo_rollout64k_preprocess():
    """BASELINE config 4's rollout size (65 536 transitions): GAE / lambda returns / action
    probabilities of the reference's preprocess_replay_buffer (ppo.py:211-293) — episodes of 97
    transitions, truncations every 131st, the bootstrap from the state after the last transition."""
    from oracle import fixture_inputs as FI
    fx = load("ppo", "cfg4_rollout64k")
    cfg = fx["config"]
    states, actions, rewards, term, trunc = FI.ppo_rollout(cfg)
    assert FI.checksum(states) == fx["checksums"]["states"]
    assert FI.checksum(actions) == fx["checksums"]["actions"]
    assert FI.checksum(rewards) == fx["checksums"]["rewards"]
    N = cfg["N"]
    orc = PpoOracle(fx["actor0"], fx["critic0"], cfg["A"], epsilon=cfg["epsilon"])
    gae, ret, ap = orc.preprocess(states[:N], onehot(actions, cfg["A"]), rewards, term, trunc, states[N])
    torch.testing.assert_close(gae, fx["gae"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ret, fx["lam_return"], rtol=1e-6, atol=1e-6)
    torch.testing.assert_close(ap, fx["action_probs"], rtol=1e-6, atol=1e-7)


DDPG = ["ddpg_tiny", "ddpg_cfg3_shape_small", "td3_tiny", "td3_cfg3_shape_small", "td3_cfg3_fullbatch"]


@pytest.mark.parametrize("name", DDPG)
def test_ddpg_td3_probe_and_trajectory(name):
    """DdpgOracle against the reference's DDPG / TD3 runs: deterministic actions, Q-values, the
    per-call losses (TD3: delayed actor, repeated last actor loss, smoothing noise replayed) and
    all four networks afterwards."""
    from oracle.actor_critic_oracle import DdpgOracle
    fx = torch.load(os.path.join(GOLDEN_DIR, f"{name}.pt"), map_location="cpu", weights_only=False)
    orc = DdpgOracle(fx["actor0"], fx["actor_target0"], fx["critic0"], fx["critic_target0"],
                     fx["low"], fx["high"], td3=fx["config"]["td3"])
    b = fx["batch"]
    with torch.no_grad():
        act = orc.policy(orc.actor, b["state"])
        torch.testing.assert_close(act, fx["probe"]["action"], rtol=1e-6, atol=1e-6)
        torch.testing.assert_close(orc.q(orc.c[0], b["state"], act), fx["probe"]["q1"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(orc.q(orc.c[1], b["state"], act), fx["probe"]["q2"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(orc.policy(orc.actor_t, b["next_state"]), fx["probe"]["next_action"],
                                   rtol=1e-6, atol=1e-6)
    for k, (noise, want) in enumerate(zip(fx["noises"], fx["reports"])):
        orc.training_steps = k
        got = orc.learn_batch(b, noise)
        for key in want:
            assert abs(got[key] - want[key]) <= 2e-5 * max(1.0, abs(want[key])), (k, key, got[key], want[key])
    for i, (w, bias) in enumerate(orc.actor):
        torch.testing.assert_close(w.detach(), fx["actor_after"][f"_model.{i}.0.weight"], rtol=1e-4, atol=1e-6)
    for i, (w, bias) in enumerate(orc.actor_t):
        torch.testing.assert_close(w, fx["actor_target_after"][f"_model.{i}.0.weight"], rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(orc.c[1][0][0].detach(),
                               fx["critic_after"]["_critic_2._model.0.0.weight"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(orc.ct[0][1][1],
                               fx["critic_target_after"]["_critic_1._model.1.0.bias"], rtol=1e-5, atol=1e-7)


def preprocessed_dsac_batch(fx):
    """PolicyLearner.preprocess_batch (policy_learner.py:197-218) of the stored raw batch."""
    A = fx["config"]["A"]
    b = {k: v.clone() for k, v in fx["batch"].items()}
    b["action"] = onehot(b["action"].view(-1), A)
    for k in ("curr_available_actions", "next_available_actions"):
        b[k] = onehot(b[k].squeeze(-1), A)
    return b


@pytest.mark.parametrize("name", ["dsac_tiny", "dsac_shape_small", "dsac_cfg2_fullbatch"])
def test_discrete_sac_probe_and_trajectory(name):
    from oracle.actor_critic_oracle import DiscreteSacOracle
    fx = torch.load(os.path.join(GOLDEN_DIR, f"{name}.pt"), map_location="cpu", weights_only=False)
    orc = DiscreteSacOracle(fx["actor0"], fx["critic0"], fx["critic_target0"], fx["config"]["A"])
    b = preprocessed_dsac_batch(fx)
    with torch.no_grad():
        torch.testing.assert_close(orc.policy(b["state"]), fx["probe"]["policy"], rtol=1e-5, atol=1e-7)
        torch.testing.assert_close(orc.q_all(orc.c[0], b["state"], b["curr_available_actions"]),
                                   fx["probe"]["q1"], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(orc.q_all(orc.c[1], b["state"], b["curr_available_actions"]),
                                   fx["probe"]["q2"], rtol=1e-5, atol=1e-6)
    for k, want in enumerate(fx["reports"]):
        got = orc.learn_batch(b)
        for key in want:
            assert abs(got[key] - want[key]) <= 2e-5 * max(1.0, abs(want[key])), (k, key, got[key], want[key])
    torch.testing.assert_close(orc.log_alpha.detach(), fx["log_entropy_after"], rtol=1e-5, atol=1e-7)
    for i, (w, _) in enumerate(orc.actor):
        torch.testing.assert_close(w.detach(), fx["actor_after"][f"_model.{i}.0.weight"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(orc.c[0][0][0].detach(),
                               fx["critic_after"]["_critic_1._model.0.0.weight"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(orc.ct[1][0][0],
                               fx["critic_target_after"]["_critic_2._model.0.0.weight"], rtol=1e-5, atol=1e-7)


IQL = ["iql_continuous_tiny", "iql_continuous_shape_small", "iql_discrete_tiny", "iql_gaussian_tiny",
       "iql_gaussian_shape_small", "iql_continuous_fullbatch"]


def iql_batch(fx):
    b = {k: v.clone() for k, v in fx["batch"].items()}
    if not fx["config"]["continuous"]:
        b["action"] = onehot(b["action"].view(-1), fx["config"]["A"])      # preprocess_batch
    return b


@pytest.mark.parametrize("name", IQL)
def test_iql_trajectory(name):
    """IqlOracle against the reference's ImplicitQLearning: the three losses per call (the two
    target-critic draws replayed by seeding torch like the generator did) and all networks."""
    from oracle.actor_critic_oracle import IqlOracle
    fx = torch.load(os.path.join(GOLDEN_DIR, f"{name}.pt"), map_location="cpu", weights_only=False)
    cfg = fx["config"]
    orc = IqlOracle(fx["actor0"], fx["value0"], fx["critic0"], fx["critic_target0"],
                    cfg["continuous"], fx["low"], fx["high"], expectile=cfg["expectile"])
    b = iql_batch(fx)
    for k, want in enumerate(fx["reports"]):
        torch.manual_seed(4000 + k)
        got = orc.learn_batch(b)
        for key in want:
            assert abs(got[key] - want[key]) <= 2e-5 * max(1.0, abs(want[key])), (k, key, got[key], want[key])
    for i, (w, _) in enumerate(orc.actor):
        torch.testing.assert_close(w.detach(), fx["actor_after"][f"_model.{i}.0.weight"], rtol=1e-4, atol=1e-6)
    for i, (w, _) in enumerate(orc.value):
        torch.testing.assert_close(w.detach(), fx["value_after"][f"_model.{i}.0.weight"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(orc.c[1][0][0].detach(),
                               fx["critic_after"]["_critic_2._model.0.0.weight"], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(orc.ct[0][0][0],
                               fx["critic_target_after"]["_critic_1._model.0.0.weight"], rtol=1e-5, atol=1e-7)


def test_squarecb_oracle():
    """squarecb_probs against the table the reference's SquareCBExploration.act hands to
    Categorical (captured by oracle/make_golden_ac.py::make_squarecb), and the seeded draw."""
    from oracle.actor_critic_oracle import squarecb_probs
    fx = torch.load(os.path.join(GOLDEN_DIR, "squarecb_tiny.pt"), map_location="cpu", weights_only=False)
    for c in fx["cases"]:
        p = squarecb_probs(c["values"].clone(), c["gamma"], c["clamp"], 0.2, 0.7)
        assert torch.equal(p, c["probs"].view(1, -1))
        torch.manual_seed(c["seed"])
        assert int(torch.distributions.Categorical(p[0]).sample()) == c["action"]


def test_squarecb_host_rule_matches_reference():
    """pearl_amd.SquareCBExploration on CPU values (the host-side statement of the rule the kernel
    implements): the reference's table and the reference's seeded action."""
    from pearl_amd import DiscreteActionSpace, SquareCBExploration
    fx = torch.load(os.path.join(GOLDEN_DIR, "squarecb_tiny.pt"), map_location="cpu", weights_only=False)
    for c in fx["cases"]:
        sp = DiscreteActionSpace([torch.tensor([k]) for k in range(c["A"])])
        e = SquareCBExploration(c["gamma"], reward_lb=0.2, reward_ub=0.7, clamp_values=c["clamp"])
        torch.testing.assert_close(e.probabilities(c["values"].clone(), c["A"]), c["probs"].view(1, -1),
                                   rtol=1e-6, atol=1e-7)
        torch.manual_seed(c["seed"])
        assert int(e.act(None, sp, values=c["values"].clone())) == c["action"]
    # batches of contexts: every row is the single-context rule of that row
    e = SquareCBExploration(5.0)
    v = torch.rand(6, 4)
    tab = e.probabilities(v, 4)
    for i in range(6):
        torch.testing.assert_close(tab[i:i + 1], e.probabilities(v[i:i + 1], 4))
    assert torch.all(tab >= 0) and torch.allclose(tab.sum(1), torch.ones(6))


def test_squarecb_cfg5_act_and_scores():
    """BASELINE config 5's act path on the oracle: 32 arms with arm features appended to 512-dim
    contexts, NeuralLinearBandit [256, 64] values -> the SquareCB table the reference handed to
    Categorical, the seeded draw, and get_scores (neural_linear_bandit.py:227-311)."""
    from oracle.actor_critic_oracle import NeuralLinearOracle, squarecb_probs
    fx = torch.load(os.path.join(GOLDEN_DIR, "squarecb_cfg5.pt"), map_location="cpu", weights_only=False)
    orc = NeuralLinearOracle(fx["model0"])
    A = fx["A"]
    for c in fx["cases"]:
        feats = torch.cat([c["state"].view(1, -1).expand(A, -1), fx["arms"]], dim=1)
        with torch.no_grad():
            values = torch.nn.functional.linear(orc.features(feats), orc.e2e).view(1, A)
        torch.testing.assert_close(values.view(-1), c["scores"].view(-1), rtol=1e-6, atol=1e-7)
        p = squarecb_probs(values.clone(), fx["gamma"])
        torch.testing.assert_close(p, c["probs"].view(1, -1), rtol=1e-5, atol=1e-8)
        assert abs(float(p.sum()) - 1.0) < 1e-5 and float(p.min()) >= 0.0
        torch.manual_seed(c["seed"])
        assert int(torch.distributions.Categorical(c["probs"].view(-1)).sample()) == c["action"]
