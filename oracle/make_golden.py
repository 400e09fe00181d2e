#!/usr/bin/env python3
"""Golden-vector generator: runs the REAL reference (facebookresearch/Pearl at
/root/reference, CPU) on seeded synthetic data and writes tests/golden/*.pt.

TEST INFRASTRUCTURE ONLY.  Run in the build container (the reference does not travel to the
GPU box):

    python oracle/make_golden.py            # needs /root/reference; uses oracle/gymstub

Each fixture pins, for one configuration of SURVEY.md §8(d):
  * the replay contract  — what `BasicReplayBuffer.sample()` returns for a known index list
                           (tensor_based_replay_buffer.py:253-400), before and after
                           `preprocess_batch` (policy_learner.py:197-218);
  * the learner numerics — Q(s,a), max_a' Q_target(s',a'), Bellman target, MSE loss, gradients
                           of one batch, and the parameter / optimizer / target-network state
                           after `training_rounds` steps of `DeepQLearning.learn()` together with
                           the per-step reported losses and the index lists it drew.
"""
import os
import random
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("PEARL_REFERENCE", "/root/reference")
sys.path.insert(0, os.path.join(HERE, "gymstub"))
sys.path.insert(0, REF)

import torch  # noqa: E402

from pearl.action_representation_modules.one_hot_action_representation_module import (  # noqa: E402
    OneHotActionTensorRepresentationModule,
)
from pearl.policy_learners.sequential_decision_making.deep_q_learning import DeepQLearning  # noqa: E402
from pearl.policy_learners.sequential_decision_making.deep_sarsa import DeepSARSA  # noqa: E402
from pearl.policy_learners.sequential_decision_making.double_dqn import DoubleDQN  # noqa: E402
from pearl.replay_buffers.sequential_decision_making.sarsa_replay_buffer import (  # noqa: E402
    SARSAReplayBuffer,
)
from pearl.replay_buffers import BasicReplayBuffer  # noqa: E402
from pearl.neural_networks.sequential_decision_making.q_value_networks import (  # noqa: E402
    DuelingQValueNetwork,
    VanillaQValueMultiHeadNetwork,
    VanillaQValueNetwork,
)
from pearl.utils.instantiations.spaces.discrete_action import DiscreteActionSpace  # noqa: E402

OUT = os.path.join(REPO, "tests", "golden")

CONFIGS = {
    # name: S, A, hidden, N (buffer fill), B, rounds, dynamic action spaces
    "tiny": dict(S=6, A=3, hidden=[16, 24], N=40, B=8, rounds=12, dynamic=False),
    "tiny_dynamic": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=11, dynamic=True),
    "cfg1_cartpole_shape": dict(S=4, A=2, hidden=[64, 64], N=600, B=128, rounds=12, dynamic=False),
    "cfg2_shape_small_batch": dict(S=128, A=16, hidden=[256, 256], N=900, B=192, rounds=12,
                                   dynamic=False),
    # DeepQLearning(is_conservative=True) (deep_td_learning.py:323-327): files cql_<name>.pt.  Pins
    # the oracle and the HIP learner's CQL path (tests/test_gpu_dqn.py::test_conservative_q_learning)
    "cql_tiny_dynamic": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=13, dynamic=True,
                             learner="cql"),
    "cql_small": dict(S=16, A=6, hidden=[32, 32], N=300, B=64, rounds=12, dynamic=False,
                      learner="cql"),
    # DoubleDQN (double_dqn.py:29-57): files ddqn_<name>.pt
    "double_tiny_dynamic": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=13, dynamic=True,
                                learner="double"),
    "double_cfg2_shape_small_batch": dict(S=128, A=16, hidden=[256, 256], N=900, B=192, rounds=12,
                                          dynamic=False, learner="double"),
}


# Q-network architectures beyond "VanillaQValueNetwork with two hidden layers" (files qnet_<name>.pt):
# other depths (common/utils.py:75-152), VanillaQValueMultiHeadNetwork (q_value_networks.py:185-249),
# DuelingQValueNetwork (:352-508), each under DeepQLearning and (multi-head, dueling) DoubleDQN.
QNET_CONFIGS = {
    "deep3_tiny": dict(S=5, A=5, hidden=[24, 16, 12], N=48, B=16, rounds=11, dynamic=True,
                       network="vanilla"),
    "wide_small": dict(S=12, A=4, hidden=[320], N=120, B=32, rounds=6, dynamic=False,
                       network="vanilla"),
    "multihead_tiny": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=13, dynamic=True,
                           network="multihead"),
    "multihead_double_tiny": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=13, dynamic=True,
                                  network="multihead", learner="double"),
    "multihead_cfg2_shape": dict(S=128, A=16, hidden=[256, 256], N=900, B=192, rounds=8,
                                 dynamic=False, network="multihead"),
    "dueling_tiny": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=13, dynamic=True,
                         network="dueling"),
    "dueling_double_small": dict(S=16, A=6, hidden=[32, 32], N=300, B=64, rounds=8, dynamic=False,
                                 network="dueling", learner="double"),
    # mlp_block's other forms (common/utils.py:75-152; round 5): nn.LayerNorm between every hidden
    # Linear and its activation (q_value_networks.py:137 use_layer_norm) and the other hidden
    # activations of ActivationType (utils.py:29-56), as `network_instance`s of DeepQLearning / DoubleDQN
    "layernorm_tiny": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=11, dynamic=True,
                           network="vanilla", use_layer_norm=True),
    "layernorm_small": dict(S=16, A=6, hidden=[64, 48], N=300, B=64, rounds=8, dynamic=False,
                            network="vanilla", use_layer_norm=True),
    "layernorm_multihead_tiny": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=11, dynamic=True,
                                     network="multihead", use_layer_norm=True, learner="double"),
    "leaky_tiny": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=11, dynamic=True,
                       network="vanilla", hidden_activation="leaky_relu"),
    "tanh_layernorm_small": dict(S=16, A=6, hidden=[64, 48], N=300, B=64, rounds=8, dynamic=False,
                                 network="vanilla", hidden_activation="tanh", use_layer_norm=True),
    "softplus_tiny": dict(S=5, A=5, hidden=[24, 16, 12], N=48, B=16, rounds=11, dynamic=True,
                          network="vanilla", hidden_activation="softplus"),
    "sigmoid_tiny": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=11, dynamic=False,
                         network="vanilla", hidden_activation="sigmoid"),
    # the CQL term (deep_td_learning.py:323-327, loss_fn_utils.py:17-72) on networks beyond the fused
    # shape: a deeper Vanilla network and one with LayerNorm (generic TD engine, round 5)
    "cql_deep3_tiny": dict(S=5, A=5, hidden=[24, 16, 12], N=48, B=16, rounds=11, dynamic=True,
                           network="vanilla", learner="cql"),
    "cql_layernorm_small": dict(S=16, A=6, hidden=[64, 48], N=300, B=64, rounds=8, dynamic=False,
                                network="vanilla", learner="cql", use_layer_norm=True),
    # ... and on the other QValueNetwork types compute_cql_loss accepts (loss_fn_utils.py:17-72 works
    # on any get_q_values; round 6): multi-head (one forward, the table is a bmm with the one-hot
    # actions) and dueling (the table's advantage mean runs over the queried actions themselves)
    "cql_multihead_tiny": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=11, dynamic=True,
                               network="multihead", learner="cql"),
    "cql_multihead_small": dict(S=16, A=6, hidden=[64, 48], N=300, B=64, rounds=8, dynamic=False,
                                network="multihead", learner="cql"),
    "cql_dueling_tiny": dict(S=5, A=5, hidden=[24, 16], N=48, B=16, rounds=11, dynamic=True,
                             network="dueling", learner="cql"),
    "cql_dueling_small": dict(S=16, A=6, hidden=[32, 32], N=300, B=64, rounds=8, dynamic=False,
                              network="dueling", learner="cql"),
    # mlp_block's skip connections / batch norm (common/utils.py:113-131; round 6) as network_instances:
    # a Vanilla network whose hidden layers are all wrapped (S + A = 10 -> 10 -> 10 -> 10 -> 1), and a
    # multi-head network with BatchNorm1d after every hidden activation plus skip connections (a
    # VanillaQValueNetwork cannot carry BatchNorm1d: its (B, A, S + AD) input is not the layer's)
    "skip_deep_tiny": dict(S=5, A=5, hidden=[10, 10, 10], N=48, B=16, rounds=11, dynamic=True,
                           network="vanilla", mlp=dict(use_skip_connections=True)),
    "bn_skip_multihead_small": dict(S=16, A=6, hidden=[16, 16], N=300, B=64, rounds=8, dynamic=False,
                                    network="multihead", learner="double",
                                    mlp=dict(use_batch_norm=True, use_skip_connections=True)),
}
NETWORK_TYPES = {"vanilla": VanillaQValueNetwork, "multihead": VanillaQValueMultiHeadNetwork,
                 "dueling": DuelingQValueNetwork}


def synthetic_transitions(cfg, gen):
    """states randn(N+1,S); transition i = (S[i], i % A, float(i % 7),
    i % 50 == 0, False, S[i+1])."""
    S, A, N = cfg["S"], cfg["A"], cfg["N"]
    states = torch.randn(N + 1, S, generator=gen)
    rows = []
    for i in range(N):
        n_curr = n_next = A
        if cfg["dynamic"]:
            n_curr = 1 + (i * 7) % A
            n_next = 1 + (i * 3 + 1) % A
        rows.append(dict(i=i, action=i % n_curr, reward=float(i % 7), terminated=(i % 50 == 0),
                         truncated=(i % 13 == 5), n_curr=n_curr, n_next=n_next))
    return states, rows


def space(n):
    return DiscreteActionSpace([torch.tensor([k]) for k in range(n)])


def clone_sd(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def batch_to_dict(batch):
    out = {}
    for k in ("state", "action", "reward", "terminated", "truncated", "next_state",
              "curr_available_actions", "curr_unavailable_actions_mask",
              "next_available_actions", "next_unavailable_actions_mask"):
        v = getattr(batch, k)
        out[k] = None if v is None else v.detach().clone()
    return out


def make(name, cfg):
    torch.manual_seed(0)
    random.seed(0)
    gen = torch.Generator().manual_seed(1234)
    S, A, B = cfg["S"], cfg["A"], cfg["B"]
    states, rows = synthetic_transitions(cfg, gen)

    rep = OneHotActionTensorRepresentationModule(A)
    torch.manual_seed(7)  # the learner's parameter init
    double = cfg.get("learner") == "double"
    cql = cfg.get("learner") == "cql"
    extra = dict(is_conservative=True, conservative_alpha=2.0) if cql else {}
    qnet = "network" in cfg
    if qnet:
        extra["network_type"] = NETWORK_TYPES[cfg["network"]]
    if qnet and cfg.get("mlp"):
        from pearl.neural_networks.common.utils import mlp_block
        multi = cfg["network"] == "multihead"
        net = NETWORK_TYPES[cfg["network"]](state_dim=S, action_dim=A, hidden_dims=cfg["hidden"],
                                            output_dim=A if multi else 1)
        net._model = mlp_block(input_dim=S if multi else S + A, hidden_dims=cfg["hidden"],
                               output_dim=A if multi else 1, **cfg["mlp"])
        extra.pop("network_type")
        extra["network_instance"] = net
    elif qnet and (cfg.get("use_layer_norm") or cfg.get("hidden_activation")):
        # a network_instance in one of mlp_block's other forms: LayerNorm through the network's own
        # use_layer_norm argument, another hidden activation by rebuilding its _model with mlp_block
        from pearl.neural_networks.common.utils import mlp_block
        multi = cfg["network"] == "multihead"
        net = NETWORK_TYPES[cfg["network"]](
            state_dim=S, action_dim=A, hidden_dims=cfg["hidden"], output_dim=A if multi else 1,
            use_layer_norm=bool(cfg.get("use_layer_norm")))
        if cfg.get("hidden_activation"):
            net._model = mlp_block(input_dim=S if multi else S + A, hidden_dims=cfg["hidden"],
                                   output_dim=A if multi else 1,
                                   use_layer_norm=bool(cfg.get("use_layer_norm")),
                                   hidden_activation=cfg["hidden_activation"])
        extra.pop("network_type")
        extra["network_instance"] = net
    pl = (DoubleDQN if double else DeepQLearning)(**extra, state_dim=S, action_space=space(A), hidden_dims=cfg["hidden"],
                       training_rounds=cfg["rounds"], batch_size=B,
                       action_representation_module=rep)
    rb = BasicReplayBuffer(cfg["N"] + 10)
    rb._is_action_continuous = False
    rb.device_for_batches = torch.device("cpu")
    for r in rows:
        rb.push(state=states[r["i"]], action=torch.tensor([r["action"]]), reward=r["reward"],
                terminated=r["terminated"], truncated=r["truncated"],
                curr_available_actions=space(r["n_curr"]), next_state=states[r["i"] + 1],
                next_available_actions=space(r["n_next"]), max_number_actions=A)

    fx = {"config": dict(cfg), "states": states,
          "rows": {k: torch.tensor([r[k] for r in rows]) for k in
                   ("action", "reward", "terminated", "truncated", "n_curr", "n_next")}}

    # ---- replay contract for a known index list
    random.seed(11)
    idx = random.sample(range(len(rb)), B)
    random.seed(11)
    raw = rb.sample(B)
    fx["sample_seed"] = 11
    fx["sample_idx"] = torch.tensor(idx)
    fx["batch_raw"] = batch_to_dict(raw)
    batch = pl.preprocess_batch(raw)
    fx["batch_pre"] = batch_to_dict(batch)

    if qnet:
        # a target network that differs from the online one from the start (otherwise DoubleDQN's
        # argmax and the plain max coincide on the first batches)
        with torch.no_grad():
            for p_ in pl._Q_target.parameters():
                p_.add_(0.1 * torch.randn(p_.shape, generator=gen))
    # ---- one-batch numerics (no parameter change)
    fx["params0"] = clone_sd(pl._Q)
    fx["target0"] = clone_sd(pl._Q_target)
    if qnet:   # exactly what forward() evaluates (deep_td_learning.py:286-290)
        q = pl._Q.get_q_values(state_batch=batch.state, action_batch=batch.action,
                               curr_available_actions_batch=batch.curr_available_actions)
    else:
        q = pl._Q.get_q_values(batch.state, batch.action)
    next_v = pl.get_next_state_values(batch, B)
    loss, target = pl.loss(batch, q)
    pl._optimizer.zero_grad()
    loss.backward()
    fx["q"] = q.detach().clone()
    fx["next_v"] = next_v.detach().clone()
    fx["target"] = target.detach().clone()
    fx["mse"] = loss.detach().clone()
    fx["mean_abs_td"] = (q - target).abs().mean().detach().clone()
    fx["grads"] = {k: p.grad.detach().clone() for k, p in pl._Q.named_parameters()}
    pl._optimizer.zero_grad()

    # ---- learn(): rounds steps with python-sampled indices
    random.seed(23)
    fx["learn_seed"] = 23
    fx["learn_idx"] = torch.tensor([random.sample(range(len(rb)), B) for _ in range(cfg["rounds"])])
    random.seed(23)
    report = pl.learn(rb)
    fx["learn_losses"] = torch.tensor(report["loss"])
    fx["params_after"] = clone_sd(pl._Q)
    fx["target_after"] = clone_sd(pl._Q_target)
    fx["training_steps_after"] = pl._training_steps
    opt_state = {}
    for k, p in pl._Q.named_parameters():
        st = pl._optimizer.state[p]
        opt_state[k] = {n: st[n].detach().clone() for n in
                        ("step", "exp_avg", "exp_avg_sq", "max_exp_avg_sq")}
    fx["opt_after"] = opt_state
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, f"qnet_{name}.pt" if qnet else
                        (f"ddqn_{name[len('double_'):]}.pt" if double
                         else (f"{name}.pt" if cql else f"dqn_{name}.pt")))
    torch.save(fx, path)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); "
          f"losses {report['loss'][0]:.5f} -> {report['loss'][-1]:.5f}")


SARSA_CONFIGS = {
    "sarsa_tiny": dict(S=5, A=3, hidden=[16, 24], N=60, B=12, rounds=12, capacity=200),
    "sarsa_wrap": dict(S=4, A=4, hidden=[24, 16], N=90, B=16, rounds=11, capacity=37),
}


def make_sarsa(name, cfg):
    """DeepSARSA + SARSAReplayBuffer (deep_sarsa.py:30-97, sarsa_replay_buffer.py:22-101): episodes
    of consecutive states (the buffer only completes a transition when the next push continues it),
    one broken chain (dropped transition), terminal and truncated ends; the stored rows incl.
    next_action for a known index list, one-batch numerics, a learn() trajectory."""
    torch.manual_seed(0)
    random.seed(0)
    gen = torch.Generator().manual_seed(4321)
    S, A, B, N = cfg["S"], cfg["A"], cfg["B"], cfg["N"]
    states = torch.randn(N + 1, S, generator=gen)
    rows = []
    for i in range(N):
        end = (i % 11 == 10)
        rows.append(dict(i=i, action=int(torch.randint(0, A, (1,), generator=gen)),
                         reward=float(i % 5) - 1.5, terminated=bool(end and i % 22 == 10),
                         truncated=bool(end and i % 22 != 10),
                         # a gap: the successor of transition 17 starts from another state
                         jump=(i == 18)))
    rep = OneHotActionTensorRepresentationModule(A)
    torch.manual_seed(7)
    pl = DeepSARSA(state_dim=S, action_space=space(A), hidden_dims=cfg["hidden"],
                   training_rounds=cfg["rounds"], batch_size=B, action_representation_module=rep)
    rb = SARSAReplayBuffer(cfg["capacity"])
    rb._is_action_continuous = False
    rb.device_for_batches = torch.device("cpu")
    pushes = []
    for r in rows:
        st = states[r["i"]] + (100.0 if r["jump"] else 0.0)
        pushes.append(dict(state=st.clone(), action=r["action"], reward=r["reward"],
                           terminated=r["terminated"], truncated=r["truncated"],
                           next_state=states[r["i"] + 1].clone()))
        rb.push(state=st, action=torch.tensor([r["action"]]), reward=r["reward"],
                terminated=r["terminated"], truncated=r["truncated"],
                curr_available_actions=space(A), next_state=states[r["i"] + 1],
                next_available_actions=space(A), max_number_actions=A)
    fx = {"config": dict(cfg), "pushes": pushes, "stored": len(rb)}
    random.seed(11)
    idx = random.sample(range(len(rb)), B)
    random.seed(11)
    raw = rb.sample(B)
    fx["sample_seed"], fx["sample_idx"] = 11, torch.tensor(idx)
    fields = ("state", "action", "reward", "terminated", "truncated", "next_state", "next_action",
              "curr_available_actions", "curr_unavailable_actions_mask", "next_available_actions",
              "next_unavailable_actions_mask")
    fx["batch_raw"] = {k: getattr(raw, k).detach().clone() for k in fields}
    batch = pl.preprocess_batch(raw)
    fx["batch_pre"] = {k: getattr(batch, k).detach().clone() for k in fields}
    fx["params0"], fx["target0"] = clone_sd(pl._Q), clone_sd(pl._Q_target)
    q = pl._Q.get_q_values(batch.state, batch.action)
    next_v = pl.get_next_state_values(batch, B)
    loss, target = pl.loss(batch, q)
    fx["q"], fx["next_v"], fx["target"] = q.detach().clone(), next_v.detach().clone(), target.detach().clone()
    random.seed(23)
    fx["learn_seed"] = 23
    report = pl.learn(rb)
    fx["learn_losses"] = torch.tensor(report["loss"])
    fx["params_after"], fx["target_after"] = clone_sd(pl._Q), clone_sd(pl._Q_target)
    fx["training_steps_after"] = pl._training_steps
    path = os.path.join(OUT, f"{name}.pt")
    torch.save(fx, path)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); stored {len(rb)} of {N} "
          f"pushes; losses {report['loss'][0]:.5f} -> {report['loss'][-1]:.5f}")


def her_reward(state, action):
    """Deterministic sparse reward of the HER fixture: 0 when the state part is close to the goal
    slot, else -1 (any pure function of (state, action) serves the fixture)."""
    g = state.shape[0] // 2
    return 0.0 if float((state[:g] - state[g:]).abs().sum()) < 1.5 else -1.0


def her_terminated(state, action):
    g = state.shape[0] // 2
    return bool(float((state[:g] - state[g:]).abs().sum()) < 0.75)


def make_her(name="her_tiny"):
    """HindsightExperienceReplayBuffer (hindsight_experience_replay_buffer.py:19-160): three
    episodes (terminated, truncated, unfinished) of states [observation | goal] with goal_dim =
    observation dim; everything the buffer holds afterwards, oldest first."""
    from pearl.replay_buffers.sequential_decision_making.hindsight_experience_replay_buffer import (
        HindsightExperienceReplayBuffer,
    )
    G, A, cap = 3, 3, 64
    gen = torch.Generator().manual_seed(77)
    fx = {"config": dict(G=G, A=A, capacity=cap), "variants": {}}
    for variant, term_fn in (("reward_only", None), ("with_terminated_fn", her_terminated)):
        rb = HindsightExperienceReplayBuffer(cap, G, her_reward, term_fn)
        rb._is_action_continuous = False
        rb.device_for_batches = torch.device("cpu")
        pushes = []
        lengths, ends = (5, 4, 3), ("terminated", "truncated", None)
        gen.manual_seed(77)
        for L, end in zip(lengths, ends):
            goal = torch.randn(G, generator=gen)
            obs = torch.randn(L + 1, G, generator=gen)
            for t in range(L):
                last = t == L - 1
                p = dict(state=torch.cat([obs[t], goal]), action=int(torch.randint(0, A, (1,), generator=gen)),
                         reward=-1.0, terminated=bool(last and end == "terminated"),
                         truncated=bool(last and end == "truncated"),
                         next_state=torch.cat([obs[t + 1], goal]))
                pushes.append({k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in p.items()})
                rb.push(state=p["state"], action=torch.tensor([p["action"]]), reward=p["reward"],
                        terminated=p["terminated"], truncated=p["truncated"],
                        curr_available_actions=space(A), next_state=p["next_state"],
                        next_available_actions=space(A), max_number_actions=A)
        rows = list(rb.memory)
        fields = ("state", "action", "reward", "terminated", "truncated", "next_state")
        fx["variants"][variant] = dict(
            pushes=pushes, stored=len(rb),
            contents={k: torch.cat([getattr(r, k) for r in rows]) for k in fields})
    path = os.path.join(OUT, f"{name}.pt")
    torch.save(fx, path)
    print(f"{name}: wrote {path}; stored {[v['stored'] for v in fx['variants'].values()]} rows")



def make_bootstrap(name="bootstrap_tiny"):
    """BootstrapReplayBuffer (bootstrap_replay_buffer.py:23-114): the Bernoulli masks every push
    draws from torch's global generator, the buffer contents after a FIFO wrap, and one sampled
    `TransitionWithBootstrapMaskBatch` for a known index list; two variants (no wrap / wrap)."""
    from pearl.replay_buffers.sequential_decision_making.bootstrap_replay_buffer import (
        BootstrapReplayBuffer,
    )
    from pearl.replay_buffers.transition import filter_batch_by_bootstrap_mask
    S, A, K, p, B = 5, 4, 6, 0.6, 16
    fx = {"config": dict(S=S, A=A, K=K, p=p, B=B), "variants": {}}
    for variant, N, cap in (("plain", 40, 64), ("wrap", 90, 37)):
        gen = torch.Generator().manual_seed(2024)
        states = torch.randn(N + 1, S, generator=gen)
        rb = BootstrapReplayBuffer(cap, p, K)
        rb._is_action_continuous = False
        rb.device_for_batches = torch.device("cpu")
        torch.manual_seed(99)              # the masks come from the global generator
        for i in range(N):
            rb.push(state=states[i], action=torch.tensor([i % A]), reward=float(i % 7),
                    terminated=(i % 10 == 9), truncated=False,
                    curr_available_actions=space(A), next_state=states[i + 1],
                    next_available_actions=space(A), max_number_actions=A)
        masks = torch.cat([t.bootstrap_mask for t in rb.memory])
        random.seed(11)
        idx = random.sample(range(len(rb)), B)
        random.seed(11)
        batch = rb.sample(B)
        d = batch_to_dict(batch)
        d["bootstrap_mask"] = batch.bootstrap_mask.detach().clone()
        filt = filter_batch_by_bootstrap_mask(batch, torch.tensor(2))
        fx["variants"][variant] = dict(
            N=N, capacity=cap, states=states, mask_seed=99, stored=len(rb), masks=masks,
            sample_seed=11, sample_idx=torch.tensor(idx), batch=d,
            filtered_z2={k: (None if getattr(filt, k) is None else getattr(filt, k).detach().clone())
                         for k in ("state", "action", "reward", "terminated", "next_state")})
    path = os.path.join(OUT, f"{name}.pt")
    torch.save(fx, path)
    print(f"{name}: wrote {path}; stored {[v['stored'] for v in fx['variants'].values()]} rows, "
          f"mask mean {float(fx['variants']['plain']['masks'].mean()):.3f}")


def make_fullbatch(name="dqn_cfg2_fullbatch"):
    """BASELINE config 2 AT ITS OWN BATCH SIZE (S=128, A=16, hidden [256,256], B=1024): the one-batch
    quantities of DeepQLearning and DoubleDQN — Q(s,a), next-state values, Bellman targets, MSE,
    mean |TD|, gradients — on one sampled batch, with a target network that differs from the
    online one (so that DoubleDQN's argmax and DQN's max disagree).  Only the sampled batch is
    stored (not the replay contents): the replay contract is pinned by the other fixtures."""
    S, A, B, N, hidden = 128, 16, 1024, 4096, [256, 256]
    torch.manual_seed(0)
    random.seed(0)
    gen = torch.Generator().manual_seed(1234)
    cfg = dict(S=S, A=A, hidden=hidden, N=N, B=B, rounds=1, dynamic=False)
    states, rows = synthetic_transitions(cfg, gen)
    rep = OneHotActionTensorRepresentationModule(A)
    rb = BasicReplayBuffer(N + 10)
    rb._is_action_continuous = False
    rb.device_for_batches = torch.device("cpu")
    for r in rows:
        rb.push(state=states[r["i"]], action=torch.tensor([r["action"]]), reward=r["reward"],
                terminated=r["terminated"], truncated=r["truncated"],
                curr_available_actions=space(A), next_state=states[r["i"] + 1],
                next_available_actions=space(A), max_number_actions=A)
    random.seed(11)
    raw = rb.sample(B)
    fx = {"config": cfg, "batch_raw": batch_to_dict(raw), "learners": {}}
    for key, cls in (("dqn", DeepQLearning), ("ddqn", DoubleDQN)):
        torch.manual_seed(7)
        pl = cls(state_dim=S, action_space=space(A), hidden_dims=hidden, training_rounds=1,
                 batch_size=B, action_representation_module=rep)
        torch.manual_seed(8)     # a second, independent initialisation for the target network
        other = cls(state_dim=S, action_space=space(A), hidden_dims=hidden, training_rounds=1,
                    batch_size=B, action_representation_module=rep)
        pl._Q_target.load_state_dict(other._Q.state_dict())
        import copy
        batch = pl.preprocess_batch(copy.deepcopy(raw))
        q = pl._Q.get_q_values(batch.state, batch.action)
        next_v = pl.get_next_state_values(batch, B)
        loss, target = pl.loss(batch, q)
        pl._optimizer.zero_grad()
        loss.backward()
        d = dict(q=q.detach().clone(), next_v=next_v.detach().clone(),
                 target=target.detach().clone(), mse=loss.detach().clone(),
                 mean_abs_td=(q - target).abs().mean().detach().clone(),
                 grads={k: p.grad.detach().clone() for k, p in pl._Q.named_parameters()})
        if key == "dqn":
            fx["params0"], fx["target0"] = clone_sd(pl._Q), clone_sd(pl._Q_target)
        else:   # same seeds -> same parameters: stored once
            for k2, v in clone_sd(pl._Q).items():
                assert torch.equal(v, fx["params0"][k2])
        fx["learners"][key] = d
    assert not torch.equal(fx["learners"]["dqn"]["next_v"], fx["learners"]["ddqn"]["next_v"])
    path = os.path.join(OUT, f"{name}.pt")
    torch.save(fx, path)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB); mse "
          f"{float(fx['learners']['dqn']['mse']):.5f} / {float(fx['learners']['ddqn']['mse']):.5f}")


def main():
    only = sys.argv[1:]          # optional: names of the configurations to (re)generate
    if not only or "her_tiny" in only:
        make_her()
    if not only or "bootstrap_tiny" in only:
        make_bootstrap()
    if not only or "dqn_cfg2_fullbatch" in only:
        make_fullbatch()
    for name, cfg in SARSA_CONFIGS.items():
        if not only or name in only:
            make_sarsa(name, cfg)
    for name, cfg in CONFIGS.items():
        if not only or name in only:
            make(name, cfg)
    for name, cfg in QNET_CONFIGS.items():
        if not only or name in only or "qnet" in only:
            make(name, cfg)


if __name__ == "__main__":
    main()
