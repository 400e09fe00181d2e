"""CPU restatement of the reference's replay + DQN learner hot path.

TEST INFRASTRUCTURE ONLY — the checker, never the product.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this file; pearl_amd/ must not (and does not).

It restates, in plain PyTorch fp32 on the CPU, what these reference functions compute
(file:line under /root/reference), with the same data structures on the replay side (a deque of
per-transition one-row tensors, `random.sample`, `torch.cat` collation) so that it is also a fair
CPU baseline of the reference's cost structure:

  ReplayOracle.push            pearl/replay_buffers/tensor_based_replay_buffer.py:55-133,
                               :143-177, :179-251; basic_replay_buffer.py:21-48
  ReplayOracle.sample          tensor_based_replay_buffer.py:253-282 (random.sample on the deque)
  ReplayOracle.collate         tensor_based_replay_buffer.py:290-400
  BootstrapReplayOracle        sequential_decision_making/bootstrap_replay_buffer.py:23-114,
                               transition.py:242-301
  one_hot / preprocess         action_representation_modules/one_hot_action_representation_module
                               .py:27-34; policy_learners/policy_learner.py:197-218
  DqnOracle.q_values           neural_networks/sequential_decision_making/q_value_networks.py:152-174
                               (+ extend_state_feature.py:12-47, common/utils.py:75-152)
  DqnOracle.next_state_values  policy_learners/sequential_decision_making/deep_q_learning.py:130-167
  DqnOracle.bellman_target     deep_td_learning.py:313-317
  DqnOracle.gradients          autograd of MSELoss(mean) through the MLP (deep_td_learning.py:319-355),
                               written out by hand
  DqnOracle.adamw              deep_td_learning.py:183-185 -> torch.optim.AdamW(amsgrad=True), i.e.
                               torch/optim/adam.py::_single_tensor_adam op for op
  DqnOracle.soft_update        neural_networks/common/utils.py:214-226
  DqnOracle.learn_batch        deep_td_learning.py:269-290, :333-360
  DqnOracle.learn              policy_learners/policy_learner.py:162-195
  QNetOracle                   q_value_networks.py:124-249, :352-508 (other depths, multi-head,
                               dueling) under deep_td_learning.py:269-360 / double_dqn.py:29-57

Parity is PINNED: tests/test_oracle_golden.py checks every function here against fixtures minted by
running the real reference (oracle/make_golden.py -> tests/golden/dqn_*.pt).

`philox_sample_indices` restates the *device* sampler of pearl_amd (arena.hip) — that one has no
reference counterpart (the reference uses Python's MT19937); it pins the kernel to its own spec.
"""
from __future__ import annotations

import math
import random
from collections import deque
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

F32 = torch.float32


# --------------------------------------------------------------------------------------
# replay
# --------------------------------------------------------------------------------------
def padded_actions_and_mask(max_number_actions: int, n: int, actions_batch: torch.Tensor):
    """(A, d) float32 table whose first n rows are the available actions, and the (A,) bool mask
    that is True for the padding rows."""
    table = torch.zeros((max_number_actions, actions_batch.shape[1]), dtype=F32)
    table[:n] = actions_batch
    mask = torch.zeros(max_number_actions, dtype=torch.bool)
    mask[n:] = True
    return table, mask


class ReplayOracle:
    FIELDS = ("state", "action", "reward", "terminated", "truncated", "next_state",
              "curr_available_actions", "curr_unavailable_actions_mask",
              "next_available_actions", "next_unavailable_actions_mask")

    def __init__(self, capacity: int) -> None:
        self.memory: deque = deque([], maxlen=capacity)

    def __len__(self) -> int:
        return len(self.memory)

    def push(self, state, action, reward, terminated, truncated, n_curr: int, next_state,
             n_next: int, max_number_actions: int) -> None:
        """Discrete index actions 0..n-1 as (n, 1) tables, like DiscreteActionSpace of
        tensor([k]) elements."""
        ca, cm = padded_actions_and_mask(max_number_actions, n_curr,
                                         torch.arange(n_curr, dtype=F32).view(-1, 1))
        na, nm = padded_actions_and_mask(max_number_actions, n_next,
                                         torch.arange(n_next, dtype=F32).view(-1, 1))
        self.memory.append(dict(
            state=state.clone().detach().unsqueeze(0),
            action=action.clone().detach().unsqueeze(0),
            reward=torch.tensor([reward]),
            terminated=torch.tensor([terminated]),
            truncated=torch.tensor([truncated]),
            next_state=next_state.clone().detach().unsqueeze(0),
            curr_available_actions=ca.unsqueeze(0), curr_unavailable_actions_mask=cm.unsqueeze(0),
            next_available_actions=na.unsqueeze(0), next_unavailable_actions_mask=nm.unsqueeze(0)))

    @staticmethod
    def collate(rows: Sequence[dict]) -> Dict[str, torch.Tensor]:
        out = {k: torch.cat([r[k] for r in rows]) for k in ReplayOracle.FIELDS}
        if rows and "next_action" in rows[0]:
            out["next_action"] = torch.cat([r["next_action"] for r in rows])
        out["state"] = out["state"].type(F32)
        out["next_state"] = out["next_state"].type(F32)
        return out

    def sample(self, batch_size: int) -> Dict[str, torch.Tensor]:
        if batch_size > len(self):
            raise ValueError(f"Can't get a batch of size {batch_size} from a replay buffer with "
                             f"only {len(self)} elements")
        return self.collate(random.sample(self.memory, batch_size))

    def sample_at(self, logical_indices: Sequence[int]) -> Dict[str, torch.Tensor]:
        return self.collate([self.memory[int(i)] for i in logical_indices])


class SarsaReplayOracle(ReplayOracle):
    """SARSAReplayBuffer (sarsa_replay_buffer.py:22-101): a push is held back until the next push
    supplies `next_action` (when its state equals the cached next_state); terminal / truncated
    pushes are stored at once with their own action as a dummy next_action; the cache survives a
    terminal push."""

    def __init__(self, capacity: int) -> None:
        super().__init__(capacity)
        self.cache = None

    def push(self, state, action, reward, terminated, truncated, n_curr: int, next_state,
             n_next: int, max_number_actions: int) -> None:
        probe = ReplayOracle(1)
        ReplayOracle.push(probe, state, action, reward, terminated, truncated, n_curr, next_state,
                          n_next, max_number_actions)
        row = probe.memory[0]
        if self.cache is not None and torch.equal(self.cache["next_state"], row["state"]):
            self.memory.append(dict(self.cache, next_action=row["action"]))        # :55-71
        if not (terminated or truncated):
            self.cache = row                                                        # :72-86
        else:
            self.memory.append(dict(row, next_action=row["action"]))               # :87-101


class HerReplayOracle(ReplayOracle):
    """HindsightExperienceReplayBuffer, "final" strategy
    (hindsight_experience_replay_buffer.py:19-160): at the end of an episode its transitions are
    pushed again with the goal slot overwritten by next_state[:-goal_dim] of the last transition and
    the reward (optionally terminated) recomputed."""

    def __init__(self, capacity: int, goal_dim: int, reward_fn, terminated_fn=None) -> None:
        super().__init__(capacity)
        self.goal_dim, self.reward_fn, self.terminated_fn = goal_dim, reward_fn, terminated_fn
        self.trajectory: list = []

    def push(self, state, action, reward, terminated, truncated, n_curr: int, next_state,
             n_next: int, max_number_actions: int) -> None:
        ReplayOracle.push(self, state, action, reward, terminated, truncated, n_curr, next_state,
                          n_next, max_number_actions)
        self.trajectory.append((state, action, next_state, n_curr, n_next, terminated, truncated))
        if terminated or truncated:
            goal = next_state[: -self.goal_dim]
            for (st, act, nst, nc, nn_, term, trunc) in self.trajectory:
                st[-self.goal_dim:] = goal
                nst[-self.goal_dim:] = goal
                ReplayOracle.push(self, st, act, self.reward_fn(st, act),
                                  term if self.terminated_fn is None else self.terminated_fn(st, act),
                                  trunc, nc, nst, nn_, max_number_actions)
            self.trajectory = []


class BootstrapReplayOracle(ReplayOracle):
    """BootstrapReplayBuffer (bootstrap_replay_buffer.py:23-114): every stored transition carries
    one (1, ensemble_size) mask drawn with ``torch.bernoulli(tensor(p).repeat(1, K))`` from torch's
    GLOBAL generator at push time (:64-66); ``sample`` concatenates the masks of the sampled rows
    (:104) and drops ``cost``."""

    def __init__(self, capacity: int, p: float, ensemble_size: int) -> None:
        super().__init__(capacity)
        self.p, self.ensemble_size = p, ensemble_size

    def push(self, state, action, reward, terminated, truncated, n_curr: int, next_state,
             n_next: int, max_number_actions: int) -> None:
        probe = ReplayOracle(1)
        ReplayOracle.push(probe, state, action, reward, terminated, truncated, n_curr, next_state,
                          n_next, max_number_actions)
        mask = torch.bernoulli(torch.tensor(self.p).repeat(1, self.ensemble_size))
        self.memory.append(dict(probe.memory[0], bootstrap_mask=mask))

    @staticmethod
    def collate(rows: Sequence[dict]) -> Dict[str, torch.Tensor]:
        out = ReplayOracle.collate(rows)
        out["bootstrap_mask"] = torch.cat([r["bootstrap_mask"] for r in rows])
        return out

    def sample_at(self, logical_indices: Sequence[int]) -> Dict[str, torch.Tensor]:
        return self.collate([self.memory[int(i)] for i in logical_indices])

    def sample(self, batch_size: int) -> Dict[str, torch.Tensor]:
        if batch_size > len(self):
            raise ValueError(f"Can't get a batch of size {batch_size} from a replay buffer with "
                             f"only {len(self)} elements")
        return self.collate(random.sample(self.memory, batch_size))


def filter_by_bootstrap_mask(batch: Dict[str, torch.Tensor], z: int) -> Dict[str, torch.Tensor]:
    """filter_batch_by_bootstrap_mask (transition.py:252-301): rows whose mask is 1 for member z."""
    keep = batch["bootstrap_mask"][:, z] == 1
    return {k: v[keep] for k, v in batch.items() if k != "bootstrap_mask" and v is not None}


def one_hot(x: torch.Tensor, n: int) -> torch.Tensor:
    if x.ndim == 1:
        x = x.unsqueeze(-1)
    return torch.nn.functional.one_hot(x.long(), num_classes=n).squeeze(dim=-2).float()


def preprocess(batch: Dict[str, torch.Tensor], n_actions: int) -> Dict[str, torch.Tensor]:
    out = dict(batch)
    out["action"] = one_hot(batch["action"], n_actions)
    if batch.get("next_action") is not None:
        out["next_action"] = one_hot(batch["next_action"], n_actions)
    for k in ("curr_available_actions", "next_available_actions"):
        if batch.get(k) is not None:
            out[k] = one_hot(batch[k], n_actions)
    return out


# --------------------------------------------------------------------------------------
# learner
# --------------------------------------------------------------------------------------
PARAM_KEYS = ("_model.0.0.weight", "_model.0.0.bias", "_model.1.0.weight", "_model.1.0.bias",
              "_model.2.0.weight", "_model.2.0.bias")


class DqnOracle:
    """Two-hidden-layer VanillaQValueNetwork + target copy + AdamW(amsgrad), fp32 on the CPU."""

    def __init__(self, params: Dict[str, torch.Tensor], target: Dict[str, torch.Tensor],
                 gamma: float = 0.99, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.01, tau: float = 0.75, target_update_freq: int = 10,
                 double_q: bool = False, sarsa: bool = False, conservative_alpha: float = 0.0):
        # double_q: DoubleDQN.get_next_state_values (double_dqn.py:29-57) instead of
        # DeepQLearning's (deep_q_learning.py:130-167); sarsa: DeepSARSA's (deep_sarsa.py:59-78)
        self.double_q = bool(double_q)
        self.sarsa = bool(sarsa)
        # > 0: DeepTDLearning(is_conservative=True): loss += alpha * compute_cql_loss
        # (deep_td_learning.py:323-327, loss_fn_utils.py:17-72)
        self.conservative_alpha = float(conservative_alpha)
        self.p = {k: params[k].detach().clone().to(F32) for k in PARAM_KEYS}
        self.t = {k: target[k].detach().clone().to(F32) for k in PARAM_KEYS}
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.vmax = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.gamma, self.lr, self.betas, self.eps = gamma, lr, betas, eps
        self.weight_decay, self.tau, self.freq = weight_decay, tau, target_update_freq
        self.adam_step = 0
        self.training_steps = 0

    # -- forward pieces
    @staticmethod
    def _mlp(w: Dict[str, torch.Tensor], x: torch.Tensor):
        z1 = x @ w[PARAM_KEYS[0]].t() + w[PARAM_KEYS[1]]
        h1 = torch.clamp_min(z1, 0)
        z2 = h1 @ w[PARAM_KEYS[2]].t() + w[PARAM_KEYS[3]]
        h2 = torch.clamp_min(z2, 0)
        q = h2 @ w[PARAM_KEYS[4]].t() + w[PARAM_KEYS[5]]
        return h1, h2, q.squeeze(-1)

    def q_values(self, state: torch.Tensor, action_rep: torch.Tensor) -> torch.Tensor:
        return self._mlp(self.p, torch.cat([state, action_rep], dim=-1))[2]

    def next_state_values(self, next_state, next_avail_rep, next_mask) -> torch.Tensor:
        B, A, _ = next_avail_rep.shape
        s = next_state.unsqueeze(1).expand(B, A, next_state.shape[1])
        xs = torch.cat([s, next_avail_rep], dim=-1)
        if self.double_q:
            # a' = argmax_a Q_online(s', a) over the available actions (double_dqn.py:40-48) ...
            q = self._mlp(self.p, xs)[2].clone()                   # (B, A)
            q[next_mask] = -float("inf")
            choice = q.max(1)[1]
            # ... valued by the target network (double_dqn.py:49-56)
            chosen = next_avail_rep[torch.arange(B), choice]         # (B, AD)
            return self._mlp(self.t, torch.cat([next_state, chosen], dim=-1))[2]
        q = self._mlp(self.t, xs)[2]  # (B, A)
        q = q.clone()
        q[next_mask] = -float("inf")
        return q.max(1)[0]

    def bellman_target(self, batch) -> torch.Tensor:
        if self.sarsa:      # Q_target(s', committed next action)
            nv = self._mlp(self.t, torch.cat([batch["next_state"], batch["next_action"]], dim=-1))[2]
        else:
            nv = self.next_state_values(batch["next_state"], batch["next_available_actions"],
                                        batch["next_unavailable_actions_mask"])
        return nv * self.gamma * (1 - batch["terminated"].float()) + batch["reward"]

    # -- backward, by hand
    def gradients(self, batch, target: torch.Tensor, scale: float = 1.0):
        x = torch.cat([batch["state"], batch["action"]], dim=-1)
        h1, h2, q = self._mlp(self.p, x)
        B = q.shape[0]
        dq = (2.0 / B) * scale * (q - target)                      # d mean((q - y)^2) / dq
        g = {}
        g[PARAM_KEYS[4]] = (dq.unsqueeze(0) @ h2)                  # (1, H2)
        g[PARAM_KEYS[5]] = dq.sum().reshape(1)
        dz2 = (dq.unsqueeze(1) * self.p[PARAM_KEYS[4]]) * (h2 > 0)
        g[PARAM_KEYS[2]] = dz2.t() @ h1
        g[PARAM_KEYS[3]] = dz2.sum(0)
        dz1 = (dz2 @ self.p[PARAM_KEYS[2]]) * (h1 > 0)
        g[PARAM_KEYS[0]] = dz1.t() @ x
        g[PARAM_KEYS[1]] = dz1.sum(0)
        return q, g

    def adamw(self, grads: Dict[str, torch.Tensor]) -> None:
        self.adam_step += 1
        t = self.adam_step
        b1, b2 = self.betas
        bc1 = 1 - b1 ** t
        bc2 = 1 - b2 ** t
        step_size = self.lr / bc1
        bc2_sqrt = bc2 ** 0.5
        for k in PARAM_KEYS:
            p, g = self.p[k], grads[k]
            p.mul_(1 - self.lr * self.weight_decay)
            self.m[k].lerp_(g, 1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            torch.maximum(self.vmax[k], self.v[k], out=self.vmax[k])
            denom = (self.vmax[k].sqrt() / bc2_sqrt).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-step_size)

    def soft_update(self) -> None:
        for k in PARAM_KEYS:
            self.t[k].copy_(self.tau * self.p[k] + (1.0 - self.tau) * self.t[k])

    def cql_loss(self, w: Dict[str, torch.Tensor], batch) -> torch.Tensor:
        """compute_cql_loss (loss_fn_utils.py:17-72), literally: logsumexp over the Q-values of all
        (padded, unmasked) current available actions, minus the mean of
        ``q_all.gather(1, batch.action.long())`` — with a one-hot ``batch.action`` that index matrix
        holds 0s and 1s, so the second term averages Q(s, a_0) (A - 1 times) and Q(s, a_1) (once)
        per row whatever action was taken.  Restated as the reference computes it."""
        ca = batch["curr_available_actions"]
        B, A, _ = ca.shape
        s = batch["state"].unsqueeze(1).expand(B, A, batch["state"].shape[1])
        q_all = self._mlp(w, torch.cat([s, ca], dim=-1))[2].view(B, -1)
        picked = q_all.gather(1, batch["action"].long())
        return torch.logsumexp(q_all, dim=-1).mean() - picked.mean()

    def conservative_gradients(self, batch, target: torch.Tensor):
        """Gradients of  mse(Q(s, a), y) + alpha * cql_loss  by autograd on copies of the parameters."""
        w = {k: v.detach().clone().requires_grad_(True) for k, v in self.p.items()}
        q = self._mlp(w, torch.cat([batch["state"], batch["action"]], dim=-1))[2]
        loss = torch.nn.functional.mse_loss(q, target) + self.conservative_alpha * self.cql_loss(w, batch)
        grads = torch.autograd.grad(loss, [w[k] for k in PARAM_KEYS])
        return q.detach(), dict(zip(PARAM_KEYS, grads)), float(loss.detach())

    def learn_batch(self, batch) -> float:
        """`batch` is preprocessed (one-hot actions).  Returns mean |Q - target|."""
        if (self.training_steps + 1) % self.freq == 0:
            self.soft_update()
        target = self.bellman_target(batch)
        if self.conservative_alpha > 0:
            q, g, _ = self.conservative_gradients(batch, target)
            self.adamw(g)
            return float((q - target).abs().mean())
        q, g = self.gradients(batch, target)
        self.adamw(g)
        return float((q - target).abs().mean())

    def learn(self, replay: ReplayOracle, rounds: int, batch_size: int, n_actions: int,
              index_lists: Optional[Sequence[Sequence[int]]] = None) -> List[float]:
        losses = []
        for r in range(rounds):
            self.training_steps += 1
            raw = replay.sample(batch_size) if index_lists is None else replay.sample_at(index_lists[r])
            losses.append(self.learn_batch(preprocess(raw, n_actions)))
        return losses


# --------------------------------------------------------------------------------------
# Q-network architectures beyond the two-hidden-layer VanillaQValueNetwork
# --------------------------------------------------------------------------------------
# ActivationType (common/utils.py:29-56) as formulas: nn.LeakyReLU() slope 0.01, nn.Softplus() beta 1 /
# threshold 20
_HIDDEN_ACTS = {
    "relu": lambda z: torch.clamp_min(z, 0),
    "leaky_relu": lambda z: torch.where(z > 0, z, 0.01 * z),
    "tanh": torch.tanh,
    "softplus": lambda z: torch.where(z > 20, z, torch.log1p(torch.exp(z))),
    "sigmoid": lambda z: 1.0 / (1.0 + torch.exp(-z)),
    "linear": lambda z: z,
}


def _is_buffer_key(k: str) -> bool:
    return k.endswith(("running_mean", "running_var", "num_batches_tracked"))


def _mlp_sd(w: Dict[str, torch.Tensor], prefix: str, x: torch.Tensor,
            hidden_activation: str = "relu") -> torch.Tensor:
    """mlp_block (common/utils.py:75-152) from state-dict tensors: per hidden block Linear
    [+ nn.LayerNorm: between the Linear and its activation, utils.py:110-113; biased variance, eps
    1e-5] + the hidden activation [+ nn.BatchNorm1d AFTER it, :119-121, in TRAINING mode: the
    statistics of the batch at hand — the reference never switches to eval()], the whole block inside a
    ResidualWrapper (`{i}.module.` keys, :122-131) when its widths agree; then the last Linear
    (possibly wrapped, :142-150).  (Dropout, which has no tensors, is not restated here.)"""
    act = _HIDDEN_ACTS[hidden_activation]

    def base_of(i):
        b = f"{prefix}{i}."
        return (b + "module.", True) if (b + "module.0.weight") in w else (b, False)

    i = 0
    while f"{prefix}{i + 1}.0.weight" in w or f"{prefix}{i + 1}.module.0.weight" in w:
        b, wrapped = base_of(i)
        inp = x
        x = torch.nn.functional.linear(x, w[b + "0.weight"], w[b + "0.bias"])
        if (b + "1.weight") in w:        # index 1 with tensors: the LayerNorm (an activation has none)
            mu = x.mean(dim=-1, keepdim=True)
            var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
            x = (x - mu) / torch.sqrt(var + 1e-5) * w[b + "1.weight"] + w[b + "1.bias"]
        x = act(x)
        # index >= 2 with tensors: the BatchNorm1d behind the activation ([Linear, act, BN],
        # [Linear, LN, act, BN], [Linear, (LN,) Dropout, act, BN])
        bn = next((f"{b}{j}." for j in (2, 3, 4) if f"{b}{j}.weight" in w), None)
        if bn is not None:
            assert x.ndim == 2, "BatchNorm1d inside an mlp_block takes (N, C) inputs"
            x = torch.nn.functional.batch_norm(x, None, None, w[bn + "weight"], w[bn + "bias"], training=True,
                                               momentum=0.1, eps=1e-5)
        if wrapped:
            x = inp + x
        i += 1
    b, wrapped = base_of(i)
    out = torch.nn.functional.linear(x, w[b + "0.weight"], w[b + "0.bias"])
    return x + out if wrapped else out


class QNetOracle:
    """DeepQLearning / DoubleDQN (deep_td_learning.py:269-360, deep_q_learning.py:130-167,
    double_dqn.py:29-57) over
      kind "vanilla"   VanillaQValueNetwork of any depth           q_value_networks.py:124-182
      kind "multihead" VanillaQValueMultiHeadNetwork               q_value_networks.py:185-249
      kind "dueling"   DuelingQValueNetwork                        q_value_networks.py:352-508
    Forward passes restated from the state-dict tensors; gradients by autograd (these networks have
    no hand-written backward here); AdamW(amsgrad) and the soft update as in DqnOracle."""

    def __init__(self, params: Dict[str, torch.Tensor], target: Dict[str, torch.Tensor], kind: str,
                 double_q: bool = False, gamma: float = 0.99, lr: float = 1e-3, betas=(0.9, 0.999),
                 eps: float = 1e-8, weight_decay: float = 0.01, tau: float = 0.75,
                 target_update_freq: int = 10, hidden_activation: str = "relu",
                 cql_alpha: Optional[float] = None) -> None:
        assert kind in ("vanilla", "multihead", "dueling")
        self.kind, self.double_q = kind, bool(double_q)
        # is_conservative (deep_td_learning.py:323-327): loss += alpha * compute_cql_loss
        self.cql_alpha = cql_alpha
        self.act = hidden_activation      # (LayerNorm is read off the state dict's keys)
        # (parameters only: a BatchNorm1d's running statistics are buffers no optimizer or soft update
        #  touches, and training-mode batch norm does not read them)
        self.keys = [k for k in params.keys() if not _is_buffer_key(k)]
        self.p = {k: params[k].detach().clone().to(F32) for k in self.keys}
        self.t = {k: target[k].detach().clone().to(F32) for k in self.keys}
        self.m = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.vmax = {k: torch.zeros_like(v) for k, v in self.p.items()}
        self.gamma, self.lr, self.betas, self.eps = gamma, lr, betas, eps
        self.weight_decay, self.tau, self.freq = weight_decay, tau, target_update_freq
        self.adam_step = 0
        self.training_steps = 0

    # ---- get_q_values(state (B, S), actions (B, Q, AD) | (B, AD), curr_available (B, M, AD) | None)
    def q(self, w, state, action, curr_avail=None) -> torch.Tensor:
        acts = action if action.ndim == 3 else action.unsqueeze(1)
        B, Q = acts.shape[0], acts.shape[1]
        if self.kind == "vanilla":
            s = state.unsqueeze(1).expand(B, Q, state.shape[1])
            out = _mlp_sd(w, "_model.", torch.cat([s, acts], dim=-1), self.act).squeeze(-1)
        elif self.kind == "multihead":
            f = _mlp_sd(w, "_model.", state, self.act).unsqueeze(-1)         # (B, A, 1)
            out = torch.bmm(acts, f).squeeze(-1)
        else:
            feats = _mlp_sd(w, "state_arch._model.", state)
            value = _mlp_sd(w, "value_arch._model.", feats)                   # (B, 1)

            def adv(a):
                f = feats.unsqueeze(1).expand(B, a.shape[1], feats.shape[1])
                return _mlp_sd(w, "advantage_arch._model.", torch.cat([f, a], dim=-1)).squeeze(-1)

            advantage = adv(acts)
            mean = (advantage if curr_avail is None else adv(curr_avail)).mean(dim=-1, keepdim=True)
            out = value + advantage - mean
        return out if action.ndim == 3 else out.squeeze(-1)

    def next_state_values(self, batch) -> torch.Tensor:
        ns, nav, mask = (batch["next_state"], batch["next_available_actions"],
                         batch["next_unavailable_actions_mask"])
        B = ns.shape[0]
        if self.double_q:
            qs = self.q(self.p, ns, nav).clone()
            qs[mask] = -float("inf")
            choice = qs.max(1)[1]
            return self.q(self.t, ns, nav[torch.arange(B), choice])
        qv = self.q(self.t, ns, nav).clone()
        qv[mask] = -float("inf")
        return qv.max(1)[0]

    def bellman_target(self, batch) -> torch.Tensor:
        return self.next_state_values(batch) * self.gamma * (1 - batch["terminated"].float()) + batch["reward"]

    def gradients(self, batch, target):
        w = {k: v.detach().clone().requires_grad_(True) for k, v in self.p.items()}
        q = self.q(w, batch["state"], batch["action"], batch.get("curr_available_actions"))
        loss = torch.nn.functional.mse_loss(q, target)
        if self.cql_alpha is not None:
            # compute_cql_loss (loss_fn_utils.py:17-72): the all-actions table, the reference's
            # gather with long(one-hot action) as the index, logsumexp mean minus the gathered mean
            q_all = self.q(w, batch["state"], batch["curr_available_actions"]).view(q.shape[0], -1)
            in_batch = q_all.gather(1, batch["action"].long())
            loss = loss + self.cql_alpha * (torch.logsumexp(q_all, dim=-1).mean() - in_batch.mean())
        grads = torch.autograd.grad(loss, [w[k] for k in self.keys])
        return q.detach(), dict(zip(self.keys, grads))

    def adamw(self, grads) -> None:
        self.adam_step += 1
        b1, b2 = self.betas
        bc1, bc2 = 1 - b1 ** self.adam_step, 1 - b2 ** self.adam_step
        for k in self.keys:
            p, g = self.p[k], grads[k]
            p.mul_(1 - self.lr * self.weight_decay)
            self.m[k].lerp_(g, 1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            torch.maximum(self.vmax[k], self.v[k], out=self.vmax[k])
            denom = (self.vmax[k].sqrt() / (bc2 ** 0.5)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-(self.lr / bc1))

    def learn_batch(self, batch) -> float:
        if (self.training_steps + 1) % self.freq == 0:
            for k in self.keys:
                self.t[k].copy_(self.tau * self.p[k] + (1.0 - self.tau) * self.t[k])
        with torch.no_grad():
            target = self.bellman_target(batch)
        q, g = self.gradients(batch, target)
        self.adamw(g)
        return float((q - target).abs().mean())

    def learn(self, replay: ReplayOracle, rounds: int, batch_size: int, n_actions: int) -> List[float]:
        losses = []
        for _ in range(rounds):
            self.training_steps += 1
            losses.append(self.learn_batch(preprocess(replay.sample(batch_size), n_actions)))
        return losses


# --------------------------------------------------------------------------------------
# device sampler spec (no reference counterpart)
# --------------------------------------------------------------------------------------
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
_U32 = 0xFFFFFFFF


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        c0, c1, c2, c3 = ((p1 >> 32) ^ c1 ^ k0) & _U32, p1 & _U32, ((p0 >> 32) ^ c3 ^ k1) & _U32, p0 & _U32
        k0 = (k0 + _W0) & _U32
        k1 = (k1 + _W1) & _U32
    return c0, c1, c2, c3


def _bounded(words, n):
    thresh = ((1 << 32) - n) % n
    m = 0
    for w in words:
        m = w * n
        if (m & _U32) >= thresh:
            break
    return m >> 32


def philox_sample_indices(population: int, seed: int, offset: int, B: int) -> np.ndarray:
    """The rule of arena.hip::sample_indices_kernel: in round t every unresolved position i
    proposes draw(philox(i, t, offset; seed)); a proposal wins iff its value was not accepted in
    an earlier round and i is the lowest position proposing it in this round."""
    k0, k1 = seed & _U32, (seed >> 32) & _U32
    o0, o1 = offset & _U32, (offset >> 32) & _U32
    out = np.full(B, -1, dtype=np.int64)
    taken = set()
    pending = list(range(B))
    t = 0
    while pending:
        proposals = {}
        for i in pending:
            v = _bounded(philox4x32_10(i, t, o0, o1, k0, k1), population)
            proposals.setdefault(v, []).append(i)
        nxt = []
        for v, who in proposals.items():
            if v in taken:
                nxt.extend(who)
                continue
            w = min(who)
            out[w] = v
            taken.add(v)
            nxt.extend(i for i in who if i != w)
        pending = sorted(nxt)
        t += 1
    return out
