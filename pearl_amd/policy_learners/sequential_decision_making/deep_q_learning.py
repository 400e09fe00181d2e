"""``DeepQLearning`` whose ``learn`` / ``learn_batch`` run as HIP kernels on MI355X.

Drop-in for pearl/policy_learners/sequential_decision_making/deep_q_learning.py:35-189 on top
of deep_td_learning.py:46-477 — same constructor arguments and defaults (AdamW lr 1e-3,
amsgrad, weight decay 0.01; gamma 0.99; target_update_freq 10; soft_update_tau 0.75), same
attributes other components read (``_Q``, ``_Q_target``, ``_optimizer``, ``_training_steps``,
``on_policy``, ``_is_action_continuous`` ...), same report ``{"loss": mean |Q - target|}``.

What differs is where the arithmetic happens.  The nn.Parameters of ``_Q`` / ``_Q_target`` stay
the source of truth (``state_dict`` / ``compare`` keep working) but are re-pointed into flat fp32
buffers; libpearl_amd.so reads and updates them in place:

* ``learn_batch(batch)``  -> ``pa_dqn_step``: soft target update (deep_td_learning.py:283-284),
  online forward, fused target-network forward + mask + max + Bellman target
  (deep_q_learning.py:130-167), MSE, backward, AdamW(amsgrad);
* ``learn(replay_buffer)`` on an arena-backed buffer -> ``pa_dqn_learn``: the whole
  ``training_rounds`` loop on the device (index lists of all rounds in one launch, gather and
  target-network pass batched per target-update window), one host synchronisation per call (for
  the report) instead of one ``.item()`` per step.

Data parallelism (not in the reference): if ``torch.distributed`` is initialised every rank keeps
its own arena shard and local batch; ``pa_dqn_learn`` pre-scales gradients by 1/world and calls
back into ``torch.distributed.all_reduce`` (RCCL) once per round through the all-reduce hooks of
``pa_learn_args``, with the next round's target-network pass enqueued while the exchange is in
flight; AdamW (``adamw_dqn_kernel``) runs after it.
"""
from __future__ import annotations

import copy
import ctypes as C
import os
import random
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist
from torch import optim

from ... import _native as N
from ...action_representation_modules import (ActionRepresentationModule,
                                              OneHotActionTensorRepresentationModule)
from ...neural_networks.sequential_decision_making.q_value_networks import (
    DuelingQValueNetwork, QValueNetwork, VanillaQValueMultiHeadNetwork, VanillaQValueNetwork)
from ...replay_buffers.basic_replay_buffer import TensorBasedReplayBuffer
from ...replay_buffers.replay_buffer import ReplayBuffer
from ...replay_buffers.transition import TransitionBatch
from ..exploration import EGreedyExploration, ExplorationModule
from ..policy_learner import PolicyLearner, accept_optimizer, perf_reported
from .generic_q import GenericTd, make_ops, mlp_spec, plain_relu_mlp

_FLAT_NAMES = ("q", "q_target", "grad", "exp_avg", "exp_avg_sq", "max_exp_avg_sq")


class _NativeDqn:
    """Owns the pa_dqn handle and the flat buffers; never deep-copied or pickled."""

    def __init__(self) -> None:
        self.handle: Optional[C.c_void_p] = None
        self.flat: Dict[str, torch.Tensor] = {}
        self.sig: Tuple = ()
        self.desc_key: Tuple = ()
        self.loss_buf: Optional[torch.Tensor] = None
        self.comm: Optional[C.c_void_p] = None

    def close(self) -> None:
        h, self.handle = self.handle, None
        if h:
            N.lib().pa_dqn_destroy(h)
        self.comm = None      # the communicator belongs to the process (pearl_amd/_comm.py)
        q, self.cql = getattr(self, "cql", None), None
        if q:
            N.lib().pa_mlp_destroy(q[1])

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __deepcopy__(self, memo: dict) -> "_NativeDqn":
        return _NativeDqn()

    def __getstate__(self) -> dict:
        return {}

    def __setstate__(self, state: dict) -> None:
        self.__init__()


def world_size() -> int:
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def allreduce_sum_(flat: torch.Tensor) -> torch.Tensor:
    """Sum a flat gradient buffer over the data-parallel group (RCCL on GPUs, gloo on CPU)."""
    if world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


class DeepQLearning(PolicyLearner):
    # How the next state is valued (pa_dqn_desc.double_q): 0 = max over the available next actions
    # of Q_target (deep_q_learning.py:130-167); 1 = DoubleDQN's rule (double_dqn.py:29-57);
    # 2 = DeepSARSA's, Q_target(s', batch.next_action) (deep_sarsa.py:59-78)
    _double_q: int = 0

    def __init__(self, action_space: Any = None, hidden_dims: Optional[List[int]] = None,
                 exploration_module: Optional[ExplorationModule] = None,
                 learning_rate: float = 0.001, discount_factor: float = 0.99,
                 training_rounds: int = 10, batch_size: int = 128, target_update_freq: int = 10,
                 soft_update_tau: float = 0.75, is_conservative: bool = False,
                 conservative_alpha: Optional[float] = 2.0, state_dim: Optional[int] = None,
                 network_type: type = VanillaQValueNetwork,
                 action_representation_module: Optional[ActionRepresentationModule] = None,
                 network_instance: Optional[QValueNetwork] = None,
                 optimizer: Optional[optim.Optimizer] = None, max_batch_size: Optional[int] = None,
                 **kwargs: Any) -> None:
        super().__init__(
            training_rounds=training_rounds, batch_size=batch_size,
            exploration_module=(exploration_module if exploration_module is not None
                                else EGreedyExploration(0.05)),
            on_policy=False, is_action_continuous=False,
            action_representation_module=action_representation_module, action_space=action_space)
        self._action_space = action_space
        self._learning_rate = learning_rate
        self._discount_factor = discount_factor
        self._target_update_freq = target_update_freq
        self._soft_update_tau = soft_update_tau
        self._is_conservative = is_conservative
        self._conservative_alpha = conservative_alpha
        rep = self.action_representation_module
        if network_instance is not None:
            if not isinstance(network_instance, (VanillaQValueNetwork, VanillaQValueMultiHeadNetwork,
                                                 DuelingQValueNetwork)):
                raise NotImplementedError(
                    f"pearl_amd DeepQLearning: no HIP path for {type(network_instance).__name__} "
                    "(VanillaQValueNetwork, VanillaQValueMultiHeadNetwork, DuelingQValueNetwork are built)")
            self._Q: QValueNetwork = network_instance
        else:
            # deep_td_learning.py:133-173 make_specified_network
            assert state_dim is not None and hidden_dims is not None
            if network_type is VanillaQValueMultiHeadNetwork:
                self._Q = VanillaQValueMultiHeadNetwork(
                    state_dim=state_dim, action_dim=rep.representation_dim,
                    hidden_dims=list(hidden_dims), output_dim=rep.max_number_actions)
            elif network_type in (VanillaQValueNetwork, DuelingQValueNetwork):
                self._Q = network_type(state_dim=state_dim, action_dim=rep.representation_dim,
                                       hidden_dims=list(hidden_dims), output_dim=1)
            else:
                raise NotImplementedError(
                    f"pearl_amd DeepQLearning: network_type {network_type.__name__} is not built "
                    "(VanillaQValueNetwork, VanillaQValueMultiHeadNetwork, DuelingQValueNetwork are)")
        # Which engine trains it.  Fused MI355X path (pa_dqn_*): VanillaQValueNetwork, two ReLU hidden
        # layers of at most 256 units.  Everything else the reference's mlp_block can express with
        # Linear [+ LayerNorm] + relu / leaky_relu / tanh / softplus / sigmoid / linear hidden layers
        # goes through the generic pa_mlp engine (generic_q.py); batch norm / dropout / residual blocks
        # are refused, loudly — never trained as if they were something else.
        self._fused = False
        if isinstance(self._Q, VanillaQValueNetwork):
            spec = mlp_spec(self._Q._model)
            if spec is None:
                raise NotImplementedError(
                    "pearl_amd DeepQLearning: the Q network is not an mlp_block the HIP engine computes "
                    "(batch norm, dropout or residual blocks, or an activation without a kernel)")
            if spec["bnorms"]:
                raise NotImplementedError(
                    "pearl_amd DeepQLearning: BatchNorm1d inside a VanillaQValueNetwork — the reference's "
                    "own forward raises there (get_q_values feeds a (B, A, S + AD) tensor, which is not "
                    "BatchNorm1d's (N, C) / (N, C, L): q_value_networks.py:152-174); multi-head networks "
                    "take batch norm")
            lin = spec["linears"]
            self._fused = (spec["plain"] and len(lin) == 3 and lin[0].out_features <= 256
                           and lin[1].out_features <= 256 and lin[2].out_features == 1)
        self._Q_target: VanillaQValueNetwork = copy.deepcopy(self._Q)
        if optimizer is not None:
            # (deep_td_learning.py:183-185: used as handed over; see accept_optimizer for what the HIP
            #  step can honour)
            self._optimizer: optim.Optimizer = accept_optimizer(optimizer, self._Q.parameters(),
                                                                type(self).__name__)
        else:
            self._optimizer = optim.AdamW(self._Q.parameters(), lr=learning_rate, amsgrad=True)
        self._max_batch_size = max_batch_size
        self._native = _NativeDqn()
        self._generic_td: Optional[GenericTd] = None
        self.data_parallel = True

    # ------------------------------------------------------------------ plumbing
    @property
    def optimizer(self) -> optim.Optimizer:
        return self._optimizer

    def set_history_summarization_module(self, value: torch.nn.Module) -> None:
        if any(True for _ in value.parameters()):
            raise NotImplementedError(
                "pearl_amd DeepQLearning: trainable history summarisation modules are not built")
        self._history_summarization_module = value

    def reset(self, action_space: Any) -> None:
        self._action_space = action_space

    def _linears(self) -> Tuple[List[torch.nn.Linear], List[torch.nn.Linear]]:
        """Linear layers of the online and the target network.  Walking the module tree costs
        ~10 us per call and learn() asks several times per call: cached per module pair."""
        key = (id(self._Q), id(self._Q_target))
        hit = self.__dict__.get("_linears_cache")
        if hit is None or hit[0] != key:
            hit = (key, self._Q.linear_layers(), self._Q_target.linear_layers())
            self.__dict__["_linears_cache"] = hit
        return hit[1], hit[2]

    def _dims(self) -> Tuple[int, int, int, int]:
        l1, l2, _ = self._linears()[0]
        return self._Q.state_dim, self._Q.action_dim, l1.out_features, l2.out_features

    def _param_pairs(self) -> List[Tuple[torch.nn.Parameter, torch.nn.Parameter]]:
        # (straight out of the modules' parameter dicts: nn.Module.__getattr__ costs 0.6 us per
        #  `layer.weight`, and learn() walks the twelve of them several times per call — the host
        #  side of a short learn() call is measured in those microseconds, tools/shortcall.py)
        memo = self.__dict__.get("_pairs_memo")     # set for the duration of one learn() call
        if memo is not None:
            return memo
        out = []
        for lq, lt in zip(*self._linears()):
            pq, pt = lq._parameters, lt._parameters
            out.append((pq["weight"], pt["weight"]))
            out.append((pq["bias"], pt["bias"]))
        return out

    def _adam_steps(self) -> int:
        state = self._optimizer.state
        for pq, _ in self._param_pairs():
            st = state.get(pq)
            if st and "step" in st:
                return int(st["step"].item())
        return 0

    def _set_adam_steps(self, n: int) -> None:
        # one 0-d step tensor PER parameter, as torch.optim keeps them (a shared tensor would be
        # advanced six times by a torch-side optimizer.step()).  The six are views of ONE 6-element
        # host tensor when this learner created them (_ensure_bound): one fill per learn() call
        # instead of six; step tensors that came from elsewhere (optimizer.load_state_dict) are
        # filled one by one.
        bank = self.__dict__.get("_step_bank")
        state = self._optimizer.state
        banked = bank is not None
        steps = []
        for pq, _ in self._param_pairs():
            st = state.get(pq)
            if st is not None and "step" in st:
                steps.append(st["step"])
        if banked:
            base = bank.data_ptr()
            banked = len(steps) == bank.numel() and all(
                t.data_ptr() == base + 4 * i for i, t in enumerate(steps))
        if banked:
            bank.fill_(float(n))
        else:
            for t in steps:
                t.fill_(float(n))

    def _signature(self) -> Tuple:
        sig = []
        for pq, pt in self._param_pairs():
            st = self._optimizer.state.get(pq, {})
            sig.append((pq.data_ptr(), pt.data_ptr(),
                        tuple(st[k].data_ptr() if k in st else 0
                              for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"))))
        return tuple(sig)

    def _ensure_bound(self, batch_hint: int = 0, actions_hint: int = 0) -> _NativeDqn:
        """(Re)build the flat parameter/optimizer buffers and the pa_dqn handle when needed."""
        nat = self._native
        S, AD, H1, H2 = self._dims()
        p0 = self._linears()[0][0].weight
        if not p0.is_cuda:
            N.require_gpu()
            raise N.NativeError(
                "pearl_amd DeepQLearning: parameters are on the CPU; move the learner to a HIP "
                "device first (PearlAgent does this) — there is no CPU learner path")
        dev = p0.device
        max_b = max(int(self._max_batch_size or 0), int(self._batch_size), int(batch_hint), 1)
        max_a = max(int(getattr(self._action_space, "n", 0) or 0),
                    int(self.action_representation_module.max_number_actions or 0),
                    int(actions_hint), 1)
        opt = self._optimizer.param_groups[0]
        desc_key = (dev.index, S, AD, H1, H2, max_b, max_a, self._discount_factor,
                    self._soft_update_tau, opt["lr"], tuple(opt["betas"]), opt["eps"],
                    opt["weight_decay"], bool(opt["amsgrad"]), int(self._double_q))
        if nat.handle is not None and nat.desc_key != desc_key:
            torch.cuda.synchronize(dev)
            nat.close()
        if nat.handle is None:
            desc = N.DqnDesc(device=dev.index, state_dim=S, action_dim=AD, hidden1=H1, hidden2=H2,
                             max_batch=max_b, max_actions=max_a, discount=self._discount_factor,
                             tau=self._soft_update_tau, lr=opt["lr"], beta1=opt["betas"][0],
                             beta2=opt["betas"][1], eps=opt["eps"],
                             weight_decay=opt["weight_decay"], amsgrad=int(opt["amsgrad"]),
                             double_q=int(self._double_q))
            handle = C.c_void_p()
            N.check(N.lib().pa_dqn_create(C.byref(handle), C.byref(desc)))
            nat.handle, nat.desc_key, nat.sig = handle, desc_key, ()
            # room for 1024 rounds up front: growing it later is an allocation inside learn()
            nat.loss_buf = torch.zeros(max(self._training_rounds, 1024), dtype=torch.float32,
                                       device=dev)
        if nat.sig == self._signature() and nat.sig:
            return nat
        # ---- flatten: re-point every Parameter / grad / optimizer state into flat buffers
        P = int(N.lib().pa_dqn_param_count(S, AD, H1, H2))
        offs = (C.c_int64 * 6)()
        N.check(N.lib().pa_dqn_param_offsets(S, AD, H1, H2, offs))
        flat = {k: torch.zeros(P, dtype=torch.float32, device=dev) for k in _FLAT_NAMES}
        steps = self._adam_steps()
        bank = torch.full((6,), float(steps), dtype=torch.float32)     # the six step counters
        self.__dict__["_step_bank"] = bank
        with torch.no_grad():
            for i, ((pq, pt), off) in enumerate(zip(self._param_pairs(), list(offs))):
                n = pq.numel()
                sl = slice(int(off), int(off) + n)
                flat["q"][sl].copy_(pq.data.reshape(-1).to(dev, torch.float32))
                flat["q_target"][sl].copy_(pt.data.reshape(-1).to(dev, torch.float32))
                st = self._optimizer.state.get(pq) or {}
                for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                    if k in st:
                        flat[k][sl].copy_(st[k].reshape(-1).to(dev, torch.float32))
                pq.data = flat["q"][sl].view(pq.shape)
                pt.data = flat["q_target"][sl].view(pt.shape)
                pq.grad = flat["grad"][sl].view(pq.shape)
                self._optimizer.state[pq] = {
                    "step": bank[i],
                    "exp_avg": flat["exp_avg"][sl].view(pq.shape),
                    "exp_avg_sq": flat["exp_avg_sq"][sl].view(pq.shape),
                }
                if opt["amsgrad"]:      # (torch keeps it only then: optim/adam.py _init_group)
                    self._optimizer.state[pq]["max_exp_avg_sq"] = flat["max_exp_avg_sq"][sl].view(pq.shape)
        bufs = N.DqnBuffers(**{k: flat[k].data_ptr() for k in _FLAT_NAMES})
        N.check(N.lib().pa_dqn_bind(nat.handle, C.byref(bufs)))
        nat.flat = flat
        nat.sig = self._signature()
        return nat

    # ------------------------------------------------------------------ batches
    def _default_next_actions(self, device: torch.device) -> torch.Tensor:
        assert self._action_space is not None and hasattr(self._action_space, "actions_batch"), \
            "next_available_actions missing and no discrete action space configured"
        rep = self.action_representation_module(self._action_space.actions_batch.to(device))
        return rep.to(torch.float32).contiguous()  # (A, AD)   deep_td_learning.py:362-372

    def _native_batch(self, batch: TransitionBatch) -> Tuple[N.DqnBatch, list]:
        S, AD, _, _ = self._dims()
        dev = next(self._Q.parameters()).device

        def f32(t: torch.Tensor) -> torch.Tensor:
            return t.to(device=dev, dtype=torch.float32).contiguous()

        state = f32(batch.state)
        B = state.shape[0]
        assert state.ndim == 2 and state.shape[1] == S, f"state must be ({B}, {S})"
        action = f32(batch.action).reshape(B, -1)
        assert action.shape[1] == AD, (
            f"action representation has width {action.shape[1]}, expected {AD} "
            "(did preprocess_batch run?)")
        assert batch.next_state is not None, "Q-learning needs next_state"
        next_state = f32(batch.next_state)
        reward = f32(batch.reward).reshape(B)
        term = batch.terminated.to(dev).reshape(B).to(torch.uint8).contiguous()
        nav, mask = batch.next_available_actions, batch.next_unavailable_actions_mask
        bcast = 0
        if nav is None:
            nav = self._default_next_actions(dev)
            if mask is None:
                bcast = 1
            else:
                nav = nav.unsqueeze(0).expand(B, -1, -1).contiguous()
        else:
            nav = f32(nav)
            assert nav.ndim == 3 and nav.shape[0] == B and nav.shape[2] == AD, (
                f"next_available_actions must be ({B}, A, {AD}) after preprocess_batch, got "
                f"{tuple(nav.shape)}")
        A = nav.shape[-2]
        if mask is not None:
            mask = mask.to(dev).reshape(B, A).to(torch.uint8).contiguous()
        next_action = None
        if int(self._double_q) == 2:
            assert batch.next_action is not None, "SARSA needs to have next action"
            next_action = f32(batch.next_action).reshape(B, -1)
            assert next_action.shape[1] == AD, (
                f"next_action representation has width {next_action.shape[1]}, expected {AD}")
        nb = N.DqnBatch(B=B, A=A, x=None, state=state.data_ptr(), action_rep=action.data_ptr(),
                        reward=reward.data_ptr(), terminated=term.data_ptr(),
                        next_state=next_state.data_ptr(), next_avail_rep=nav.data_ptr(),
                        next_mask=N.ptr(mask), next_avail_bcast=bcast,
                        next_action_rep=N.ptr(next_action))
        return nb, [state, action, next_state, reward, term, nav, mask, next_action]

    # ------------------------------------------------------------------ generic engine (generic_q.py)
    def _generic(self) -> GenericTd:
        g = self._generic_td
        if g is None or g.ops_key != (id(self._Q), id(self._Q_target), id(self._optimizer)):
            mb = max(int(self._max_batch_size or 0), int(self._batch_size), 1)
            g = GenericTd(make_ops(self._Q, self._Q_target, self._optimizer, mb), int(self._double_q),
                          self._discount_factor, self._soft_update_tau,
                          cql_alpha=self._conservative_alpha if self._is_conservative else None)
            g.ops_key = (id(self._Q), id(self._Q_target), id(self._optimizer))
            self._generic_td = g
        return g

    def __deepcopy__(self, memo: dict) -> "DeepQLearning":
        # the native handles are per-object: a copy rebuilds its own lazily
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            if k in ("_generic_td", "_linears_cache", "_pairs_memo"):
                new.__dict__[k] = None
            else:
                new.__dict__[k] = copy.deepcopy(v, memo)
        return new

    def _generic_batch(self, batch: TransitionBatch) -> Dict[str, Any]:
        dev = next(self._Q.parameters()).device
        if dev.type != "cuda":
            N.require_gpu()
            raise N.NativeError("pearl_amd DeepQLearning: parameters are on the CPU; move the learner "
                                "to a HIP device first — there is no CPU learner path")

        def f32(t: torch.Tensor) -> torch.Tensor:
            return t.to(device=dev, dtype=torch.float32).contiguous()

        state = f32(batch.state)
        B = state.shape[0]
        b: Dict[str, Any] = dict(state=state, action=f32(batch.action).reshape(B, -1),
                                 reward=f32(batch.reward).reshape(B),
                                 terminated=batch.terminated.to(dev).reshape(B).to(torch.uint8).contiguous())
        assert batch.next_state is not None, "Q-learning needs next_state"
        b["next_state"] = f32(batch.next_state)
        ca = batch.curr_available_actions
        b["curr_avail"] = None if ca is None else f32(ca)
        nav, mask = batch.next_available_actions, batch.next_unavailable_actions_mask
        if nav is None:
            nav = self._default_next_actions(dev)            # (A, AD), shared by every row
        else:
            nav = f32(nav)
        b["next_avail"] = nav
        b["next_mask"] = (None if mask is None
                          else mask.to(dev).reshape(B, -1).to(torch.uint8).contiguous())
        b["next_action"] = None
        if int(self._double_q) == 2:
            assert batch.next_action is not None, "SARSA needs to have next action"
            b["next_action"] = f32(batch.next_action).reshape(B, -1)
        return b

    def _learn_batch_generic(self, batch: TransitionBatch) -> Dict[str, Any]:
        g = self._generic()
        b = self._generic_batch(batch)
        g.ops.ensure(b["state"].shape[0])
        losses = torch.empty(2, dtype=torch.float32, device=b["state"].device)
        g.step(b, self._target_update_due(), losses)
        return {"loss": losses[0].item()}     # the reference's per-step .item() (:359)

    # ------------------------------------------------------------------ API
    def _target_update_due(self) -> bool:
        return (self._training_steps + 1) % self._target_update_freq == 0

    def forward(self, batch: TransitionBatch) -> torch.Tensor:
        """Q(s, a) of the online network, after the conditional soft target update
        (deep_td_learning.py:269-290)."""
        if not self._fused:
            g, b = self._generic(), self._generic_batch(batch)
            g.ops.ensure(b["state"].shape[0])
            if self._target_update_due():
                g.ops.soft_update(self._soft_update_tau)
            return g.q_values(b)
        nb, keep = self._native_batch(batch)
        nat = self._ensure_bound(nb.B, nb.A)
        stream = N.stream_ptr(keep[0].device)
        if self._target_update_due():
            N.check(N.lib().pa_dqn_update_target(nat.handle, stream))
        q = torch.empty(nb.B, dtype=torch.float32, device=keep[0].device)
        N.check(N.lib().pa_dqn_qvalues(nat.handle, C.byref(nb), q.data_ptr(), None, None, stream))
        return q

    @torch.no_grad()
    def get_next_state_values(self, batch: TransitionBatch, batch_size: int) -> torch.Tensor:
        """max over available next actions of Q_target(s', a') (deep_q_learning.py:130-167); for
        DoubleDQN, Q_target(s', argmax_a' Q(s', a')) (double_dqn.py:29-57)."""
        if not self._fused:
            g, b = self._generic(), self._generic_batch(batch)
            g.ops.ensure(b["state"].shape[0])
            return g.targets(b, want_next_v=True)[0]
        nb, keep = self._native_batch(batch)
        nat = self._ensure_bound(nb.B, nb.A)
        v = torch.empty(nb.B, dtype=torch.float32, device=keep[0].device)
        N.check(N.lib().pa_dqn_qvalues(nat.handle, C.byref(nb), None, v.data_ptr(), None,
                                       N.stream_ptr(keep[0].device)))
        return v

    def q_values_and_targets(self, batch: TransitionBatch) -> Dict[str, torch.Tensor]:
        """Parity probe: Q(s,a), max_a' Q_target(s',a') and the Bellman target of a batch."""
        if not self._fused:
            g, b = self._generic(), self._generic_batch(batch)
            g.ops.ensure(b["state"].shape[0])
            nv, y = g.targets(b, want_next_v=True)
            return {"q": g.q_values(b), "next_v": nv, "target": y}
        nb, keep = self._native_batch(batch)
        nat = self._ensure_bound(nb.B, nb.A)
        dev = keep[0].device
        q, v, y = (torch.empty(nb.B, dtype=torch.float32, device=dev) for _ in range(3))
        N.check(N.lib().pa_dqn_qvalues(nat.handle, C.byref(nb), q.data_ptr(), v.data_ptr(),
                                       y.data_ptr(), N.stream_ptr(dev)))
        return {"q": q, "next_v": v, "target": y}

    def _dp_world(self) -> int:
        return world_size() if self.data_parallel else 1

    # ------------------------------------------------------------------ CQL (experimental)
    def _cql_engine(self, nat: _NativeDqn, rows: int, dev: torch.device) -> C.c_void_p:
        """A generic pa_mlp handle over THIS learner's flat buffers (the two layouts coincide:
        W1 | b1 | W2 | b2 | W3 | b3, every offset rounded up to 4 floats): forward with kept
        activations, backward and AdamW for row counts and output gradients the fused DQN kernels
        do not take."""
        S, AD, H1, H2 = self._dims()
        opt = self._optimizer.param_groups[0]
        key = (id(nat.flat["q"]), rows, opt["lr"], tuple(opt["betas"]), opt["eps"],
               opt["weight_decay"], bool(opt["amsgrad"]))
        hit = getattr(nat, "cql", None)
        if hit is not None and hit[0] == key:
            return hit[1]
        if hit is not None:
            torch.cuda.synchronize(dev)
            N.lib().pa_mlp_destroy(hit[1])
        desc = N.MlpDesc(device=dev.index, n_layers=3, max_batch=rows, lr=opt["lr"],
                         beta1=opt["betas"][0], beta2=opt["betas"][1], eps=opt["eps"],
                         weight_decay=opt["weight_decay"], amsgrad=int(opt["amsgrad"]),
                         no_last_bias=0, identity_layers=0)
        for i, d in enumerate((S + AD, H1, H2, 1)):
            desc.dims[i] = d
        assert int(N.lib().pa_mlp_param_count(C.byref(desc))) == nat.flat["q"].numel()
        h = C.c_void_p()
        N.check(N.lib().pa_mlp_create(C.byref(h), C.byref(desc)))
        f = nat.flat
        bufs = N.MlpBuffers(p=f["q"].data_ptr(), p_target=f["q_target"].data_ptr(),
                            grad=f["grad"].data_ptr(), exp_avg=f["exp_avg"].data_ptr(),
                            exp_avg_sq=f["exp_avg_sq"].data_ptr(),
                            max_exp_avg_sq=f["max_exp_avg_sq"].data_ptr())
        N.check(N.lib().pa_mlp_bind(h, C.byref(bufs)))
        nat.cql = (key, h)
        return h

    def _learn_batch_conservative(self, batch: TransitionBatch) -> Dict[str, Any]:
        """DeepTDLearning.learn_batch with is_conservative (deep_td_learning.py:292-360):
        loss = mse(Q(s, a), y) + alpha * compute_cql_loss (loss_fn_utils.py:17-72).  One pass of the
        generic engine over B + B A rows — the taken pairs, then every (state, available action)
        pair — with `pa_cql_head` supplying the output gradients."""
        nb, keep = self._native_batch(batch)
        nat = self._ensure_bound(nb.B, nb.A)
        dev = keep[0].device
        lib, stream = N.lib(), N.stream_ptr(dev)
        state, action = keep[0], keep[1]
        B, S = state.shape
        AD = action.shape[1]
        assert batch.curr_available_actions is not None, "the CQL term needs curr_available_actions"
        rep = batch.curr_available_actions.to(dev, torch.float32).contiguous()
        A = int(rep.shape[-2])
        assert rep.ndim == 3 and rep.shape[0] == B and rep.shape[2] == AD
        if self._target_update_due():
            N.check(lib.pa_dqn_update_target(nat.handle, stream))
        y = torch.empty(B, dtype=torch.float32, device=dev)
        N.check(lib.pa_dqn_qvalues(nat.handle, C.byref(nb), None, None, y.data_ptr(), stream))
        R = B + B * A
        mlp = self._cql_engine(nat, R, dev)
        X = torch.empty(R, S + AD, dtype=torch.float32, device=dev)
        N.check(lib.pa_concat_cols(state.data_ptr(), state.stride(0), action.data_ptr(),
                                   action.stride(0), X.data_ptr(), B, S, AD, stream))
        N.check(lib.pa_expand_state_actions(state.data_ptr(), state.stride(0), rep.data_ptr(), A * AD,
                                            B, A, S, AD, X[B:].data_ptr(), stream))
        q_rows = torch.empty(R, dtype=torch.float32, device=dev)
        N.check(lib.pa_mlp_forward(mlp, 0, X.data_ptr(), X.stride(0), R, q_rows.data_ptr(), 1, 1,
                                   stream))
        dq = torch.empty(R, dtype=torch.float32, device=dev)
        losses = torch.empty(2, dtype=torch.float32, device=dev)     # mean |q - y| | total loss
        N.check(lib.pa_cql_head(q_rows.data_ptr(), y.data_ptr(), action.data_ptr(), action.stride(0),
                                B, A, AD, float(self._conservative_alpha), dq.data_ptr(),
                                losses.data_ptr(), stream))
        N.check(lib.pa_mlp_backward(mlp, X.data_ptr(), X.stride(0), R, dq.data_ptr(), 1, 1, None,
                                    S + AD, stream))
        step = self._adam_steps() + 1
        if self._dp_world() > 1:
            nat.flat["grad"].mul_(1.0 / self._dp_world())
            allreduce_sum_(nat.flat["grad"])
        N.check(lib.pa_mlp_adam(mlp, step, stream))
        self._set_adam_steps(step)
        loss = losses[0].item()
        if self._dp_world() > 1:
            from ... import _comm
            _comm.check_exchange_after_sync()
        return {"loss": loss}

    def learn_batch(self, batch: TransitionBatch) -> Dict[str, Any]:
        """One TD(0) update on a preprocessed batch (deep_td_learning.py:333-360)."""
        if not self._fused:
            return self._learn_batch_generic(batch)
        if self._is_conservative:
            return self._learn_batch_conservative(batch)
        nb, keep = self._native_batch(batch)
        nat = self._ensure_bound(nb.B, nb.A)
        dev = keep[0].device
        stream = N.stream_ptr(dev)
        step = self._adam_steps() + 1
        world = self._dp_world()
        N.check(N.lib().pa_dqn_step(nat.handle, C.byref(nb), int(self._target_update_due()), step,
                                    world, nat.loss_buf.data_ptr(), stream))
        if world > 1:
            allreduce_sum_(nat.flat["grad"])
            N.check(N.lib().pa_dqn_apply(nat.handle, step, stream))
        self._set_adam_steps(step)
        loss = nat.loss_buf[0].item()            # the reference's per-step .item() (:359)
        if world > 1:
            from ... import _comm
            _comm.check_exchange_after_sync()    # (P2P exchange: a peer that never answered)
        return {"loss": loss}

    def _arena_path_ok(self, replay_buffer: ReplayBuffer) -> bool:
        if not isinstance(replay_buffer, TensorBasedReplayBuffer) or replay_buffer.arena is None:
            return False
        if not self._fused:
            return False    # generic engine: sample -> preprocess -> learn_batch per round
        if int(self._double_q) == 2:
            return False    # SARSA batches carry the committed next action: generic loop
        if self._is_conservative:
            return False    # the CQL term is a per-batch pass of the generic engine
        rep = self.action_representation_module
        z = replay_buffer._layout
        onehot = isinstance(rep, OneHotActionTensorRepresentationModule)
        if not z.has_next_state or z.max_actions <= 0 or not replay_buffer._has_next_avail:
            return False
        if onehot:
            return z.action_elems == 1 and z.avail_dim == 1
        return type(rep).__name__ == "IdentityActionRepresentationModule" and \
            z.action_elems == self._Q.action_dim and z.avail_dim == self._Q.action_dim
        # anything else goes through the generic sample -> preprocess -> learn_batch loop

    # performance report (policy_learner.perf_reported): the fused loop's own event timers at level
    # 1 | 4 — the sampled target-pass and gather launches of the side stream and one mid-window
    # round's chain launches — which leave the overlapped two-stream loop as it is
    _PERF_STAGES = (("target", "target_pass"), ("gather", "gather"), ("gather_nox", "gather"),
                    ("rowpass", "row_pass"), ("bwd_dw", "weight_grad_adamw"))

    def _perf_begin(self) -> None:
        nat = getattr(self, "_native", None)
        mine = nat is not None and getattr(nat, "handle", None) is not None and self._fused
        self.__dict__["_perf_dqn"] = mine      # (a learner not bound yet reports no stages this call)
        if mine:
            N.check(N.lib().pa_dqn_enable_timing(nat.handle, 5))
        else:
            super()._perf_begin()

    def _perf_end(self) -> Dict[str, float]:
        if not self.__dict__.get("_perf_dqn"):
            return super()._perf_end()
        nat = self._native
        out: Dict[str, float] = {}
        for name, key in self._PERF_STAGES:
            ms, cnt = C.c_double(), C.c_int64()
            N.check(N.lib().pa_dqn_get_timing(nat.handle, name.encode(), C.byref(ms), C.byref(cnt)))
            if cnt.value:
                out[key] = ms.value * 1e3
        N.check(N.lib().pa_dqn_enable_timing(nat.handle, 0))
        return out

    @perf_reported
    def learn(self, replay_buffer: ReplayBuffer) -> Dict[str, Any]:
        """``training_rounds`` x (sample, preprocess, learn_batch) — policy_learner.py:162-195."""
        if len(replay_buffer) == 0:
            return {}
        if not self._arena_path_ok(replay_buffer):
            return super().learn(replay_buffer)
        self.__dict__["_pairs_memo"] = None
        self.__dict__["_pairs_memo"] = self._param_pairs()
        try:
            return self._learn_fused(replay_buffer)
        finally:
            self.__dict__["_pairs_memo"] = None

    def _learn_fused(self, replay_buffer: ReplayBuffer) -> Dict[str, Any]:
        batch_size = self._clamped_batch_size(replay_buffer)
        rounds = int(self._training_rounds)
        arena = replay_buffer.arena
        nat = self._ensure_bound(batch_size, arena.layout.max_actions)
        dev = arena.device
        # Parameters written through torch since the last call (load_state_dict copies in place; every
        # in-place torch op bumps the tensor's version counter, our kernels do not): the library's
        # derived copies are stale.  Otherwise back-to-back calls skip their rebuild.
        # (the Parameters alias the flat buffers' storage but keep their own version counters)
        ver = tuple(v for pq, pt in self._param_pairs() for v in (pq._version, pt._version))
        if ver != getattr(nat, "versions", None):
            N.check(N.lib().pa_dqn_invalidate(nat.handle))
            nat.versions = ver
        # learn() reports every round's loss: the optimizer kernel stores them straight into pinned,
        # device-mapped host memory (4 bytes per round over PCIe, fire-and-forget), so the end of the
        # call is one stream synchronisation — no device-to-host copy, no pageable staging
        self._loss_host(nat, rounds)
        onehot = isinstance(self.action_representation_module,
                            OneHotActionTensorRepresentationModule)
        if rounds == 0:
            return {}
        if self._dp_world() > 1 or (os.environ.get("PEARL_AMD_FORCE_DP") == "1"
                                    and dist.is_available() and dist.is_initialized()):
            # (the env override drives the hook path through RCCL with a single rank: a test aid)
            return self._learn_data_parallel(replay_buffer, batch_size, rounds, onehot)
        idx_host = None
        if replay_buffer.sampler == "python" or batch_size > replay_buffer.DEVICE_SAMPLER_MAX_B:
            n = len(replay_buffer)
            idx_host = np.asarray([random.sample(range(n), batch_size) for _ in range(rounds)],
                                  dtype=np.int64)
        args = N.LearnArgs(
            rounds=rounds, batch_size=batch_size, rep_onehot=int(onehot),
            target_update_freq=int(self._target_update_freq),
            training_steps0=int(self._training_steps), adam_step0=self._adam_steps(),
            seed=random.getrandbits(64) if idx_host is None else 0, offset0=0,
            losses_out=nat.loss_host.data_ptr(),
            idx_host=None if idx_host is None else idx_host.ctypes.data)
        N.check(N.lib().pa_dqn_learn(nat.handle, arena.handle, C.byref(args), N.stream_ptr(dev)))
        self._training_steps += rounds
        self._set_adam_steps(args.adam_step0 + rounds)
        torch.cuda.current_stream(dev).synchronize()   # the single host sync of this call
        losses = nat.loss_host[:rounds].tolist()
        N.check(N.lib().pa_dqn_check(nat.handle))
        return {"loss": losses}

    @staticmethod
    def _loss_host(nat: "_NativeDqn", rounds: int) -> torch.Tensor:
        if getattr(nat, "loss_host", None) is None or nat.loss_host.numel() < rounds:
            nat.loss_host = torch.zeros(max(2 * rounds, 1024), dtype=torch.float32, pin_memory=True)
        return nat.loss_host

    def _native_comm(self, dev: torch.device) -> Optional[C.c_void_p]:
        """The process's RCCL communicator for the native all-reduce hooks (pearl_amd/_comm.py: one
        per process, shared by every learner).  None when RCCL cannot be loaded, the process group
        is not RCCL-backed, or PEARL_AMD_TORCH_ALLREDUCE=1 (then torch.distributed.all_reduce
        hooks are used)."""
        from ... import _comm
        h = _comm.native_comm(dev)
        self._native.comm = h          # (bench.py reads it back; owned by _comm, never destroyed here)
        return h

    def _learn_data_parallel(self, replay_buffer: TensorBasedReplayBuffer, batch_size: int,
                             rounds: int, onehot: bool, force_world: Optional[int] = None
                             ) -> Dict[str, Any]:
        """world > 1: the same fused ``pa_dqn_learn`` loop with all-reduce hooks.  Every round the
        library scales the local gradient by 1/world, calls ``allreduce_start`` (an asynchronous
        ``torch.distributed.all_reduce`` = RCCL over xGMI on GPUs), enqueues the NEXT round's
        target-network pass, calls ``allreduce_wait`` and runs AdamW.  Ranks sample their own arena
        shard with their own index stream; parameters stay identical across ranks."""
        nat, arena = self._native, replay_buffer.arena
        dev = arena.device
        self._loss_host(nat, rounds)
        world = int(force_world) if force_world is not None else self._dp_world()
        grad = nat.flat["grad"]
        state: Dict[str, Any] = {"work": None, "error": None}

        def start(_ctx, _ptr, _n, _stream) -> int:
            try:
                if dist.is_available() and dist.is_initialized():
                    state["work"] = dist.all_reduce(grad, op=dist.ReduceOp.SUM, async_op=True)
                return 0
            except BaseException as e:  # never unwind through the C frame
                state["error"] = e
                return 1

        def wait(_ctx, _stream) -> int:
            try:
                w, state["work"] = state["work"], None
                if w is not None:
                    w.wait()
                return 0
            except BaseException as e:
                state["error"] = e
                return 1

        cb_start, cb_wait = N.ALLREDUCE_START_FN(start), N.ALLREDUCE_WAIT_FN(wait)
        fn_start, fn_wait, ctx = C.cast(cb_start, C.c_void_p), C.cast(cb_wait, C.c_void_p), None
        comm = self._native_comm(dev) if force_world is None else None
        if comm is not None:
            # native path: ncclAllReduce enqueued from C on the library's exchange stream — no
            # Python in the per-round loop
            lib = N.lib()
            fn_start = C.cast(lib.pa_comm_allreduce_start, C.c_void_p)
            fn_wait = C.cast(lib.pa_comm_allreduce_wait, C.c_void_p)
            ctx = comm
        idx_host = None
        if replay_buffer.sampler == "python" or batch_size > replay_buffer.DEVICE_SAMPLER_MAX_B:
            n = len(replay_buffer)
            idx_host = np.asarray([random.sample(range(n), batch_size) for _ in range(rounds)],
                                  dtype=np.int64)
        args = N.LearnArgs(
            rounds=rounds, batch_size=batch_size, rep_onehot=int(onehot),
            target_update_freq=int(self._target_update_freq),
            training_steps0=int(self._training_steps), adam_step0=self._adam_steps(),
            seed=random.getrandbits(64) if idx_host is None else 0, offset0=0,
            losses_out=nat.loss_host.data_ptr(),
            idx_host=None if idx_host is None else idx_host.ctypes.data,
            grad_world=world, allreduce_start=fn_start, allreduce_wait=fn_wait, allreduce_ctx=ctx)
        rc = N.lib().pa_dqn_learn(nat.handle, arena.handle, C.byref(args), N.stream_ptr(dev))
        if state["error"] is not None:
            raise state["error"]
        N.check(rc)
        self._training_steps += rounds
        self._set_adam_steps(args.adam_step0 + rounds)
        torch.cuda.current_stream(dev).synchronize()
        losses = nat.loss_host[:rounds].tolist()
        N.check(N.lib().pa_dqn_check(nat.handle))
        if comm is not None:
            from ... import _comm
            _comm.check_exchange()     # (P2P exchange: a peer that never answered is an exception here)
        return {"loss": losses}

    # ------------------------------------------------------------------ act / compare
    def act(self, subjective_state: torch.Tensor, available_action_space: Any,
            exploit: bool = False) -> Any:
        """Greedy action + exploration module (deep_td_learning.py:200-254).  Act-time only: a
        single (1, A, S+AD) forward through torch, not part of the learner hot path."""
        assert hasattr(available_action_space, "actions_batch")
        if subjective_state.ndim == 1:
            subjective_state = subjective_state.unsqueeze(0)
        with torch.no_grad():
            reps = self.action_representation_module(
                available_action_space.actions_batch.to(subjective_state)).unsqueeze(0)
            q_values = self._Q.get_q_values(subjective_state, reps.to(subjective_state.dtype))
            q_values = q_values.squeeze(0)
            exploit_action = available_action_space.actions[int(torch.argmax(q_values))]
        if exploit:
            return exploit_action
        return self.exploration_module.act(
            subjective_state=subjective_state, action_space=available_action_space,
            exploit_action=exploit_action, values=q_values)

    def act_many(self, states: torch.Tensor, available_action_space: Any,
                 exploit: bool = False) -> torch.Tensor:
        """``act`` for E states that share one action space (pearl_amd.vector_env): ONE
        (E, A, S + AD) forward, row-wise argmax, then the exploration module row by row in row
        order — epsilon-greedy consumes one ``random.random()`` per row and one
        ``action_space.sample()`` per exploring row, exactly the draws E successive ``act`` calls
        make (epsilon_greedy_exploration.py:28-102).  Returns the (E, *action_shape) actions on
        the states' device."""
        assert hasattr(available_action_space, "actions_batch") and states.ndim == 2
        E = int(states.shape[0])
        with torch.no_grad():
            table = available_action_space.actions_batch.to(states.device)      # (A, *action_shape)
            reps = self.action_representation_module(table.to(states)).to(states.dtype)
            q_values = self._Q.get_q_values(states, reps.unsqueeze(0).expand(E, *reps.shape))
            greedy = torch.argmax(q_values, dim=1)
            actions = table[greedy]
        if exploit:
            return actions
        ex = self.exploration_module
        if isinstance(ex, EGreedyExploration):
            if ex._epsilon_scheduling and ex.time_step < ex.warmup_steps:
                rows = []
                for e in range(E):      # the warm-up moves epsilon with every call
                    if ex.time_step < ex.warmup_steps:
                        frac = ex.time_step / ex.warmup_steps
                        ex.curr_epsilon = ex.start_epsilon + (ex.end_epsilon - ex.start_epsilon) * frac
                    ex.time_step += 1
                    if random.random() < ex.curr_epsilon:
                        rows.append(e)
            else:
                eps, rnd = ex.curr_epsilon, random.random
                rows = [e for e, u in enumerate([rnd() for _ in range(E)]) if u < eps]
                ex.time_step += E
            if rows:
                # (the sampled element back to its index: one small index copy instead of a stack
                #  of len(rows) action tensors)
                index_of = {id(a): k for k, a in enumerate(available_action_space.actions)}
                picks = [index_of[id(available_action_space.sample(None))] for _ in rows]
                actions = actions.clone()
                actions[torch.tensor(rows, device=actions.device)] = table[
                    torch.tensor(picks, device=actions.device)]
            return actions
        out = [ex.act(subjective_state=states[e:e + 1], action_space=available_action_space,
                      exploit_action=actions[e], values=q_values[e]) for e in range(E)]
        return torch.stack([torch.as_tensor(a).to(actions.device) for a in out])

    def compare(self, other: PolicyLearner) -> str:
        diffs = [super().compare(other)]
        if not isinstance(other, DeepQLearning):
            diffs.append("other is not an instance of DeepQLearning")
        else:
            for attr in ("_learning_rate", "_discount_factor", "_target_update_freq",
                         "_soft_update_tau", "_is_conservative", "_conservative_alpha"):
                if getattr(self, attr) != getattr(other, attr):
                    diffs.append(f"{attr} is different: {getattr(self, attr)} vs "
                                 f"{getattr(other, attr)}")
            for name in ("_Q", "_Q_target"):
                mine, theirs = getattr(self, name).state_dict(), getattr(other, name).state_dict()
                if mine.keys() != theirs.keys():
                    diffs.append(f"{name} is different: state_dict keys differ")
                    continue
                for k in mine:
                    if not torch.allclose(mine[k].cpu(), theirs[k].cpu(), rtol=1e-5, atol=1e-8):
                        diffs.append(f"{name} is different: key {k} differs")
        return "\n".join(d for d in diffs if d)
