"""SoftActorCritic for discrete action spaces
(reference: pearl/policy_learners/sequential_decision_making/soft_actor_critic.py:46-330).

Same constructor, buffers and ``learn_batch`` effects as the reference: a softmax actor
(``VanillaActorNetwork``), twin Q critics with target copies evaluated on EVERY available action of
a state, the expected-value Bellman target ``sum_a pi(a|s') (min Q'(s', a) - alpha log pi(a|s'))``
(:180-252), the actor objective ``mean(pi (alpha log pi - min Q))`` (:254-287), entropy-coefficient
autotuning with Adam(eps=1e-4) towards ``-0.89 log(1/n)`` (:103-151), and the actor learning-rate
decay (``ExponentialLR(0.99)`` stepped in ``reset``, :98-101, :128-130).

All arithmetic runs in libpearl_amd on flat parameter views.  The critics' values on every
available action come from ``pa_mlp_q_all`` — DQN's fused all-actions kernel (state half of layer 1
once per state, layers 2-3 out of one LDS tile per 64 (state, action) rows) — for the
[S+AD, <=256, <=256, 1] critics it supports; other shapes expand the (B, A, S+AD) input
(``pa_expand_state_actions``) and run the twin critics as paired launches over the B*A rows.  The
row-local softmax / expectation / gradient work is ``pa_dsac_actor_head`` / ``pa_dsac_target``, the
entropy step ``pa_sac_alpha_step``.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn, optim

from ... import _native as N
from ...action_representation_modules import ActionRepresentationModule
from ...neural_networks.sequential_decision_making.actor_networks import VanillaActorNetwork
from ...neural_networks.sequential_decision_making.q_value_networks import VanillaQValueNetwork
from ...replay_buffers.transition import TransitionBatch
from ..exploration import ExplorationModule, PropensityExploration
from ..policy_learner import PolicyLearner
from .actor_critic_base import ActorCriticBase
from .flat_mlp import FlatMlp, layers_of


class SoftActorCritic(ActorCriticBase):
    def __init__(self, action_space: Any, state_dim: Optional[int] = None,
                 actor_hidden_dims: Optional[List[int]] = None,
                 critic_hidden_dims: Optional[List[int]] = None,
                 actor_learning_rate: float = 1e-4, critic_learning_rate: float = 1e-4,
                 history_summarization_learning_rate: float = 1e-4,
                 actor_network_type: type = VanillaActorNetwork,
                 critic_network_type: type = VanillaQValueNetwork,
                 critic_soft_update_tau: float = 0.005,
                 exploration_module: Optional[ExplorationModule] = None,
                 discount_factor: float = 0.99, training_rounds: int = 100, batch_size: int = 128,
                 entropy_coef: float = 0.2, entropy_autotune: bool = True,
                 action_representation_module: Optional[ActionRepresentationModule] = None,
                 actor_network_instance: Optional[nn.Module] = None,
                 critic_network_instance: Optional[nn.Module] = None,
                 target_entropy_scale: float = 0.89, **kwargs: Any) -> None:
        if actor_network_type is not VanillaActorNetwork or critic_network_type is not VanillaQValueNetwork:
            raise NotImplementedError("pearl_amd SoftActorCritic: only VanillaActorNetwork + "
                                      "VanillaQValueNetwork twin critics have HIP kernels")
        super().__init__(
            state_dim=state_dim, action_space=action_space, actor_hidden_dims=actor_hidden_dims,
            critic_hidden_dims=critic_hidden_dims, actor_learning_rate=actor_learning_rate,
            critic_learning_rate=critic_learning_rate,
            history_summarization_learning_rate=history_summarization_learning_rate,
            actor_network_type=actor_network_type, critic_network_type=critic_network_type,
            use_actor_target=False, use_critic_target=True, actor_soft_update_tau=0.0,
            critic_soft_update_tau=critic_soft_update_tau, use_twin_critic=True,
            exploration_module=(exploration_module if exploration_module is not None
                                else PropensityExploration()),
            discount_factor=discount_factor, training_rounds=training_rounds, batch_size=batch_size,
            is_action_continuous=False, on_policy=False,
            action_representation_module=action_representation_module,
            actor_network_instance=actor_network_instance,
            critic_network_instance=critic_network_instance, **kwargs)
        # "needed to avoid actor softmax overflow" (:98-101); stepped in reset()
        self.scheduler = optim.lr_scheduler.ExponentialLR(self._actor_optimizer, gamma=0.99)
        self._entropy_autotune = entropy_autotune
        if entropy_autotune:
            self.register_parameter("_log_entropy", nn.Parameter(torch.zeros(1, requires_grad=True)))
            self._entropy_optimizer: optim.Optimizer = optim.Adam(
                [self._log_entropy], lr=self._critic_learning_rate, eps=1e-4)
            self.register_buffer("_entropy_coef", torch.exp(self._log_entropy).detach())
            assert hasattr(action_space, "n"), "SoftActorCritic needs a discrete action space"
            self.register_buffer("_target_entropy",
                                 -target_entropy_scale * torch.log(1.0 / torch.tensor(action_space.n)))
        else:
            self.register_buffer("_entropy_coef", torch.tensor(entropy_coef))
        self._neg_entropy_rows: Tensor = torch.tensor(0.0)   # sum_a P log(P + 1e-8) per row (:275-276)

    def reset(self, action_space: Any) -> None:
        self._action_space = action_space
        with warnings.catch_warnings():
            # (torch counts optimizer.step() calls to warn about the order; the actor's AdamW step runs in
            #  the HIP optimizer epilogue, torch's own step() is never called)
            warnings.filterwarnings("ignore", message="Detected call of `lr_scheduler.step\\(\\)` before")
            self.scheduler.step()

    # ------------------------------------------------------------------ flat views
    def _nets(self, batch_hint: int = 0, validate: bool = True):
        """(actor, critic 1, critic 2).  The critics see B * A rows per batch."""
        if not self._flat:
            mb = max(self._batch_size, 1)
            A = int(self.action_representation_module.max_number_actions)
            self._flat["actor"] = FlatMlp(layers_of(self._actor.linear_layers()),
                                          self._actor_optimizer, mb)
            for i, (c, ct) in enumerate(((self._critic._critic_1, self._critic_target._critic_1),
                                         (self._critic._critic_2, self._critic_target._critic_2)), 1):
                self._flat[f"critic{i}"] = FlatMlp(layers_of(c.linear_layers()),
                                                   self._critic_optimizer, mb * A,
                                                   target_layers=layers_of(ct.linear_layers()))
        A = self._flat["actor"].dims[-1]
        nets = (self._flat["actor"], self._flat["critic1"], self._flat["critic2"])
        hints = (batch_hint, batch_hint * A, batch_hint * A)
        if validate:
            return tuple(m.ensure(h) for m, h in zip(nets, hints))
        return tuple(m.ready(h) for m, h in zip(nets, hints))

    def _alpha_state(self, dev: torch.device) -> Dict[str, Any]:
        """Device scalars of the entropy coefficient and (autotune) its Adam state."""
        st = self._flat.get("alpha")
        if st is not None and st["alpha"].device == dev:
            return st
        st = {"alpha": self._entropy_coef.detach().to(dev, torch.float32).reshape(1).clone()}
        if self._entropy_autotune:
            ost = self._entropy_optimizer.state.get(self._log_entropy, {})
            for k in ("exp_avg", "exp_avg_sq"):
                st[k] = (ost[k].to(dev, torch.float32).reshape(1).clone() if k in ost
                         else torch.zeros(1, device=dev))
            st["step"] = int(float(ost["step"])) if "step" in ost else 0
            if self._log_entropy.device != dev:
                self._log_entropy.data = self._log_entropy.data.to(dev)
            self._entropy_optimizer.state[self._log_entropy] = {
                "step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"],
                "exp_avg_sq": st["exp_avg_sq"]}
        self._entropy_coef = st["alpha"].reshape(self._entropy_coef.shape)
        self._flat["alpha"] = st
        return st

    def _target_entropy_value(self) -> float:
        """The target entropy as a host float, read from the (device) buffer once per value — a
        per-step ``float(buffer)`` is a host synchronisation that stops ``learn()`` from running
        ahead of the device."""
        t = self._target_entropy
        hit = self._flat.get("target_entropy")
        if hit is None or hit[0] is not t or hit[1] != t._version:
            hit = (t, t._version, float(t))
            self._flat["target_entropy"] = hit
        return hit[2]

    @staticmethod
    def _f32(t: Tensor, dev: torch.device) -> Tensor:
        return t.to(device=dev, dtype=torch.float32).contiguous()

    def _all_action_input(self, state: Tensor, rep: Tensor) -> Tensor:
        """[B * A, S + AD]: every state next to each of its available actions' representation."""
        B, S = state.shape
        A, AD = int(rep.shape[-2]), int(rep.shape[-1])
        x = torch.empty(B * A, S + AD, dtype=torch.float32, device=state.device)
        N.check(N.lib().pa_expand_state_actions(
            state.data_ptr(), state.stride(0), rep.data_ptr(), A * AD if rep.ndim == 3 else 0, B, A,
            S, AD, x.data_ptr(), N.stream_ptr(state.device)))
        return x

    def _twin_q_all(self, c1: FlatMlp, c2: FlatMlp, state: Tensor, rep: Tensor, use_target: bool):
        """(q1, q2), each (B * A,): both critics on every (state, available action) pair."""
        if c1.supports_q_all(int(rep.shape[-2])) and c2.supports_q_all(int(rep.shape[-2])):
            if c1.dims == c2.dims:
                return FlatMlp.q_all_pair(c1, c2, state, rep, use_target=use_target)
            return (c1.q_all(state, rep, use_target=use_target),
                    c2.q_all(state, rep, use_target=use_target))
        q1, q2 = FlatMlp.forward_pair(c1, c2, self._all_action_input(state, rep),
                                      use_target=use_target)
        return q1.reshape(-1), q2.reshape(-1)

    @staticmethod
    def _mask_u8(mask: Optional[Tensor], dev: torch.device) -> Optional[Tensor]:
        """The (B, A) availability mask as bytes: a bool tensor is reinterpreted (same storage, no
        launch), anything else converted."""
        if mask is None:
            return None
        m = mask.to(dev).contiguous()
        return m.view(torch.uint8) if m.dtype == torch.bool else m.to(torch.uint8)

    # ------------------------------------------------------------------ losses
    def _actor_update(self, batch: TransitionBatch) -> Tensor:
        actor, c1, c2 = self._nets(len(batch))
        dev = actor.device
        al = self._alpha_state(dev)
        state = self._f32(batch.state, dev)
        B = state.shape[0]
        A = actor.dims[-1]
        s = N.stream_ptr(dev)
        assert batch.curr_available_actions is not None, "SoftActorCritic needs curr_available_actions"
        rep = self._f32(batch.curr_available_actions, dev)
        assert rep.shape[-2] == A, "the actor outputs one logit per available-action slot"
        # min Q(s, a) for every available action; the reference lets this loss reach the critics'
        # parameters too, then discards those gradients (actor_critic_base.py:342-348)
        q1, q2 = self._twin_q_all(c1, c2, state, rep, use_target=False)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        h = torch.empty(B, dtype=torch.float32, device=dev)
        mask = self._mask_u8(batch.curr_unavailable_actions_mask, dev)
        if N.lib().pa_rowstep_supported(actor.handle, None, A):
            # forward, the policy loss's row math and the backward pass in one launch
            d_logits = torch.empty(B, A, dtype=torch.float32, device=dev)
            actor.ready(B)
            N.check(N.lib().pa_dsac_actor_rowstep(
                actor.handle, state.data_ptr(), state.stride(0), B, q1.data_ptr(), q2.data_ptr(),
                N.ptr(mask), al["alpha"].data_ptr(), d_logits.data_ptr(), d_logits.stride(0),
                h.data_ptr(), loss.data_ptr(), s))
            actor._pending_x = (state, d_logits)
            self._neg_entropy_rows = h
            actor.adam()
            return loss[0]
        logits = actor.forward(state, keep=True)
        d_logits = torch.empty_like(logits)
        N.check(N.lib().pa_dsac_actor_head(logits.data_ptr(), logits.stride(0), q1.data_ptr(),
                                           q2.data_ptr(), N.ptr(mask), al["alpha"].data_ptr(), B, A,
                                           d_logits.data_ptr(), d_logits.stride(0), loss.data_ptr(),
                                           h.data_ptr(), s))
        self._neg_entropy_rows = h
        actor.backward(state, d_logits, want_dw=True, defer=True)
        actor.adam()
        return loss[0]

    def _critic_update(self, batch: TransitionBatch) -> Tensor:
        actor, c1, c2 = self._nets(len(batch), validate=False)
        dev = actor.device
        al = self._alpha_state(dev)
        state = self._f32(batch.state, dev)
        nstate = self._f32(batch.next_state, dev)
        B, S = state.shape
        A = actor.dims[-1]
        s = N.stream_ptr(dev)
        # ---- expected next-state value under the (already updated) policy (:180-252)
        assert batch.next_available_actions is not None, "SoftActorCritic needs next_available_actions"
        nrep = self._f32(batch.next_available_actions, dev)
        nq1, nq2 = self._twin_q_all(c1, c2, nstate, nrep, use_target=True)
        y = torch.empty(B, dtype=torch.float32, device=dev)
        reward = self._f32(batch.reward, dev).reshape(B)
        term = self._mask_u8(batch.terminated.reshape(B), dev)
        nmask = self._mask_u8(batch.next_unavailable_actions_mask, dev)
        if N.lib().pa_rowstep_supported(actor.handle, None, A):
            actor.ready(B)
            N.check(N.lib().pa_dsac_target_rowstep(
                actor.handle, nstate.data_ptr(), nstate.stride(0), B, nq1.data_ptr(), nq2.data_ptr(),
                N.ptr(nmask), al["alpha"].data_ptr(), reward.data_ptr(), term.data_ptr(),
                float(self._discount_factor), y.data_ptr(), s))
        else:
            nlogits = actor.forward(nstate)
            N.check(N.lib().pa_dsac_target(nlogits.data_ptr(), nlogits.stride(0), nq1.data_ptr(),
                                           nq2.data_ptr(), N.ptr(nmask), al["alpha"].data_ptr(),
                                           reward.data_ptr(), term.data_ptr(),
                                           float(self._discount_factor), B, A, y.data_ptr(), s))
        # ---- (mse(q1, y) + mse(q2, y)) / 2 on the taken action (critic_utils.py:170-203)
        act = self._f32(batch.action, dev).reshape(B, -1)
        AD = act.shape[1]
        xq = torch.empty(B, S + AD, dtype=torch.float32, device=dev)
        N.check(N.lib().pa_concat_cols(state.data_ptr(), state.stride(0), act.data_ptr(),
                                       act.stride(0), xq.data_ptr(), B, S, AD, s))
        if FlatMlp.rowstep_supported(c1, c2):
            # both critics' forward, MSE heads and backward in one launch (mlp_rowstep.hpp)
            loss = FlatMlp.mse_rowstep_pair(c1, c2, xq, y, 1.0 / B, 0.5)
            self._step_twin_critics(c1, c2)
            return loss[0]
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        qs = [q.reshape(B) for q in FlatMlp.forward_pair(c1, c2, xq, keep=True)]
        dqs = [torch.empty_like(q) for q in qs]
        for i in range(2):
            N.check(N.lib().pa_mse_head(qs[i].data_ptr(), 1, y.data_ptr(), B, 1.0 / B, 0.5, int(i > 0),
                                        dqs[i].data_ptr(), loss.data_ptr(), s))
        FlatMlp.backward_pair(c1, c2, xq, dqs[0], dqs[1], want_dw=True, defer=True)
        self._step_twin_critics(c1, c2)
        return loss[0]

    def _update_critic_target(self) -> None:
        _, c1, c2 = self._nets(validate=False)
        self._twin_target_update(c1, c2)

    # ------------------------------------------------------------------ one-call step
    def _one_call_ok(self, actor: FlatMlp, c1: FlatMlp, c2: FlatMlp) -> bool:
        """pa_dsac_step sequences the whole learn_batch in C.  It is the single-process step of
        exactly this class on networks the fused row steps and the all-actions kernel take; a
        data-parallel step all-reduces between backward and AdamW, and a subclass that overrides a
        stage keeps the per-stage path."""
        if os.environ.get("PEARL_AMD_DSAC_ONE_CALL", "1") == "0":
            return False
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False
        cls, base = type(self), SoftActorCritic
        if not (cls._actor_update is base._actor_update and cls._critic_update is base._critic_update
                and cls._update_critic_target is base._update_critic_target
                and self._use_critic and self._use_critic_target and not self._use_actor_target):
            return False
        if getattr(c1, "_steps", 0) != getattr(c2, "_steps", 0):
            return False     # pa_dsac_step steps both twins with c1's count (see ImplicitQLearning)
        memo = self._flat.get("one_call_ok")
        key = (actor.handle.value, c1.handle.value, c2.handle.value)
        if memo is None or memo[0] != key:
            A = actor.dims[-1]
            ok = (bool(N.lib().pa_rowstep_supported(actor.handle, None, A))
                  and FlatMlp.rowstep_supported(c1, c2) and c1.dims == c2.dims
                  and c1.supports_q_all(A) and c2.supports_q_all(A))
            memo = (key, ok)
            self._flat["one_call_ok"] = memo
        return memo[1]

    def _one_call_ws(self, dev: torch.device, B: int, A: int) -> Dict[str, Any]:
        ws = self._flat.get("one_call")
        if ws is None or ws["key"] != (dev, B, A):
            n = int(N.lib().pa_dsac_scratch_floats(B, A))
            ws = {"key": (dev, B, A), "scratch": torch.empty(n, dtype=torch.float32, device=dev),
                  "args": N.DsacStepArgs()}
            self._flat["one_call"] = ws
        return ws

    def _step_args(self, ws: Dict[str, Any], actor: FlatMlp, c1: FlatMlp, c2: FlatMlp, B: int, S: int,
                   AD: int, losses: Tensor, h: Tensor) -> "N.DsacStepArgs":
        """pa_dsac_step_args of the next step (batch pointers left to the caller); advances the
        entropy optimizer's step count by one."""
        dev = actor.device
        al = self._alpha_state(dev)
        a = ws["args"]
        a.actor, a.critic1, a.critic2 = actor.handle.value, c1.handle.value, c2.handle.value
        a.B, a.S, a.A, a.AD = B, S, actor.dims[-1], AD
        a.gamma, a.tau = float(self._discount_factor), float(self._critic_soft_update_tau)
        a.alpha = al["alpha"].data_ptr()
        if self._entropy_autotune:
            g = self._entropy_optimizer.param_groups[0]
            al["step"] += 1
            a.log_alpha = self._log_entropy.data.data_ptr()
            a.alpha_m, a.alpha_v = al["exp_avg"].data_ptr(), al["exp_avg_sq"].data_ptr()
            a.target_entropy = self._target_entropy_value()
            a.alpha_lr, a.alpha_beta1, a.alpha_beta2 = g["lr"], g["betas"][0], g["betas"][1]
            a.alpha_eps, a.alpha_weight_decay = g["eps"], g["weight_decay"]
            a.alpha_step = al["step"]
        else:
            a.log_alpha = None
        a.actor_step, a.critic_step = actor.next_adam_step(), c1.next_adam_step()
        a.scratch, a.losses, a.h_out = ws["scratch"].data_ptr(), losses.data_ptr(), h.data_ptr()
        return a

    def _learn_batch_one_call(self, batch: TransitionBatch, actor: FlatMlp, c1: FlatMlp,
                              c2: FlatMlp) -> Dict[str, Any]:
        dev = actor.device
        state = self._f32(batch.state, dev)
        nstate = self._f32(batch.next_state, dev)
        B, S = state.shape
        A = actor.dims[-1]
        s = N.stream_ptr(dev)
        assert batch.curr_available_actions is not None, "SoftActorCritic needs curr_available_actions"
        assert batch.next_available_actions is not None, "SoftActorCritic needs next_available_actions"
        rep = self._f32(batch.curr_available_actions, dev)
        nrep = self._f32(batch.next_available_actions, dev)
        assert rep.shape[-2] == A, "the actor outputs one logit per available-action slot"
        act = self._f32(batch.action, dev).reshape(B, -1)
        AD = act.shape[1]
        xq = torch.empty(B, S + AD, dtype=torch.float32, device=dev)
        N.check(N.lib().pa_concat_cols(state.data_ptr(), state.stride(0), act.data_ptr(),
                                       act.stride(0), xq.data_ptr(), B, S, AD, s))
        reward = self._f32(batch.reward, dev).reshape(B)
        term = self._mask_u8(batch.terminated.reshape(B), dev)
        mask = self._mask_u8(batch.curr_unavailable_actions_mask, dev)
        nmask = self._mask_u8(batch.next_unavailable_actions_mask, dev)
        ws = self._one_call_ws(dev, B, A)
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        h = torch.empty(B, dtype=torch.float32, device=dev)
        a = self._step_args(ws, actor, c1, c2, B, S, AD, losses, h)
        a.state, a.ld_state = state.data_ptr(), state.stride(0)
        a.next_state, a.ld_next_state = nstate.data_ptr(), nstate.stride(0)
        a.xq, a.ld_xq = xq.data_ptr(), xq.stride(0)
        a.reward, a.terminated = reward.data_ptr(), term.data_ptr()
        a.curr_rep, a.curr_rep_bstride = rep.data_ptr(), (A * AD if rep.ndim == 3 else 0)
        a.next_rep, a.next_rep_bstride = nrep.data_ptr(), (A * AD if nrep.ndim == 3 else 0)
        a.curr_mask, a.next_mask = N.ptr(mask), N.ptr(nmask)
        N.check(N.lib().pa_dsac_step(C.byref(a), s))
        for m in (actor, c1, c2):
            m.stepped_natively()
        self._neg_entropy_rows = h
        report = {"actor_loss": losses[0], "critic_loss": losses[1]}
        if self._entropy_autotune:
            self._entropy_optimizer.state[self._log_entropy]["step"].fill_(
                float(self._alpha_state(dev)["step"]))
            report["entropy_coef"] = losses[2]
        return report

    def _learn_native_loop(self, replay_buffer: Any, batch_size: int) -> Optional[Dict[str, List[Any]]]:
        """learn() as ONE pa_dsac_learn call: every round's gather + step sequenced in C.  The
        rounds are the ones the per-round loop would run — same index lists, same kernels; one
        gather launch per group of rounds writes the learner-side views directly (state || one-hot
        action rows, the one-hot representation of both availability tables), so the per-round
        one-hot and concat launches are gone as well.  None: this call takes the per-round loop."""
        from ...action_representation_modules import OneHotActionTensorRepresentationModule
        from ...replay_buffers.basic_replay_buffer import TensorBasedReplayBuffer
        from ..policy_learner import IdentityHistorySummarizationModule
        if os.environ.get("PEARL_AMD_AC_LOOP", "1") == "0":
            return None
        cls, base = type(self), SoftActorCritic
        rb = replay_buffer
        if cls._learn_batch_device is not base._learn_batch_device \
                or cls._learn_batch_one_call is not base._learn_batch_one_call \
                or cls.preprocess_batch is not ActorCriticBase.preprocess_batch \
                or cls._preprocess_for_learn is not ActorCriticBase._preprocess_for_learn \
                or cls.learn_batch is not ActorCriticBase.learn_batch \
                or not isinstance(rb, TensorBasedReplayBuffer) or rb.arena is None \
                or type(rb).sample is not TensorBasedReplayBuffer.sample \
                or type(rb)._gather_batch is not TensorBasedReplayBuffer._gather_batch:
            return None
        arm = self.action_representation_module
        if type(arm) is not OneHotActionTensorRepresentationModule \
                or type(self._history_summarization_module) is not IdentityHistorySummarizationModule \
                or hasattr(getattr(self, "safety_module", None), "lambda_constraint"):
            return None
        actor, c1, c2 = self._nets(batch_size)
        if not self._one_call_ok(actor, c1, c2):
            return None
        dev = actor.device
        B, S, A = int(batch_size), actor.dims[0], actor.dims[-1]
        AD = int(arm.max_number_actions)
        rounds = int(self._training_rounds)
        pre, z, arena = rb._presampled, rb._layout, rb.arena
        if pre is None or pre[1] != 0 or tuple(pre[0].shape) != (rounds, B) or rounds <= 0 \
                or arena.device != dev or rb._device_for_batches != dev \
                or len(z.state_shape) > 1 or z.state_dim != S or not z.has_next_state or z.has_cost \
                or z.action_elems != 1 or z.action_dtype.is_floating_point or z.avail_dim != 1 \
                or z.max_actions != A or rb._is_action_continuous \
                or not (rb._has_curr_avail and rb._has_next_avail) or c1.dims[0] != S + AD:
            return None
        row_bytes = 4 * (2 * S + (S + AD) + 1 + 2 * A * AD) + 1 + 2 * A
        G = max(1, min(rounds, self._LOOP_GATHER_BYTES // (row_bytes * B), len(rb) // B))
        ws = self._flat.get("loop_ws")
        key = (dev, B, S, A, AD, G)
        if ws is None or ws["key"] != key:
            n = G * B

            def new(shape, dtype=torch.float32):
                return torch.empty(shape, dtype=dtype, device=dev)
            ws = {"key": key, "state": new((n, S)), "next": new((n, S)), "x": new((n, S + AD)),
                  "reward": new((n,)), "term": new((n,), torch.uint8),
                  "cmask": new((n, A), torch.uint8), "nmask": new((n, A), torch.uint8),
                  "crep": new((n, A, AD)), "nrep": new((n, A, AD))}
            self._flat["loop_ws"] = ws
        lp = N.AcLoopArgs()
        o = lp.batch
        o.state, o.next_state, o.x = ws["state"].data_ptr(), ws["next"].data_ptr(), ws["x"].data_ptr()
        o.reward_f32, o.terminated = ws["reward"].data_ptr(), ws["term"].data_ptr()
        o.curr_mask, o.next_mask = ws["cmask"].data_ptr(), ws["nmask"].data_ptr()
        o.curr_avail_rep, o.next_avail_rep = ws["crep"].data_ptr(), ws["nrep"].data_ptr()
        o.rep_dim, o.rep_onehot = AD, 1
        losses = self._loop_losses(rounds, 3)
        h = torch.empty(B, dtype=torch.float32, device=dev)
        one = self._one_call_ws(dev, B, A)
        a = self._step_args(one, actor, c1, c2, B, S, AD, losses, h)
        a.state, a.ld_state, a.next_state, a.ld_next_state = o.state, S, o.next_state, S
        a.xq, a.ld_xq = o.x, S + AD
        a.reward, a.terminated = o.reward_f32, o.terminated
        a.curr_rep, a.curr_rep_bstride, a.next_rep, a.next_rep_bstride = (
            o.curr_avail_rep, A * AD, o.next_avail_rep, A * AD)
        a.curr_mask, a.next_mask = o.curr_mask, o.next_mask
        lp.rounds, lp.gather_rounds = rounds, G
        lp.idx_lists = pre[0].data_ptr()
        lp.losses, lp.losses_stride = losses.data_ptr(), 3
        N.check(N.lib().pa_dsac_learn(C.byref(a), arena.handle, C.byref(lp), N.stream_ptr(dev)))
        for m in (actor, c1, c2):
            m.stepped_natively(rounds)
        if self._entropy_autotune:
            self._alpha_state(dev)["step"] += rounds - 1       # (_step_args counted the first)
        self._training_steps += rounds
        rb._presampled = (pre[0], rounds, len(rb))              # all consumed
        rb._last_idx = pre[0][rounds - 1]
        self._neg_entropy_rows = h
        torch.cuda.current_stream(dev).synchronize()           # the single host sync of this call
        got = [losses[:, k].tolist() for k in range(3)]
        report: Dict[str, List[Any]] = {"actor_loss": got[0], "critic_loss": got[1]}
        if self._entropy_autotune:
            self._entropy_optimizer.state[self._log_entropy]["step"].fill_(
                float(self._alpha_state(dev)["step"]))
            report["entropy_coef"] = got[2]
        return report

    def _learn_batch_device(self, batch: TransitionBatch) -> Dict[str, Any]:
        actor, c1, c2 = self._nets(len(batch))
        if self._one_call_ok(actor, c1, c2):
            return self._learn_batch_one_call(batch, actor, c1, c2)
        report = super()._learn_batch_device(batch)
        if self._entropy_autotune:
            actor, _, _ = self._nets(validate=False)
            dev = actor.device
            al = self._alpha_state(dev)
            g = self._entropy_optimizer.param_groups[0]
            al["step"] += 1
            loss = torch.empty(1, dtype=torch.float32, device=dev)
            # loss = exp(log_alpha) (entropy - target), entropy = -mean_b sum_a P log(P + 1e-8)
            #      = mean_b( -exp(log_alpha) (h_b + target) ): pa_sac_alpha_step's form with h as
            #        the "log-prob" column (:134-151)
            h = self._neg_entropy_rows
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                h = h.mean().reshape(1)
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                h = (h / dist.get_world_size()).contiguous()
            N.check(N.lib().pa_sac_alpha_step(
                self._log_entropy.data.data_ptr(), al["exp_avg"].data_ptr(),
                al["exp_avg_sq"].data_ptr(), None, al["alpha"].data_ptr(), h.data_ptr(),
                int(h.numel()), self._target_entropy_value(), g["lr"], g["betas"][0], g["betas"][1],
                g["eps"], g["weight_decay"], 0, al["step"], loss.data_ptr(), N.stream_ptr(dev)))
            self._entropy_optimizer.state[self._log_entropy]["step"].fill_(float(al["step"]))
            report = {**report, "entropy_coef": loss[0]}
        return report

    def act(self, subjective_state: Tensor, available_action_space: Any, exploit: bool = False) -> Any:
        """actor_critic_base.py:245-303 (act-time only; torch expression of the same network)."""
        with torch.no_grad():
            probs = self._actor.get_policy_distribution(
                state_batch=subjective_state,
                available_actions=self.action_representation_module(
                    available_action_space.actions_batch.to(subjective_state.device)))
            exploit_action = available_action_space.actions[int(torch.argmax(probs))]
        if exploit:
            return exploit_action
        return self.exploration_module.act(exploit_action=exploit_action,
                                           action_space=available_action_space,
                                           subjective_state=subjective_state, values=probs)

    def compare(self, other: PolicyLearner) -> str:
        diffs = [super().compare(other)]
        if not isinstance(other, SoftActorCritic):
            diffs.append("other is not an instance of SoftActorCritic")
        else:
            if self._entropy_autotune != other._entropy_autotune:
                diffs.append(f"_entropy_autotune is different: {self._entropy_autotune} vs "
                             f"{other._entropy_autotune}")
            if not torch.allclose(self._entropy_coef.cpu(), other._entropy_coef.cpu()):
                diffs.append(f"_entropy_coef is different: {self._entropy_coef} vs "
                             f"{other._entropy_coef}")
            if self._entropy_autotune and not torch.allclose(self._target_entropy.cpu(),
                                                             other._target_entropy.cpu()):
                diffs.append(f"_target_entropy is different: {self._target_entropy} vs "
                             f"{other._target_entropy}")
        return "\n".join(d for d in diffs if d)
