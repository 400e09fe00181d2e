"""``ContinuousSoftActorCritic`` on HIP.

Mirror of pearl/policy_learners/sequential_decision_making/soft_actor_critic_continuous.py:40-268
(twin-Q SAC with a tanh-Gaussian actor and entropy autotune), same constructor and defaults.

One ``learn_batch`` (actor_critic_base.py:309-366 + :131-153 here) is, on the device:

  actor:   head = actor(s) ; (a, log pi) = pa_gauss_sample(head, noise)        actor_networks.py:551-591
           q1, q2 = critics(s || a) ; loss = mean(alpha log pi - min(q1, q2))   :208-231   (pa_sac_twin 0)
           d a through both critics (pa_mlp_backward, input gradient only), pa_gauss_actor_grad,
           actor backward + AdamW(amsgrad)
  critic:  (a', log pi') = sample(actor(s'))  with the UPDATED actor            :178-206
           y = (min(q1', q2')_target - alpha log pi') * gamma * (1 - term) + r   :155-176   (pa_sac_twin 1)
           loss = (mse(q1, y) + mse(q2, y)) / 2 ; backward + AdamW on both critics
  targets: critic_target <- tau critic + (1 - tau) critic_target
  alpha:   AdamW on log_alpha with mean(-exp(log_alpha)(log pi + target_entropy))  :134-151

The reparameterisation noise is the one torch-side input: ``torch.randn`` on the learner's device
(or ``noise_source`` for parity tests, which replays the reference's draws).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Callable, Dict, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn, optim

from ... import _native as N
from ...action_representation_modules import ActionRepresentationModule
from ...neural_networks.sequential_decision_making.actor_networks import GaussianActorNetwork
from ...neural_networks.sequential_decision_making.q_value_networks import VanillaQValueNetwork
from ...replay_buffers.transition import TransitionBatch
from ..exploration import ExplorationModule, NoExploration
from ..policy_learner import PolicyLearner
from .actor_critic_base import ActorCriticBase
from .flat_mlp import FlatMlp, layers_of


class ContinuousSoftActorCritic(ActorCriticBase):
    def __init__(self, action_space: Any, state_dim: Optional[int] = None,
                 actor_hidden_dims: Optional[List[int]] = None,
                 critic_hidden_dims: Optional[List[int]] = None, actor_learning_rate: float = 1e-3,
                 critic_learning_rate: float = 1e-3,
                 history_summarization_learning_rate: float = 1e-3,
                 actor_network_type: type = GaussianActorNetwork,
                 critic_network_type: type = VanillaQValueNetwork,
                 critic_soft_update_tau: float = 0.005,
                 exploration_module: Optional[ExplorationModule] = None,
                 discount_factor: float = 0.99, training_rounds: int = 100, batch_size: int = 256,
                 entropy_coef: float = 0.2, entropy_autotune: bool = True,
                 action_representation_module: Optional[ActionRepresentationModule] = None,
                 actor_network_instance: Optional[nn.Module] = None,
                 critic_network_instance: Optional[nn.Module] = None, **kwargs: Any) -> None:
        if actor_network_type is not GaussianActorNetwork or critic_network_type is not VanillaQValueNetwork:
            raise NotImplementedError("pearl_amd SAC: only GaussianActorNetwork actors and "
                                      "VanillaQValueNetwork critics have HIP kernels")
        super().__init__(
            state_dim=state_dim, action_space=action_space, actor_hidden_dims=actor_hidden_dims,
            critic_hidden_dims=critic_hidden_dims, actor_learning_rate=actor_learning_rate,
            critic_learning_rate=critic_learning_rate,
            history_summarization_learning_rate=history_summarization_learning_rate,
            actor_network_type=actor_network_type, critic_network_type=critic_network_type,
            use_actor_target=False, use_critic_target=True, actor_soft_update_tau=0.0,
            critic_soft_update_tau=critic_soft_update_tau, use_twin_critic=True,
            exploration_module=(exploration_module if exploration_module is not None
                                else NoExploration()),
            discount_factor=discount_factor, training_rounds=training_rounds, batch_size=batch_size,
            is_action_continuous=True, on_policy=False,
            action_representation_module=action_representation_module,
            actor_network_instance=actor_network_instance,
            critic_network_instance=critic_network_instance, **kwargs)
        self._entropy_autotune = entropy_autotune
        if entropy_autotune:
            self.register_parameter("_log_entropy", nn.Parameter(torch.zeros(1, requires_grad=True)))
            self._entropy_optimizer: optim.Optimizer = optim.AdamW(
                [self._log_entropy], lr=self._critic_learning_rate, amsgrad=True)
            self.register_buffer("_entropy_coef", torch.exp(self._log_entropy).detach())
            self.register_buffer("_target_entropy", -torch.tensor(action_space.shape[0]))
        else:
            self.register_buffer("_entropy_coef", torch.tensor(entropy_coef))
        self._action_batch_log_prob_cache: Tensor = torch.tensor(0.0)
        # parity hook: callable (B, A, device) -> standard-normal tensor; default torch.randn.
        # A source that also has `.rounds(first_round, n, B, A, device) -> [n, 2, B, A]` (actor
        # draw, then critic-target draw, per round — the order the reference consumes torch's
        # generator in) keeps learn() on the native loop.
        self.noise_source: Optional[Callable[[int, int, torch.device], Tensor]] = None

    # ------------------------------------------------------------------ flat views
    def _nets(self, batch_hint: int = 0, validate: bool = True):
        """The flat networks.  `validate` re-checks that the torch parameters and optimizer state
        still alias the flat buffers (once per learn_batch, in _actor_update); the later stages of
        the same learn_batch only need bound handles."""
        if not self._flat:
            mb = max(self._batch_size, 1)
            a = self._actor
            head = ([a.fc_mu.weight, a.fc_std.weight], [a.fc_mu.bias, a.fc_std.bias])
            self._flat["actor"] = FlatMlp(layers_of(a.trunk_layers()) + [head],
                                          self._actor_optimizer, mb)
            for i, (c, ct) in enumerate(((self._critic._critic_1, self._critic_target._critic_1),
                                         (self._critic._critic_2, self._critic_target._critic_2)), 1):
                self._flat[f"critic{i}"] = FlatMlp(layers_of(c.linear_layers()),
                                                   self._critic_optimizer, mb,
                                                   target_layers=layers_of(ct.linear_layers()))
        nets = (self._flat["actor"], self._flat["critic1"], self._flat["critic2"])
        if validate:
            return tuple(m.ensure(batch_hint) for m in nets)
        return tuple(m.ready(batch_hint) for m in nets)

    def _alpha_state(self, dev: torch.device) -> Dict[str, Tensor]:
        """Device scalars of the entropy coefficient and (autotune) its AdamW state."""
        st = self._flat.get("alpha")
        if st is not None and st["alpha"].device == dev:
            return st
        st = {"alpha": self._entropy_coef.detach().to(dev, torch.float32).reshape(1).clone()}
        if self._entropy_autotune:
            ost = self._entropy_optimizer.state.get(self._log_entropy, {})
            for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                st[k] = (ost[k].to(dev, torch.float32).reshape(1).clone() if k in ost
                         else torch.zeros(1, device=dev))
            st["step"] = int(float(ost["step"])) if "step" in ost else 0
            if self._log_entropy.device != dev:
                self._log_entropy.data = self._log_entropy.data.to(dev)
            self._entropy_optimizer.state[self._log_entropy] = {
                "step": torch.tensor(float(st["step"])), "exp_avg": st["exp_avg"],
                "exp_avg_sq": st["exp_avg_sq"], "max_exp_avg_sq": st["max_exp_avg_sq"]}
        self._entropy_coef = st["alpha"].reshape(self._entropy_coef.shape)
        self._flat["alpha"] = st
        return st

    def _noise(self, B: int, A: int, dev: torch.device) -> Tensor:
        if self.noise_source is not None:
            return self.noise_source(B, A, dev).to(dev, torch.float32).contiguous()
        return torch.randn(B, A, device=dev, dtype=torch.float32)

    def _bounds(self, dev: torch.device):
        sp = self._actor._action_space
        hit = self._flat.get("bounds")
        if hit is None or hit[0] is not sp or hit[1] != dev:
            hit = (sp, dev, sp.low.to(dev, torch.float32).contiguous(),
                   sp.high.to(dev, torch.float32).contiguous())
            self._flat["bounds"] = hit
        return hit[2], hit[3]

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

    def _sample(self, actor: FlatMlp, state: Tensor, xa: Tensor, keep: bool):
        """head = actor(state); writes the sampled action into xa[:, S:]; returns (head, noise,
        log_prob)."""
        dev = state.device
        B, S = state.shape
        A = actor.dims[-1] // 2
        head = actor.forward(state, keep=keep)
        noise = self._noise(B, A, dev)
        low, high = self._bounds(dev)
        logp = torch.empty(B, dtype=torch.float32, device=dev)
        act_view = xa[:, S:]
        N.check(N.lib().pa_gauss_sample(head.data_ptr(), head.stride(0), noise.data_ptr(),
                                        noise.stride(0), low.data_ptr(), high.data_ptr(), B, A,
                                        act_view.data_ptr(), xa.stride(0), logp.data_ptr(),
                                        N.stream_ptr(dev)))
        return head, noise, logp

    # ------------------------------------------------------------------ losses
    def _actor_update(self, batch: TransitionBatch) -> Tensor:
        actor, c1, c2 = self._nets(len(batch))
        dev = actor.device
        al = self._alpha_state(dev)
        state = self._f32(batch.state, dev)
        B, S = state.shape
        A = actor.dims[-1] // 2
        s = N.stream_ptr(dev)
        xa = torch.empty(B, S + A, dtype=torch.float32, device=dev)
        xa[:, :S].copy_(state)
        head, noise, logp = self._sample(actor, state, xa, keep=True)
        self._action_batch_log_prob_cache = logp
        q1, q2 = FlatMlp.forward_pair(c1, c2, xa, keep=True)   # twin critics: one launch per layer
        q1, q2 = q1.reshape(B), q2.reshape(B)
        dq1, dq2 = torch.empty_like(q1), torch.empty_like(q2)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        N.check(N.lib().pa_sac_twin(0, q1.data_ptr(), q2.data_ptr(), logp.data_ptr(),
                                    al["alpha"].data_ptr(), None, None, 0.0, B, dq1.data_ptr(),
                                    dq2.data_ptr(), loss.data_ptr(), s))
        # the reference also forms (then discards) the critics' parameter gradients here
        # (actor_critic_base.py:342-348); only the input gradient matters
        dx1, dx2 = FlatMlp.backward_pair(c1, c2, xa, dq1, dq2, want_dw=False, want_dx=True)
        d_head = torch.empty_like(head)
        low, high = self._bounds(dev)
        N.check(N.lib().pa_gauss_actor_grad(
            head.data_ptr(), head.stride(0), noise.data_ptr(), noise.stride(0), low.data_ptr(),
            high.data_ptr(), dx1[:, S:].data_ptr(), dx2[:, S:].data_ptr(), dx1.stride(0),
            al["alpha"].data_ptr(), B, A, d_head.data_ptr(), d_head.stride(0), s))
        actor.backward(state, d_head, want_dw=True, defer=True)
        actor.adam()
        return loss[0]

    def _critic_update(self, batch: TransitionBatch) -> Tensor:
        actor, c1, c2 = self._nets(len(batch), validate=False)
        dev = actor.device
        al = self._alpha_state(dev)
        state = self._f32(batch.state, dev)
        nstate = self._f32(batch.next_state, dev)
        B, S = state.shape
        A = actor.dims[-1] // 2
        s = N.stream_ptr(dev)
        # ---- Bellman target with the updated actor and the target critics (:178-206)
        xn = torch.empty(B, S + A, dtype=torch.float32, device=dev)
        xn[:, :S].copy_(nstate)
        _, _, nlogp = self._sample(actor, nstate, xn, keep=False)
        nq1, nq2 = FlatMlp.forward_pair(c1, c2, xn, use_target=True)
        nq1, nq2 = nq1.reshape(B), nq2.reshape(B)
        y = torch.empty(B, dtype=torch.float32, device=dev)
        reward = self._f32(batch.reward, dev).reshape(B)
        term = batch.terminated.to(dev).reshape(B).to(torch.uint8).contiguous()
        N.check(N.lib().pa_sac_twin(1, nq1.data_ptr(), nq2.data_ptr(), nlogp.data_ptr(),
                                    al["alpha"].data_ptr(), reward.data_ptr(), term.data_ptr(),
                                    float(self._discount_factor), B, y.data_ptr(), None, None, s))
        # ---- (mse(q1, y) + mse(q2, y)) / 2 (critic_utils.py:170-203)
        xq = torch.empty(B, S + A, dtype=torch.float32, device=dev)
        act = self._f32(batch.action, dev).reshape(B, A)
        N.check(N.lib().pa_concat_cols(state.data_ptr(), state.stride(0), act.data_ptr(),
                                       act.stride(0), xq.data_ptr(), B, S, A, s))
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
    def _one_call_ok(self) -> bool:
        """pa_sac_step sequences the whole learn_batch in C.  It is the single-process step of
        exactly this class: a data-parallel step all-reduces between backward and AdamW, and a
        subclass that overrides a stage keeps the per-stage path."""
        if os.environ.get("PEARL_AMD_SAC_ONE_CALL", "1") == "0":
            return False
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False
        cls = type(self)
        base = ContinuousSoftActorCritic
        return (cls._actor_update is base._actor_update and cls._critic_update is base._critic_update
                and cls._update_critic_target is base._update_critic_target
                and self._use_critic and self._use_critic_target and not self._use_actor_target)

    def _step_args(self, ws: Dict[str, Any], actor: FlatMlp, c1: FlatMlp, c2: FlatMlp,
                   state: Tensor, act: Tensor, reward: Tensor, term: Tensor, nstate: Tensor,
                   losses: Tensor, logp: Tensor) -> "N.SacStepArgs":
        """pa_sac_step_args of the next step on these batch tensors (noise pointers left to the
        caller); advances the entropy optimizer's step count by one."""
        dev = actor.device
        al = self._alpha_state(dev)
        B, S = state.shape
        A = actor.dims[-1] // 2
        low, high = self._bounds(dev)
        a = ws["args"]
        a.actor, a.critic1, a.critic2 = actor.handle, c1.handle, c2.handle
        a.state, a.ld_state = state.data_ptr(), state.stride(0)
        a.action, a.ld_action = act.data_ptr(), act.stride(0)
        a.reward, a.terminated = reward.data_ptr(), term.data_ptr()
        a.next_state, a.ld_next_state = nstate.data_ptr(), nstate.stride(0)
        a.low, a.high = low.data_ptr(), high.data_ptr()
        a.alpha = al["alpha"].data_ptr()
        if self._entropy_autotune:
            g = self._entropy_optimizer.param_groups[0]
            al["step"] += 1
            a.log_alpha = self._log_entropy.data.data_ptr()
            a.alpha_m, a.alpha_v = al["exp_avg"].data_ptr(), al["exp_avg_sq"].data_ptr()
            a.alpha_vmax = al["max_exp_avg_sq"].data_ptr()
            a.target_entropy = self._target_entropy_value()
            a.alpha_lr, a.alpha_beta1, a.alpha_beta2 = g["lr"], g["betas"][0], g["betas"][1]
            a.alpha_eps, a.alpha_weight_decay = g["eps"], g["weight_decay"]
            a.alpha_amsgrad, a.alpha_step = int(bool(g["amsgrad"])), al["step"]
        else:
            a.log_alpha = None
        a.B, a.S, a.A = B, S, A
        a.gamma, a.tau = float(self._discount_factor), float(self._critic_soft_update_tau)
        a.actor_step, a.critic_step = actor._steps + 1, c1._steps + 1
        a.scratch, a.losses, a.log_prob_out = ws["scratch"].data_ptr(), losses.data_ptr(), logp.data_ptr()
        return a

    def _one_call_ws(self, dev: torch.device, B: int, S: int, A: int) -> Dict[str, Any]:
        ws = self._flat.get("one_call")
        if ws is None or ws["key"] != (dev, B, S, A):
            n = int(N.lib().pa_sac_scratch_floats(B, S, A))
            ws = {"key": (dev, B, S, A), "scratch": torch.empty(n, dtype=torch.float32, device=dev),
                  "args": N.SacStepArgs()}
            self._flat["one_call"] = ws
        return ws

    def _learn_batch_one_call(self, batch: TransitionBatch) -> Dict[str, Any]:
        actor, c1, c2 = self._nets(len(batch))
        dev = actor.device
        state = self._f32(batch.state, dev)
        nstate = self._f32(batch.next_state, dev)
        B, S = state.shape
        A = actor.dims[-1] // 2
        act = self._f32(batch.action, dev).reshape(B, A)
        reward = self._f32(batch.reward, dev).reshape(B)
        term = batch.terminated.to(dev).reshape(B)
        term = (term.view(torch.uint8) if term.dtype == torch.bool else term.to(torch.uint8)).contiguous()
        if self.noise_source is None:
            ring = self._flat.get("noise_ring")
            if ring is not None and ring["next"] < ring["buf"].shape[0] and \
                    ring["buf"].shape[2:] == (B, A) and ring["buf"].device == dev:
                noise = ring["buf"][ring["next"]]     # drawn for the whole learn() call at once
                ring["next"] += 1
            else:
                noise = torch.randn(2, B, A, device=dev, dtype=torch.float32)
            noise_a, noise_c = noise[0], noise[1]
        else:   # parity: the reference draws the actor update's noise first, then the target's
            noise_a, noise_c = self._noise(B, A, dev), self._noise(B, A, dev)
        ws = self._one_call_ws(dev, B, S, A)
        logp = torch.empty(B, dtype=torch.float32, device=dev)
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        a = self._step_args(ws, actor, c1, c2, state, act, reward, term, nstate, losses, logp)
        a.noise_actor, a.noise_critic = noise_a.data_ptr(), noise_c.data_ptr()
        N.check(N.lib().pa_sac_step(C.byref(a), N.stream_ptr(dev)))
        for m in (actor, c1, c2):
            m.stepped_natively()
        self._action_batch_log_prob_cache = logp
        report = {"actor_loss": losses[0], "critic_loss": losses[1]}
        if self._entropy_autotune:
            self._entropy_optimizer.state[self._log_entropy]["step"].fill_(
                float(self._alpha_state(dev)["step"]))
            report["entropy_coef"] = losses[2]
        return report

    _NOISE_CHUNK = 1 << 26       # floats of reparameterisation noise drawn at once (256 MB)

    def _learn_native_loop(self, replay_buffer: Any, batch_size: int) -> Optional[Dict[str, List[Any]]]:
        """learn() as pa_sac_learn calls: every round's gather + step sequenced in C (the
        interpreter's ~80 us per round is more than the device needs for the step).  The rounds
        are the ones the per-round loop would run — same index lists, same kernels; the noise is
        drawn for many rounds at once, as `_begin_learn_loop` already does."""
        cls, base = type(self), ContinuousSoftActorCritic
        noise_rounds = getattr(self.noise_source, "rounds", None)
        if (self.noise_source is not None and noise_rounds is None) or not self._one_call_ok() \
                or cls._learn_batch_device is not base._learn_batch_device \
                or cls._learn_batch_one_call is not base._learn_batch_one_call \
                or cls.learn_batch is not ActorCriticBase.learn_batch:
            return None
        actor, c1, c2 = self._nets(batch_size)
        dev = actor.device
        B, S, A = int(batch_size), actor.dims[0], actor.dims[-1] // 2
        plan = self._arena_loop_plan(replay_buffer, B, dev, S, A)
        if plan is None:
            return None
        rounds, w = plan["rounds"], plan["ws"]
        ws = self._one_call_ws(dev, B, S, A)
        losses = self._loop_losses(rounds, 3)
        logp = torch.empty(B, dtype=torch.float32, device=dev)
        lp = N.AcLoopArgs()
        lp.batch = plan["out"]
        lp.gather_rounds = w["G"]
        lp.losses_stride, lp.noise_stride = 3, 2 * B * A
        chunk = max(1, self._NOISE_CHUNK // (2 * B * A))
        done = 0
        while done < rounds:
            n = min(chunk, rounds - done)
            if noise_rounds is not None:
                noise = noise_rounds(done, n, B, A, dev).to(dev, torch.float32).contiguous()
                assert tuple(noise.shape) == (n, 2, B, A)
            else:
                noise = torch.randn(n, 2, B, A, device=dev, dtype=torch.float32)
            a = self._step_args(ws, actor, c1, c2, w["state"][:B], w["action"][:B], w["reward"][:B],
                                w["term"][:B], w["next"][:B], losses, logp)
            lp.rounds = n
            lp.idx_lists = plan["lists"][done].data_ptr()
            lp.noise, lp.losses = noise.data_ptr(), losses[done].data_ptr()
            N.check(N.lib().pa_sac_learn(C.byref(a), plan["arena"].handle, C.byref(lp),
                                         N.stream_ptr(dev)))
            for m in (actor, c1, c2):
                m.stepped_natively(n)
            if self._entropy_autotune:
                self._alpha_state(dev)["step"] += n - 1      # (_step_args counted the first)
            done += n
        self._training_steps += rounds
        replay_buffer._presampled = (plan["lists"], rounds, len(replay_buffer))   # all consumed
        replay_buffer._last_idx = plan["lists"][rounds - 1]
        self._action_batch_log_prob_cache = logp
        torch.cuda.current_stream(dev).synchronize()           # the single host sync of this call
        N.check(N.lib().pa_ac_check(actor.handle))             # a split launch's hand-off expired?
        got = [losses[:, k].tolist() for k in range(3)]     # per key: one list of floats
        report: Dict[str, List[Any]] = {"actor_loss": got[0], "critic_loss": got[1]}
        if self._entropy_autotune:
            self._entropy_optimizer.state[self._log_entropy]["step"].fill_(
                float(self._alpha_state(dev)["step"]))
            report["entropy_coef"] = got[2]
        return report

    # performance report (policy_learner.perf_reported): the fused row launches of pa_sac_step
    def _perf_begin(self) -> None:
        super()._perf_begin()
        N.check(N.lib().pa_sac_timing(1))

    def _perf_end(self) -> Dict[str, float]:
        out = super()._perf_end()
        a, b, n = C.c_double(), C.c_double(), C.c_int64()
        N.check(N.lib().pa_sac_timing_read(C.byref(a), C.byref(b), C.byref(n)))
        if n.value:
            out["sac_rows_a"], out["sac_rows_b"] = a.value, b.value
        N.check(N.lib().pa_sac_timing(0))
        return out

    def _begin_learn_loop(self, rounds: int, batch_size: int) -> None:
        """The reparameterisation noise of every round of this learn() call in ONE generator launch
        (2 draws of (B, A) per round) instead of one launch per round on the step's critical path."""
        self._flat.pop("noise_ring", None)
        if self.noise_source is None and self._one_call_ok() and rounds > 1 and self._flat.get("actor"):
            actor = self._flat["actor"]
            if actor.handle is not None:
                A = actor.dims[-1] // 2
                n = rounds * 2 * batch_size * A
                if n <= (1 << 26):      # 256 MB of fp32 at most
                    self._flat["noise_ring"] = {"next": 0, "buf": torch.randn(
                        rounds, 2, batch_size, A, device=actor.device, dtype=torch.float32)}

    def _end_learn_loop(self) -> None:
        self._flat.pop("noise_ring", None)

    def _learn_batch_device(self, batch: TransitionBatch) -> Dict[str, Any]:
        if self._one_call_ok():
            return self._learn_batch_one_call(batch)
        report = super()._learn_batch_device(batch)
        if self._entropy_autotune:
            actor, _, _ = self._nets(validate=False)
            dev = actor.device
            al = self._alpha_state(dev)
            g = self._entropy_optimizer.param_groups[0]
            al["step"] += 1
            loss = torch.empty(1, dtype=torch.float32, device=dev)
            logp = self._action_batch_log_prob_cache
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                # data parallel: the entropy loss depends on the batch only through mean(log_prob);
                # every rank steps alpha with the GLOBAL mean so the coefficient stays replicated
                logp = logp.mean().reshape(1)
                dist.all_reduce(logp, op=dist.ReduceOp.SUM)
                logp = (logp / dist.get_world_size()).contiguous()
            N.check(N.lib().pa_sac_alpha_step(
                self._log_entropy.data.data_ptr(), al["exp_avg"].data_ptr(),
                al["exp_avg_sq"].data_ptr(), al["max_exp_avg_sq"].data_ptr(),
                al["alpha"].data_ptr(), logp.data_ptr(), int(logp.numel()),
                self._target_entropy_value(), g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                g["weight_decay"], int(bool(g["amsgrad"])), al["step"], loss.data_ptr(),
                N.stream_ptr(dev)))
            self._entropy_optimizer.state[self._log_entropy]["step"].fill_(float(al["step"]))
            report = {**report, "entropy_coef": loss[0]}
        return report

    def act(self, subjective_state: Tensor, available_action_space: Any, exploit: bool = False) -> Any:
        with torch.no_grad():
            exploit_action = self._actor.sample_action(subjective_state)
        if exploit:
            return exploit_action
        return self.exploration_module.act(exploit_action=exploit_action,
                                           action_space=available_action_space,
                                           subjective_state=subjective_state, values=None)

    def compare(self, other: PolicyLearner) -> str:
        diffs = [super().compare(other)]
        if not isinstance(other, ContinuousSoftActorCritic):
            diffs.append("other is not an instance of ContinuousSoftActorCritic")
        else:
            if self._entropy_autotune != other._entropy_autotune:
                diffs.append(f"_entropy_autotune is different: {self._entropy_autotune} vs "
                             f"{other._entropy_autotune}")
            if not torch.allclose(self._entropy_coef.cpu(), other._entropy_coef.cpu()):
                diffs.append(f"_entropy_coef is different: {self._entropy_coef} vs "
                             f"{other._entropy_coef}")
        return "\n".join(d for d in diffs if d)
