"""DeepDeterministicPolicyGradient
(reference: pearl/policy_learners/sequential_decision_making/ddpg.py:41-190).

Same constructor, attributes and ``learn_batch`` effects as the reference: a deterministic tanh
actor with a target copy, twin Q critics with target copies, actor objective ``-mean(Q1(s, pi(s)))``
(:106-121), clipped double-Q Bellman targets from the TARGET actor and TARGET critics (:123-156),
``(mse(q1, y) + mse(q2, y)) / 2`` (critic_utils.py:170-203), soft target updates every step.

All arithmetic runs in libpearl_amd on flat parameter views (``FlatMlp``): the actor's tanh +
action scaling is ``pa_tanh_action`` (written straight into the critic's ``[state | action]``
input), its gradient ``pa_tanh_action_grad`` on the critic's input gradient
(``pa_mlp_backward(want_dw=0)``), the twin critics run as paired launches.  No CPU path.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Callable, Dict, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn

from ... import _native as N
from ...action_representation_modules import ActionRepresentationModule
from ...neural_networks.sequential_decision_making.actor_networks import (
    VanillaContinuousActorNetwork)
from ...neural_networks.sequential_decision_making.q_value_networks import VanillaQValueNetwork
from ...replay_buffers.transition import TransitionBatch
from ..exploration import ExplorationModule, NormalDistributionExploration
from ..policy_learner import PolicyLearner
from .actor_critic_base import ActorCriticBase
from .flat_mlp import FlatMlp, layers_of


class DeepDeterministicPolicyGradient(ActorCriticBase):
    def __init__(self, action_space: Any, state_dim: Optional[int] = None,
                 actor_hidden_dims: Optional[List[int]] = None,
                 critic_hidden_dims: Optional[List[int]] = None,
                 exploration_module: Optional[ExplorationModule] = None,
                 actor_learning_rate: float = 1e-3, critic_learning_rate: float = 1e-3,
                 history_summarization_learning_rate: float = 1e-3,
                 actor_network_type: type = VanillaContinuousActorNetwork,
                 critic_network_type: type = VanillaQValueNetwork,
                 actor_soft_update_tau: float = 0.005, critic_soft_update_tau: float = 0.005,
                 discount_factor: float = 0.99, training_rounds: int = 1, batch_size: int = 256,
                 action_representation_module: Optional[ActionRepresentationModule] = None,
                 actor_network_instance: Optional[nn.Module] = None,
                 critic_network_instance: Optional[nn.Module] = None, **kwargs: Any) -> None:
        if (actor_network_type is not VanillaContinuousActorNetwork
                or critic_network_type is not VanillaQValueNetwork):
            raise NotImplementedError("pearl_amd DDPG / TD3: only VanillaContinuousActorNetwork + "
                                      "VanillaQValueNetwork twin critics have HIP kernels")
        super().__init__(
            state_dim=state_dim, action_space=action_space, actor_hidden_dims=actor_hidden_dims,
            critic_hidden_dims=critic_hidden_dims, actor_learning_rate=actor_learning_rate,
            critic_learning_rate=critic_learning_rate,
            history_summarization_learning_rate=history_summarization_learning_rate,
            actor_network_type=actor_network_type, critic_network_type=critic_network_type,
            use_actor_target=True, use_critic_target=True,
            actor_soft_update_tau=actor_soft_update_tau,
            critic_soft_update_tau=critic_soft_update_tau, use_twin_critic=True,
            exploration_module=(exploration_module if exploration_module is not None
                                else NormalDistributionExploration(mean=0.0, std_dev=0.1)),
            discount_factor=discount_factor, training_rounds=training_rounds, batch_size=batch_size,
            is_action_continuous=True, on_policy=False,
            action_representation_module=action_representation_module,
            actor_network_instance=actor_network_instance,
            critic_network_instance=critic_network_instance, **kwargs)

    # ------------------------------------------------------------------ flat views
    def _nets(self, batch_hint: int = 0, validate: bool = True):
        """(actor, critic 1, critic 2) as flat networks, each with its target copy."""
        if not self._flat:
            mb = max(self._batch_size, 1)
            self._flat["actor"] = FlatMlp(
                layers_of(self._actor.linear_layers()), self._actor_optimizer, mb,
                target_layers=layers_of(self._actor_target.linear_layers()))
            for i, (c, ct) in enumerate(((self._critic._critic_1, self._critic_target._critic_1),
                                         (self._critic._critic_2, self._critic_target._critic_2)), 1):
                self._flat[f"critic{i}"] = FlatMlp(layers_of(c.linear_layers()),
                                                   self._critic_optimizer, mb,
                                                   target_layers=layers_of(ct.linear_layers()))
        nets = (self._flat["actor"], self._flat["critic1"], self._flat["critic2"])
        if validate:
            return tuple(m.ensure(batch_hint) for m in nets)
        return tuple(m.ready(batch_hint) for m in nets)

    def _bounds(self, dev: torch.device):
        sp = self._actor._action_space
        hit = self._flat.get("bounds")
        if hit is None or hit[0] is not sp or hit[1] != dev:
            hit = (sp, dev, sp.low.to(dev, torch.float32).contiguous(),
                   sp.high.to(dev, torch.float32).contiguous())
            self._flat["bounds"] = hit
        return hit[2], hit[3]

    @staticmethod
    def _f32(t: Tensor, dev: torch.device) -> Tensor:
        return t.to(device=dev, dtype=torch.float32).contiguous()

    def _policy_input(self, actor: FlatMlp, state: Tensor, use_target: bool, keep: bool,
                      noise: Optional[Tensor] = None, clip: float = 0.0):
        """[state | pi(state)] for the critics; returns (x, pre-tanh head)."""
        dev = state.device
        B, S = state.shape
        A = actor.dims[-1]
        head = actor.forward(state, use_target=use_target, keep=keep)
        x = torch.empty(B, S + A, dtype=torch.float32, device=dev)
        x[:, :S].copy_(state)
        low, high = self._bounds(dev)
        act = x[:, S:]
        N.check(N.lib().pa_tanh_action(head.data_ptr(), head.stride(0), N.ptr(noise),
                                       noise.stride(0) if noise is not None else 0,
                                       low.data_ptr(), high.data_ptr(), float(clip), B, A,
                                       act.data_ptr(), x.stride(0), N.stream_ptr(dev)))
        return x, head

    # ------------------------------------------------------------------ losses
    def _actor_update(self, batch: TransitionBatch) -> Tensor:
        """-mean(Q1(s, pi(s))) (ddpg.py:106-121), backward through critic 1's input, AdamW."""
        actor, c1, _ = self._nets(len(batch))
        dev = actor.device
        state = self._f32(batch.state, dev)
        B, S = state.shape
        A = actor.dims[-1]
        s = N.stream_ptr(dev)
        xa, head = self._policy_input(actor, state, use_target=False, keep=True)
        q1 = c1.forward(xa, keep=True)
        dq = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        N.check(N.lib().pa_neg_mean_head(q1.data_ptr(), q1.stride(0), B, dq.data_ptr(),
                                         loss.data_ptr(), s))
        # the reference also forms (then discards) critic 1's parameter gradients here
        # (actor_critic_base.py:342-348); only its input gradient matters
        dx = c1.backward(xa, dq, want_dw=False, want_dx=True)
        d_head = torch.empty_like(head)
        low, high = self._bounds(dev)
        da = dx[:, S:]
        N.check(N.lib().pa_tanh_action_grad(head.data_ptr(), head.stride(0), low.data_ptr(),
                                            high.data_ptr(), da.data_ptr(), dx.stride(0), B, A,
                                            d_head.data_ptr(), d_head.stride(0), s))
        actor.backward(state, d_head, want_dw=True, defer=True)
        actor.adam()
        return loss[0]

    # ------------------------------------------------------------------ one-call step
    def _one_call_ok(self) -> bool:
        """pa_ddpg_step sequences the whole learn_batch in C (the per-stage path below needed ~185 us
        of host time per step for ~160 us of kernels).  Single process, and only for the stages as
        this class defines them: a data-parallel step all-reduces between backward and AdamW, and a
        subclass that overrides a stage keeps the per-stage path."""
        if os.environ.get("PEARL_AMD_DDPG_ONE_CALL", "1") == "0":
            return False
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False
        cls, base = type(self), DeepDeterministicPolicyGradient
        return (cls._actor_update is base._actor_update and cls._critic_update is base._critic_update
                and cls._update_critic_target is base._update_critic_target
                and cls._update_actor_target is base._update_actor_target
                and cls._policy_input is base._policy_input
                and self._use_critic and self._use_critic_target and self._use_actor_target)

    def _one_call_ws(self, dev: torch.device, B: int, S: int, A: int) -> Dict[str, Any]:
        ws = self._flat.get("one_call")
        if ws is None or ws["key"] != (dev, B, S, A):
            n = int(N.lib().pa_ddpg_scratch_floats(B, S, A))
            ws = {"key": (dev, B, S, A), "scratch": torch.empty(n, dtype=torch.float32, device=dev),
                  "zeros": torch.zeros(max(B, 1), dtype=torch.float32, device=dev),
                  "args": N.DdpgStepArgs()}
            self._flat["one_call"] = ws
        return ws

    def _step_args(self, ws: Dict[str, Any], actor: FlatMlp, c1: FlatMlp, c2: FlatMlp,
                   state: Tensor, act: Tensor, reward: Tensor, term: Tensor, nstate: Tensor,
                   losses: Tensor, clip: float) -> "N.DdpgStepArgs":
        """pa_ddpg_step_args of the next step on these batch tensors (noise pointer and the
        do_actor / do_targets switches left to the caller)."""
        dev = actor.device
        B, S = state.shape
        A = actor.dims[-1]
        low, high = self._bounds(dev)
        a = ws["args"]
        a.actor, a.critic1, a.critic2 = actor.handle, c1.handle, c2.handle
        a.state, a.ld_state = state.data_ptr(), state.stride(0)
        a.action, a.ld_action = act.data_ptr(), act.stride(0)
        a.reward, a.terminated = reward.data_ptr(), term.data_ptr()
        a.next_state, a.ld_next_state = nstate.data_ptr(), nstate.stride(0)
        a.noise_clip = float(clip)
        a.low, a.high, a.zeros = low.data_ptr(), high.data_ptr(), ws["zeros"].data_ptr()
        a.B, a.S, a.A = B, S, A
        a.gamma = float(self._discount_factor)
        a.critic_tau, a.actor_tau = float(self._critic_soft_update_tau), float(self._actor_soft_update_tau)
        a.actor_step, a.critic_step = actor._steps + 1, c1._steps + 1
        a.scratch, a.losses = ws["scratch"].data_ptr(), losses.data_ptr()
        return a

    def _learn_one_call(self, batch: TransitionBatch, do_actor: bool, do_targets: bool):
        actor, c1, c2 = self._nets(len(batch))
        dev = actor.device
        state = self._f32(batch.state, dev)
        nstate = self._f32(batch.next_state, dev)
        B, S = state.shape
        A = actor.dims[-1]
        act = self._f32(batch.action, dev).reshape(B, A)
        reward = self._f32(batch.reward, dev).reshape(B)
        term = batch.terminated.to(dev).reshape(B)
        term = (term.view(torch.uint8) if term.dtype == torch.bool else term.to(torch.uint8)).contiguous()
        noise, clip = self._target_noise(B, A, dev)
        ws = self._one_call_ws(dev, B, S, A)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        a = self._step_args(ws, actor, c1, c2, state, act, reward, term, nstate, losses, clip)
        if noise is not None:
            noise = noise.to(dev, torch.float32).contiguous()
            assert noise.shape == (B, A)
        a.target_noise = N.ptr(noise)
        a.do_actor, a.do_targets = int(do_actor), int(do_targets)
        N.check(N.lib().pa_ddpg_step(C.byref(a), N.stream_ptr(dev)))
        if do_actor:
            actor.stepped_natively()
        c1.stepped_natively()
        c2.stepped_natively()
        return (losses[0] if do_actor else None), losses[1]

    # ------------------------------------------------------------------ learn() as one call
    _NOISE_CHUNK = 1 << 26       # floats of target-smoothing noise drawn at once (256 MB)
    _actor_update_freq = 1       # (TD3 sets its own)

    def _loop_noise(self, rounds: int, B: int, A: int, dev: torch.device):
        """(noise [rounds, B, A] or None, clip) for `rounds` steps of the native loop; False when
        this learner's noise cannot be drawn ahead (a parity `noise_source`, an overridden
        `_target_noise`)."""
        if type(self)._target_noise is not DeepDeterministicPolicyGradient._target_noise:
            return False
        return None, 0.0

    def _native_loop_is_mine(self) -> bool:
        cls, base = type(self), DeepDeterministicPolicyGradient
        return (cls._learn_batch_device is base._learn_batch_device
                and cls._learn_one_call is base._learn_one_call
                and cls.learn_batch is ActorCriticBase.learn_batch)

    def _learn_native_loop(self, replay_buffer: Any, batch_size: int) -> Optional[Dict[str, List[Any]]]:
        """learn() as pa_ddpg_learn calls: every round's gather + step sequenced in C — the rounds
        the per-round loop would run (same index lists, same kernels, TD3's delayed actor by the
        same rule); the smoothing noise is drawn for many rounds at once."""
        if not self._one_call_ok() or not self._native_loop_is_mine():
            return None
        actor, c1, c2 = self._nets(batch_size)
        dev = actor.device
        B, S, A = int(batch_size), actor.dims[0], actor.dims[-1]
        if self._loop_noise(0, B, A, dev) is False:
            return None
        plan = self._arena_loop_plan(replay_buffer, B, dev, S, A)
        if plan is None:
            return None
        rounds, w = plan["rounds"], plan["ws"]
        ws = self._one_call_ws(dev, B, S, A)
        losses = self._loop_losses(rounds, 2)
        freq = max(int(self._actor_update_freq), 1)
        lp = N.AcLoopArgs()
        lp.batch = plan["out"]
        lp.gather_rounds = w["G"]
        lp.losses_stride, lp.noise_stride = 2, B * A
        lp.actor_update_freq = freq
        chunk = max(1, self._NOISE_CHUNK // (B * A))
        step0 = int(self._training_steps)
        done = 0
        while done < rounds:
            n = min(chunk, rounds - done)
            noise, clip = self._loop_noise(n, B, A, dev)
            a = self._step_args(ws, actor, c1, c2, w["state"][:B], w["action"][:B], w["reward"][:B],
                                w["term"][:B], w["next"][:B], losses, clip)
            lp.rounds = n
            lp.idx_lists = plan["lists"][done].data_ptr()
            lp.noise, lp.losses = N.ptr(noise), losses[done].data_ptr()
            lp.training_step0 = step0 + done
            N.check(N.lib().pa_ddpg_learn(C.byref(a), plan["arena"].handle, C.byref(lp),
                                          N.stream_ptr(dev)))
            actor.stepped_natively(sum(1 for r in range(n) if (step0 + done + r + 1) % freq == 0))
            c1.stepped_natively(n)
            c2.stepped_natively(n)
            done += n
        self._training_steps += rounds
        replay_buffer._presampled = (plan["lists"], rounds, len(replay_buffer))   # all consumed
        replay_buffer._last_idx = plan["lists"][rounds - 1]
        torch.cuda.current_stream(dev).synchronize()           # the single host sync of this call
        N.check(N.lib().pa_ac_check(actor.handle))             # a split launch's hand-off expired?
        got = [losses[:, k].tolist() for k in range(2)]     # per key: one list of floats
        return self._loop_report(got, step0, freq)

    def _loop_report(self, got: List[List[float]], step0: int, freq: int) -> Dict[str, List[Any]]:
        """got: [actor losses, critic losses] of the rounds."""
        return {"actor_loss": got[0], "critic_loss": got[1]}

    def _learn_batch_device(self, batch: TransitionBatch) -> Dict[str, Any]:
        if self._one_call_ok():
            al, cl = self._learn_one_call(batch, True, True)
            return {"actor_loss": al, "critic_loss": cl}
        return super()._learn_batch_device(batch)

    def _target_noise(self, B: int, A: int, dev: torch.device):
        """(noise, clip) added to the target policy's action; DDPG adds none (ddpg.py:123-131)."""
        return None, 0.0

    def _critic_update(self, batch: TransitionBatch) -> Tensor:
        actor, c1, c2 = self._nets(len(batch), validate=False)
        dev = actor.device
        state = self._f32(batch.state, dev)
        nstate = self._f32(batch.next_state, dev)
        B, S = state.shape
        A = actor.dims[-1]
        s = N.stream_ptr(dev)
        # ---- Bellman target from the target actor and the target critics (ddpg.py:123-147)
        noise, clip = self._target_noise(B, A, dev)
        xn, _ = self._policy_input(actor, nstate, use_target=True, keep=False, noise=noise, clip=clip)
        nq1, nq2 = FlatMlp.forward_pair(c1, c2, xn, use_target=True)
        y = torch.empty(B, dtype=torch.float32, device=dev)
        reward = self._f32(batch.reward, dev).reshape(B)
        term = batch.terminated.to(dev).reshape(B).to(torch.uint8).contiguous()
        zero = self._flat.get("zeros")
        if zero is None or zero[0].numel() < B or zero[0].device != dev:
            zero = (torch.zeros(max(B, 1), dtype=torch.float32, device=dev),
                    torch.zeros(1, dtype=torch.float32, device=dev))
            self._flat["zeros"] = zero
        # pa_sac_twin(mode 1) with alpha = 0, log_prob = 0: y = min(q1', q2') gamma (1 - term) + r
        N.check(N.lib().pa_sac_twin(1, nq1.data_ptr(), nq2.data_ptr(), zero[0].data_ptr(),
                                    zero[1].data_ptr(), reward.data_ptr(), term.data_ptr(),
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

    def _update_actor_target(self) -> None:
        actor, _, _ = self._nets(validate=False)
        actor.soft_update(self._actor_soft_update_tau)

    # ------------------------------------------------------------------ act (act-time torch)
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
        if not isinstance(other, DeepDeterministicPolicyGradient):
            diffs.append("other is not an instance of DeepDeterministicPolicyGradient")
        return "\n".join(d for d in diffs if d)
