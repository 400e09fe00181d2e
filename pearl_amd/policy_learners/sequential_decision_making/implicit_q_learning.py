"""ImplicitQLearning (reference: pearl/policy_learners/sequential_decision_making/
implicit_q_learning.py:52-351).

Same constructor and ``learn_batch`` effects as the reference: a state-value network trained by
expectile regression towards ONE randomly chosen target critic (:186-196, :271-285), twin critics
regressed to ``r + gamma V(s')`` (:248-269), policy extraction by advantage-weighted regression
(:197-246), all three losses formed on the PRE-step parameters, one backward, then the value /
actor / critic AdamW(amsgrad) steps and the critic-target soft update (:159-184).

The two ``torch.randint(0, 2, (1,))`` draws that pick the target critic (value loss first, then
actor loss) are made here exactly as in the reference (host generator, no device work), so a seeded
run picks the same critics.  Everything else is libpearl_amd on flat parameter views: expectile /
advantage head ``pa_iql_value_head``, AWR heads ``pa_awr_head``, the deterministic actor's tanh
scaling ``pa_tanh_action`` / ``pa_tanh_action_grad``, paired twin-critic launches.  Actor types
with HIP heads: ``VanillaContinuousActorNetwork`` (weighted MSE on the action) and
``VanillaActorNetwork`` (weighted log-likelihood of the dataset action) and
``GaussianActorNetwork`` (weighted ``get_log_probability`` of the dataset action, :231-236,
``pa_gauss_awr_head``).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn, optim

from ... import _native as N
from ...action_representation_modules import ActionRepresentationModule
from ...neural_networks.common.value_networks import VanillaValueNetwork
from ...neural_networks.sequential_decision_making.actor_networks import (
    GaussianActorNetwork, VanillaActorNetwork, VanillaContinuousActorNetwork)
from ...neural_networks.sequential_decision_making.q_value_networks import VanillaQValueNetwork
from ...replay_buffers.transition import TransitionBatch
from ..exploration import ExplorationModule, NoExploration
from ..policy_learner import PolicyLearner
from .actor_critic_base import ActorCriticBase
from .flat_mlp import FlatMlp, layers_of


class ImplicitQLearning(ActorCriticBase):
    def __init__(self, action_space: Any, state_dim: Optional[int] = None,
                 actor_hidden_dims: Optional[List[int]] = None,
                 critic_hidden_dims: Optional[List[int]] = None,
                 value_critic_hidden_dims: Optional[List[int]] = None,
                 exploration_module: Optional[ExplorationModule] = None,
                 actor_network_type: type = VanillaActorNetwork,
                 critic_network_type: type = VanillaQValueNetwork,
                 value_network_type: type = VanillaValueNetwork,
                 value_critic_learning_rate: float = 1e-3, actor_learning_rate: float = 1e-3,
                 critic_learning_rate: float = 1e-3,
                 history_summarization_learning_rate: float = 1e-3,
                 critic_soft_update_tau: float = 0.05, discount_factor: float = 0.99,
                 training_rounds: int = 5, batch_size: int = 128, expectile: float = 0.5,
                 temperature_advantage_weighted_regression: float = 0.5,
                 advantage_clamp: float = 100.0,
                 action_representation_module: Optional[ActionRepresentationModule] = None,
                 actor_network_instance: Optional[nn.Module] = None,
                 critic_network_instance: Optional[nn.Module] = None,
                 value_network_instance: Optional[nn.Module] = None, **kwargs: Any) -> None:
        if actor_network_type not in (VanillaActorNetwork, VanillaContinuousActorNetwork,
                                      GaussianActorNetwork):
            raise NotImplementedError("pearl_amd ImplicitQLearning: HIP policy-extraction heads exist "
                                      "for VanillaActorNetwork, VanillaContinuousActorNetwork and "
                                      "GaussianActorNetwork")
        if critic_network_type is not VanillaQValueNetwork or value_network_type is not VanillaValueNetwork:
            raise NotImplementedError("pearl_amd ImplicitQLearning: only VanillaQValueNetwork twin "
                                      "critics and a VanillaValueNetwork have HIP kernels")
        continuous = bool(getattr(action_space, "is_continuous", hasattr(action_space, "low")))
        super().__init__(
            state_dim=state_dim, action_space=action_space, actor_hidden_dims=actor_hidden_dims,
            critic_hidden_dims=critic_hidden_dims, actor_learning_rate=actor_learning_rate,
            critic_learning_rate=critic_learning_rate,
            history_summarization_learning_rate=history_summarization_learning_rate,
            actor_network_type=actor_network_type, critic_network_type=critic_network_type,
            use_actor_target=False, use_critic_target=True,
            critic_soft_update_tau=critic_soft_update_tau, use_twin_critic=True,
            exploration_module=(exploration_module if exploration_module is not None
                                else NoExploration()),
            discount_factor=discount_factor, training_rounds=training_rounds, batch_size=batch_size,
            is_action_continuous=continuous, on_policy=False,
            action_representation_module=action_representation_module,
            actor_network_instance=actor_network_instance,
            critic_network_instance=critic_network_instance, **kwargs)
        self._expectile = expectile
        self._temperature_advantage_weighted_regression = temperature_advantage_weighted_regression
        self._advantage_clamp = advantage_clamp
        if value_network_instance is not None:
            self._value_network: nn.Module = value_network_instance
        else:
            assert state_dim is not None and value_critic_hidden_dims is not None
            self._value_network = value_network_type(input_dim=state_dim,
                                                     hidden_dims=value_critic_hidden_dims,
                                                     output_dim=1)
        self._value_network_optimizer: optim.Optimizer = optim.AdamW(
            self._value_network.parameters(), lr=value_critic_learning_rate, amsgrad=True)

    # ------------------------------------------------------------------ flat views
    def _nets(self, batch_hint: int = 0, validate: bool = True):
        """(actor, value network, critic 1, critic 2)."""
        if not self._flat:
            mb = max(self._batch_size, 1)
            if isinstance(self._actor, GaussianActorNetwork):
                # fc_mu and fc_std are ONE last layer of 2A rows (as in ContinuousSoftActorCritic)
                a = self._actor
                head = ([a.fc_mu.weight, a.fc_std.weight], [a.fc_mu.bias, a.fc_std.bias])
                actor_layers = layers_of(a.trunk_layers()) + [head]
            else:
                actor_layers = layers_of(self._actor.linear_layers())
            self._flat["actor"] = FlatMlp(actor_layers, self._actor_optimizer, mb)
            self._flat["value"] = FlatMlp(layers_of(self._value_network.linear_layers()),
                                          self._value_network_optimizer, mb)
            for i, (c, ct) in enumerate(((self._critic._critic_1, self._critic_target._critic_1),
                                         (self._critic._critic_2, self._critic_target._critic_2)), 1):
                self._flat[f"critic{i}"] = FlatMlp(layers_of(c.linear_layers()),
                                                   self._critic_optimizer, mb,
                                                   target_layers=layers_of(ct.linear_layers()))
        nets = (self._flat["actor"], self._flat["value"], self._flat["critic1"], self._flat["critic2"])
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

    # ------------------------------------------------------------------ one-call step
    def _actor_kind(self) -> int:
        """pa_iql_step's actor_kind: 0 tanh-squashed deterministic, 1 Gaussian, 2 softmax."""
        if isinstance(self._actor, VanillaContinuousActorNetwork):
            return 0
        return 1 if isinstance(self._actor, GaussianActorNetwork) else 2

    def _one_call_ok(self, c1: FlatMlp, c2: FlatMlp) -> bool:
        """pa_iql_step sequences the whole learn_batch in C: the single-process step on critics the
        fused row step takes (PEARL_AMD_IQL_ONE_CALL=0: the per-stage path)."""
        if os.environ.get("PEARL_AMD_IQL_ONE_CALL", "1") == "0":
            return False
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return False
        # pa_iql_step takes ONE critic step count (c1's) for both twins: with counters that differ
        # (a partially restored optimizer state) c2 would get c1's bias correction — the per-stage
        # path steps each critic with its own count
        if getattr(c1, "_steps", 0) != getattr(c2, "_steps", 0):
            return False
        memo = self._flat.get("one_call_ok")
        key = (c1.handle.value, c2.handle.value)
        if memo is None or memo[0] != key:
            memo = (key, FlatMlp.rowstep_supported(c1, c2))
            self._flat["one_call_ok"] = memo
        return memo[1]

    def _step_args(self, nets, B: int, S: int, A: int, losses: Tensor) -> "N.IqlStepArgs":
        """pa_iql_step_args of the next step (batch pointers and the two picks left to the caller)."""
        actor, value, c1, c2 = nets
        dev = actor.device
        HW = actor.dims[-1]
        ws = self._flat.get("one_call")
        if ws is None or ws["key"] != (dev, B, S, A, HW):
            n = int(N.lib().pa_iql_scratch_floats(B, S, A, HW))
            ws = {"key": (dev, B, S, A, HW), "scratch": torch.empty(n, dtype=torch.float32, device=dev),
                  "zeros": torch.zeros(B + 1, dtype=torch.float32, device=dev),
                  "args": N.IqlStepArgs()}
            self._flat["one_call"] = ws
        a = ws["args"]
        a.actor, a.value = actor.handle.value, value.handle.value
        a.critic1, a.critic2 = c1.handle.value, c2.handle.value
        a.B, a.S, a.A, a.actor_kind = B, S, A, self._actor_kind()
        if a.actor_kind != 2:
            low, high = self._bounds(dev)
            a.low, a.high = low.data_ptr(), high.data_ptr()
        else:
            a.low = a.high = None
        a.expectile = float(self._expectile)
        a.temperature = float(self._temperature_advantage_weighted_regression)
        a.adv_clamp, a.gamma = float(self._advantage_clamp), float(self._discount_factor)
        a.tau = float(self._critic_soft_update_tau)
        a.actor_step, a.value_step = actor.next_adam_step(), value.next_adam_step()
        a.critic_step = c1.next_adam_step()
        a.zeros, a.scratch, a.losses = ws["zeros"].data_ptr(), ws["scratch"].data_ptr(), losses.data_ptr()
        return a

    def _learn_batch_one_call(self, batch: TransitionBatch, nets) -> Dict[str, Any]:
        actor = nets[0]
        dev = actor.device
        state = self._f32(batch.state, dev)
        nstate = self._f32(batch.next_state, dev)
        B, S = state.shape
        act = self._f32(batch.action, dev).reshape(B, -1)
        A = act.shape[1]
        # the reference's own two host draws, value loss first (:189-190, :205-206)
        pick_value = int(torch.randint(0, 2, (1,)).item())
        pick_actor = int(torch.randint(0, 2, (1,)).item())
        reward = self._f32(batch.reward, dev).reshape(B)
        term = batch.terminated.to(dev).reshape(B).to(torch.uint8).contiguous()
        losses = torch.empty(3, dtype=torch.float32, device=dev)    # value | critic | actor
        a = self._step_args(nets, B, S, A, losses)
        a.state, a.ld_state = state.data_ptr(), state.stride(0)
        a.next_state, a.ld_next_state = nstate.data_ptr(), nstate.stride(0)
        a.action, a.ld_action = act.data_ptr(), act.stride(0)
        a.xq, a.ld_xq = None, 0
        a.reward, a.terminated = reward.data_ptr(), term.data_ptr()
        a.pick_value, a.pick_actor = pick_value, pick_actor
        N.check(N.lib().pa_iql_step(C.byref(a), N.stream_ptr(dev)))
        for m in nets:
            m.stepped_natively()
        return {"value_loss": losses[0], "actor_loss": losses[2], "critic_loss": losses[1]}

    def _learn_native_loop(self, replay_buffer: Any, batch_size: int) -> Optional[Dict[str, List[Any]]]:
        """learn() as ONE pa_iql_learn call: every round's gather + step sequenced in C (the
        per-round loop spends 234 us of interpreter time per ~165 us of device work).  The rounds
        the per-round loop would run — same index lists, same kernels, the same host draws in the
        same order.  None: this call takes the per-round loop."""
        from ...action_representation_modules import OneHotActionTensorRepresentationModule
        from ...replay_buffers.basic_replay_buffer import TensorBasedReplayBuffer
        from ..policy_learner import IdentityHistorySummarizationModule
        if os.environ.get("PEARL_AMD_AC_LOOP", "1") == "0":
            return None
        cls, base = type(self), ImplicitQLearning
        rb = replay_buffer
        if cls._learn_batch_device is not base._learn_batch_device \
                or cls._learn_batch_one_call is not base._learn_batch_one_call \
                or cls.preprocess_batch is not ActorCriticBase.preprocess_batch \
                or cls._preprocess_for_learn is not ActorCriticBase._preprocess_for_learn \
                or cls.learn_batch is not ActorCriticBase.learn_batch \
                or not isinstance(rb, TensorBasedReplayBuffer) or rb.arena is None \
                or type(rb).sample is not TensorBasedReplayBuffer.sample \
                or type(rb)._gather_batch is not TensorBasedReplayBuffer._gather_batch \
                or type(self._history_summarization_module) is not IdentityHistorySummarizationModule \
                or hasattr(getattr(self, "safety_module", None), "lambda_constraint"):
            return None
        nets = self._nets(batch_size)
        actor, value, c1, c2 = nets
        if not self._one_call_ok(c1, c2):
            return None
        dev = actor.device
        B, S = int(batch_size), actor.dims[0]
        A = c1.dims[0] - S                     # width of the action representation
        arm = self.action_representation_module
        rounds = int(self._training_rounds)
        pre, z, arena = rb._presampled, rb._layout, rb.arena
        if pre is None or pre[1] != 0 or tuple(pre[0].shape) != (rounds, B) or rounds <= 0 \
                or arena.device != dev or rb._device_for_batches != dev or A <= 0 \
                or len(z.state_shape) > 1 or z.state_dim != S or not z.has_next_state or z.has_cost:
            return None
        if type(arm) is OneHotActionTensorRepresentationModule:
            if z.action_elems != 1 or z.action_dtype.is_floating_point or arm.max_number_actions != A:
                return None
            onehot = 1
        elif type(arm).__name__ == "IdentityActionRepresentationModule":
            if z.action_elems != A or z.action_dtype != torch.float32:
                return None
            onehot = 0
        else:
            return None
        G = max(1, min(rounds, self._LOOP_GATHER_BYTES // ((4 * (3 * S + A + 1) + 1) * B), len(rb) // B))
        ws = self._flat.get("loop_ws")
        key = (dev, B, S, A, G)
        if ws is None or ws["key"] != key:
            n = G * B

            def new(shape, dtype=torch.float32):
                return torch.empty(shape, dtype=dtype, device=dev)
            ws = {"key": key, "state": new((n, S)), "next": new((n, S)), "x": new((n, S + A)),
                  "reward": new((n,)), "term": new((n,), torch.uint8)}
            self._flat["loop_ws"] = ws
        lp = N.AcLoopArgs()
        o = lp.batch
        o.state, o.next_state, o.x = ws["state"].data_ptr(), ws["next"].data_ptr(), ws["x"].data_ptr()
        o.reward_f32, o.terminated = ws["reward"].data_ptr(), ws["term"].data_ptr()
        o.rep_dim, o.rep_onehot = A, onehot
        losses = self._loop_losses(rounds, 3)
        a = self._step_args(nets, B, S, A, losses)
        a.state, a.ld_state, a.next_state, a.ld_next_state = o.state, S, o.next_state, S
        a.xq, a.ld_xq = o.x, S + A
        a.action, a.ld_action = o.x + 4 * S, S + A
        a.reward, a.terminated = o.reward_f32, o.terminated
        # the two draws of every round, in the per-round loop's order
        picks = (C.c_int32 * (2 * rounds))()
        for r in range(2 * rounds):
            picks[r] = int(torch.randint(0, 2, (1,)).item())
        lp.rounds, lp.gather_rounds = rounds, G
        lp.idx_lists = pre[0].data_ptr()
        lp.losses, lp.losses_stride = losses.data_ptr(), 3
        N.check(N.lib().pa_iql_learn(C.byref(a), arena.handle, C.byref(lp), picks, N.stream_ptr(dev)))
        for m in nets:
            m.stepped_natively(rounds)
        self._training_steps += rounds
        rb._presampled = (pre[0], rounds, len(rb))              # all consumed
        rb._last_idx = pre[0][rounds - 1]
        torch.cuda.current_stream(dev).synchronize()           # the single host sync of this call
        got = [losses[:, k].tolist() for k in range(3)]
        return {"value_loss": got[0], "actor_loss": got[2], "critic_loss": got[1]}

    # ------------------------------------------------------------------ learn_batch (:159-184)
    def _learn_batch_device(self, batch: TransitionBatch) -> Dict[str, Any]:
        nets = self._nets(len(batch))
        actor, value, c1, c2 = nets
        if self._one_call_ok(c1, c2):
            return self._learn_batch_one_call(batch, nets)
        dev = actor.device
        lib, s = N.lib(), N.stream_ptr(dev)
        state = self._f32(batch.state, dev)
        nstate = self._f32(batch.next_state, dev)
        B, S = state.shape
        act = self._f32(batch.action, dev).reshape(B, -1)
        A = act.shape[1]
        # which target critic each loss regresses to / weighs with (:189-190, :205-206): the
        # reference's own two host draws, value loss first
        pick_value = int(torch.randint(0, 2, (1,)).item())
        pick_actor = int(torch.randint(0, 2, (1,)).item())
        xq = torch.empty(B, S + A, dtype=torch.float32, device=dev)
        N.check(lib.pa_concat_cols(state.data_ptr(), state.stride(0), act.data_ptr(), act.stride(0),
                                   xq.data_ptr(), B, S, A, s))
        tq = [q.reshape(B) for q in FlatMlp.forward_pair(c1, c2, xq, use_target=True)]
        # V(s') first: the engine keeps ONE forward's activations per network for the backward pass
        vn = value.forward(nstate).reshape(B)
        # ---- value loss (:186-196) and the advantage weights (:203-215)
        v = value.forward(state, keep=True)
        dv = torch.empty(B, dtype=torch.float32, device=dev)
        adv = torch.empty(B, dtype=torch.float32, device=dev)
        losses = torch.empty(3, dtype=torch.float32, device=dev)    # value | critic | actor
        N.check(lib.pa_iql_value_head(tq[pick_value].data_ptr(), tq[pick_actor].data_ptr(),
                                      v.data_ptr(), v.stride(0), float(self._expectile),
                                      float(self._temperature_advantage_weighted_regression),
                                      float(self._advantage_clamp), B, dv.data_ptr(),
                                      adv.data_ptr(), losses.data_ptr(), s))
        # ---- critic loss (:248-269): r + gamma V(s') with the pre-step value network
        y = torch.empty(B, dtype=torch.float32, device=dev)
        reward = self._f32(batch.reward, dev).reshape(B)
        term = batch.terminated.to(dev).reshape(B).to(torch.uint8).contiguous()
        zero = self._flat.get("zeros")
        if zero is None or zero[0].numel() < B or zero[0].device != dev:
            zero = (torch.zeros(max(B, 1), dtype=torch.float32, device=dev),
                    torch.zeros(1, dtype=torch.float32, device=dev))
            self._flat["zeros"] = zero
        # pa_sac_twin(mode 1) with q1 = q2 = V(s'), alpha = 0: y = V(s') gamma (1 - term) + r
        N.check(lib.pa_sac_twin(1, vn.data_ptr(), vn.data_ptr(), zero[0].data_ptr(),
                                zero[1].data_ptr(), reward.data_ptr(), term.data_ptr(),
                                float(self._discount_factor), B, y.data_ptr(), None, None, s))
        # the critics' forward, MSE heads and backward: one launch when the pair qualifies
        # (mlp_rowstep.hpp; nothing steps before the end of this method, so the order is free)
        critics_fused = FlatMlp.rowstep_supported(c1, c2)
        if critics_fused:
            FlatMlp.mse_rowstep_pair(c1, c2, xq, y, 1.0 / B, 0.5, loss_out=losses[1:2])
        else:
            qs = [q.reshape(B) for q in FlatMlp.forward_pair(c1, c2, xq, keep=True)]
            dqs = [torch.empty_like(q) for q in qs]
            for i in range(2):
                N.check(lib.pa_mse_head(qs[i].data_ptr(), 1, y.data_ptr(), B, 1.0 / B, 0.5, int(i > 0),
                                        dqs[i].data_ptr(), losses[1:].data_ptr(), s))
        # ---- actor loss (:197-246)
        head = actor.forward(state, keep=True)
        d_head = torch.empty_like(head)
        if isinstance(self._actor, VanillaContinuousActorNetwork):
            low, high = self._bounds(dev)
            pred = torch.empty(B, A, dtype=torch.float32, device=dev)
            N.check(lib.pa_tanh_action(head.data_ptr(), head.stride(0), None, 0, low.data_ptr(),
                                       high.data_ptr(), 0.0, B, A, pred.data_ptr(), pred.stride(0), s))
            d_pred = torch.empty_like(pred)
            N.check(lib.pa_awr_head(0, pred.data_ptr(), pred.stride(0), act.data_ptr(), act.stride(0),
                                    adv.data_ptr(), B, A, d_pred.data_ptr(), d_pred.stride(0),
                                    losses[2:].data_ptr(), s))
            N.check(lib.pa_tanh_action_grad(head.data_ptr(), head.stride(0), low.data_ptr(),
                                            high.data_ptr(), d_pred.data_ptr(), d_pred.stride(0), B,
                                            A, d_head.data_ptr(), d_head.stride(0), s))
        elif isinstance(self._actor, GaussianActorNetwork):
            # stochastic continuous actor: -mean(adv * log pi(a | s)) (:231-236, :260-261)
            low, high = self._bounds(dev)
            assert head.shape[1] == 2 * A
            logp = torch.empty(B, dtype=torch.float32, device=dev)
            N.check(lib.pa_gauss_awr_head(head.data_ptr(), head.stride(0), act.data_ptr(),
                                          act.stride(0), low.data_ptr(), high.data_ptr(),
                                          adv.data_ptr(), B, A, d_head.data_ptr(), d_head.stride(0),
                                          logp.data_ptr(), losses[2:].data_ptr(), s))
        else:
            assert head.shape[1] == A, "the softmax actor outputs one logit per action slot"
            N.check(lib.pa_awr_head(1, head.data_ptr(), head.stride(0), act.data_ptr(), act.stride(0),
                                    adv.data_ptr(), B, A, d_head.data_ptr(), d_head.stride(0),
                                    losses[2:].data_ptr(), s))
        # ---- one backward, then the steps in the reference's order (:171-176), target update
        value.backward(state, dv, want_dw=True, defer=True)
        actor.backward(state, d_head, want_dw=True, defer=True)
        if not critics_fused:
            FlatMlp.backward_pair(c1, c2, xq, dqs[0], dqs[1], want_dw=True, defer=True)
        value.adam()
        actor.adam()
        if not FlatMlp.adam_pair(c1, c2, self._critic_soft_update_tau):
            c1.soft_update(self._critic_soft_update_tau)
            c2.soft_update(self._critic_soft_update_tau)
        return {"value_loss": losses[0], "actor_loss": losses[2], "critic_loss": losses[1]}

    def _actor_update(self, batch: TransitionBatch) -> Tensor:      # pragma: no cover
        raise NotImplementedError("ImplicitQLearning forms its three losses in one learn_batch")

    def _critic_update(self, batch: TransitionBatch) -> Tensor:     # pragma: no cover
        raise NotImplementedError("ImplicitQLearning forms its three losses in one learn_batch")

    # ------------------------------------------------------------------ act (act-time torch)
    def act(self, subjective_state: Tensor, available_action_space: Any, exploit: bool = False) -> Any:
        with torch.no_grad():
            if isinstance(self._actor, (VanillaContinuousActorNetwork, GaussianActorNetwork)):
                exploit_action = self._actor.sample_action(subjective_state)
                probs = None
            else:
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

    def get_extra_state(self) -> Dict[str, Any]:
        state = super().get_extra_state()
        state["value_optimizer"] = self._value_network_optimizer.state_dict()
        return state

    def set_extra_state(self, state: Dict[str, Any]) -> None:
        super().set_extra_state(state)
        if "value_optimizer" in state:
            self._value_network_optimizer.load_state_dict(state["value_optimizer"])

    def compare(self, other: PolicyLearner) -> str:
        diffs = [super().compare(other)]
        if not isinstance(other, ImplicitQLearning):
            diffs.append("other is not an instance of ImplicitQLearning")
        else:
            for attr in ("_expectile", "_temperature_advantage_weighted_regression",
                         "_advantage_clamp"):
                if getattr(self, attr) != getattr(other, attr):
                    diffs.append(f"{attr} is different: {getattr(self, attr)} vs "
                                 f"{getattr(other, attr)}")
            mine, theirs = self._value_network.state_dict(), other._value_network.state_dict()
            if mine.keys() != theirs.keys() or any(
                    not torch.allclose(mine[k].cpu().float(), theirs[k].cpu().float(), rtol=1e-5,
                                       atol=1e-8) for k in mine):
                diffs.append("_value_network is different")
        return "\n".join(d for d in diffs if d)
