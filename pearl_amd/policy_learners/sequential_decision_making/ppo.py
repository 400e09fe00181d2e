"""``ProximalPolicyOptimization`` + ``PPOReplayBuffer`` on HIP.

Mirror of pearl/policy_learners/sequential_decision_making/ppo.py:47-329.

* ``PPOReplayBuffer``: the arena-backed rollout buffer; the three extra per-transition columns
  of ``PPOTransition`` (``gae``, ``lam_return``, ``action_probs``, ppo.py:47-52) live beside the
  arena as device vectors in logical order and are gathered with the batch indices
  (``pa_gather_rows``).
* ``preprocess_replay_buffer`` (ppo.py:201-293): ONE whole-rollout critic forward, ONE actor
  forward + softmax, and the GAE / lambda-return recurrence as ``pa_ppo_gae`` (parallel across
  episodes, the reference's sequential fp32 arithmetic inside one) — instead of a Python loop
  that costs the reference 85 us per transition.
* ``learn_batch``: clipped-surrogate actor loss (+ the detached entropy bonus) with its gradient
  (``pa_ppo_actor_loss``), MSE critic loss vs the lambda return (``pa_mse_head``), forward /
  backward / AdamW(amsgrad) through ``pa_mlp_*``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor, nn

from ... import _comm
from ... import _native as N
from ...action_representation_modules import ActionRepresentationModule
from ...neural_networks.common.value_networks import VanillaValueNetwork
from ...neural_networks.sequential_decision_making.actor_networks import VanillaActorNetwork
from ...replay_buffers.basic_replay_buffer import TensorBasedReplayBuffer
from ...replay_buffers.replay_buffer import ReplayBuffer
from ...replay_buffers.transition import TransitionBatch
from ..exploration import ExplorationModule, PropensityExploration
from ..policy_learner import PolicyLearner, perf_reported
from .actor_critic_base import ActorCriticBase
from .flat_mlp import FlatMlp, layers_of


class PPOTransitionBatch(TransitionBatch):
    """TransitionBatch + gae / lam_return / action_probs (ppo.py:55-82)."""

    _fields = TransitionBatch._fields + ("gae", "lam_return", "action_probs")

    def __init__(self, *args: Any, gae: Optional[Tensor] = None, lam_return: Optional[Tensor] = None,
                 action_probs: Optional[Tensor] = None, **kwargs: Any) -> None:
        self.gae, self.lam_return, self.action_probs = gae, lam_return, action_probs
        super().__init__(*args, **kwargs)

    @classmethod
    def from_parent(cls, parent: TransitionBatch, gae: Optional[Tensor] = None,
                    lam_return: Optional[Tensor] = None, action_probs: Optional[Tensor] = None
                    ) -> "PPOTransitionBatch":
        fields = {k: getattr(parent, k) for k in TransitionBatch._fields}
        return cls(**fields, gae=gae, lam_return=lam_return, action_probs=action_probs)


class PPOReplayBuffer(TensorBasedReplayBuffer):
    """make_replay_buffer_class_for_specific_transition_types(PPOTransition, PPOTransitionBatch)
    (ppo.py:78-82, utils/replay_buffer_utils.py:37-128) on the HBM arena."""

    def __init__(self, capacity: int, sampler: str = "device", staging_rows: int = 0) -> None:
        super().__init__(capacity, sampler=sampler, staging_rows=staging_rows)
        self.extra: Dict[str, Tensor] = {}   # logical order, length len(self); set by preprocess

    def clear(self) -> None:
        super().clear()
        self.extra = {}

    def set_extra(self, gae: Tensor, lam_return: Tensor, action_probs: Tensor) -> None:
        assert gae.numel() == lam_return.numel() == action_probs.numel() == len(self)
        # one (3, n) matrix: the three columns are gathered by ONE launch per minibatch
        self._planes = torch.stack([gae.reshape(-1).float(), lam_return.reshape(-1).float(),
                                    action_probs.reshape(-1).float()]).contiguous()
        self.extra = {"gae": self._planes[0], "lam_return": self._planes[1],
                      "action_probs": self._planes[2]}

    def sample(self, batch_size: int) -> PPOTransitionBatch:
        batch = super().sample(batch_size)
        assert self.extra, "PPOReplayBuffer.sample before preprocess_replay_buffer"
        src = self._planes
        pg, k = self._pregathered, self._pg_slot
        if k is not None and pg is not None:
            # the rounds that share a gather share the gather of the three columns as well
            dst = pg["extras"].get("planes")
            if dst is None or pg["extras"].get("planes_src") is not src:
                idx = pg["idx_flat"]
                dst = torch.empty(3, idx.numel(), dtype=torch.float32, device=src.device)
                N.check(N.lib().pa_gather_planes(src.data_ptr(), src.stride(0), 3, idx.data_ptr(),
                                                 int(idx.numel()), dst.data_ptr(),
                                                 N.stream_ptr(src.device)))
                pg["extras"]["planes"], pg["extras"]["planes_src"] = dst, src
            dst = dst[:, k * batch_size:(k + 1) * batch_size]
        else:
            idx = self.last_indices
            dst = torch.empty(3, batch_size, dtype=torch.float32, device=src.device)
            N.check(N.lib().pa_gather_planes(src.data_ptr(), src.stride(0), 3, idx.data_ptr(),
                                             int(batch_size), dst.data_ptr(), N.stream_ptr(src.device)))
        return PPOTransitionBatch.from_parent(batch, gae=dst[0], lam_return=dst[1],
                                              action_probs=dst[2])

    def rollout(self) -> TransitionBatch:
        """The whole buffer in logical order (index 0 = oldest) as one batch."""
        n = len(self)
        arena = self.arena
        assert arena is not None and n > 0
        idx = torch.arange(n, dtype=torch.int64, device=arena.device)
        return self._gather_batch(idx)

    def rollout_inputs(self):
        """What preprocess_replay_buffer reads of the rollout (ppo.py:211-293), in logical order:
        state, action, reward, terminated, truncated of every transition and the next state of the
        NEWEST one — one gather that leaves out the other 65 535 next states and both action tables
        (`rollout()` copies 3x the bytes), plus a one-row gather."""
        n = len(self)
        arena, z = self.arena, self._layout
        assert arena is not None and z is not None and n > 0
        dev = arena.device
        idx = torch.arange(n, dtype=torch.int64, device=dev)
        state = torch.empty(n, z.state_dim, dtype=torch.float32, device=dev)
        action = torch.empty((n,) + z.action_shape, dtype=z.action_dtype, device=dev)
        reward = torch.empty(n, dtype=z.reward_dtype, device=dev)
        term = torch.empty(n, dtype=torch.bool, device=dev)
        trunc = torch.empty(n, dtype=torch.bool, device=dev)
        out = N.BatchOut()
        out.state, out.action, out.reward = state.data_ptr(), action.data_ptr(), reward.data_ptr()
        out.terminated, out.truncated = term.data_ptr(), trunc.data_ptr()
        arena.gather_device(idx, out)
        last_next = torch.empty(1, z.state_dim, dtype=torch.float32, device=dev)
        one = N.BatchOut()
        one.next_state = last_next.data_ptr()
        arena.gather_device(idx[n - 1:], one)
        shape = (lambda m: (m,) + z.state_shape if len(z.state_shape) else (m, 1))
        return state.view(shape(n)), action, reward, term, trunc, last_next.view(shape(1))


class ProximalPolicyOptimization(ActorCriticBase):
    def __init__(self, action_space: Any, state_dim: Optional[int] = None,
                 actor_hidden_dims: Optional[List[int]] = None,
                 critic_hidden_dims: Optional[List[int]] = None,
                 actor_learning_rate: float = 1e-4, critic_learning_rate: float = 1e-4,
                 history_summarization_learning_rate: float = 1e-4,
                 exploration_module: Optional[ExplorationModule] = None,
                 actor_network_type: type = VanillaActorNetwork,
                 critic_network_type: type = VanillaValueNetwork, discount_factor: float = 0.99,
                 training_rounds: int = 100, batch_size: int = 128, epsilon: float = 0.0,
                 trace_decay_param: float = 0.95, entropy_bonus_scaling: float = 0.01,
                 action_representation_module: Optional[ActionRepresentationModule] = None,
                 actor_network_instance: Optional[nn.Module] = None,
                 critic_network_instance: Optional[nn.Module] = None, **kwargs: Any) -> None:
        if actor_network_type is not VanillaActorNetwork or critic_network_type is not VanillaValueNetwork:
            raise NotImplementedError("pearl_amd PPO: only VanillaActorNetwork / VanillaValueNetwork "
                                      "have HIP kernels")
        super().__init__(
            state_dim=state_dim, action_space=action_space, actor_hidden_dims=actor_hidden_dims,
            use_critic=True, critic_hidden_dims=critic_hidden_dims,
            actor_learning_rate=actor_learning_rate, critic_learning_rate=critic_learning_rate,
            history_summarization_learning_rate=history_summarization_learning_rate,
            actor_network_type=actor_network_type, critic_network_type=critic_network_type,
            use_actor_target=False, use_critic_target=False, actor_soft_update_tau=0.0,
            critic_soft_update_tau=0.0, use_twin_critic=False,
            exploration_module=(exploration_module if exploration_module is not None
                                else PropensityExploration()),
            discount_factor=discount_factor, training_rounds=training_rounds, batch_size=batch_size,
            is_action_continuous=False, on_policy=True,
            action_representation_module=action_representation_module,
            actor_network_instance=actor_network_instance,
            critic_network_instance=critic_network_instance, **kwargs)
        self._epsilon = epsilon
        self._trace_decay_param = trace_decay_param
        self._entropy_bonus_scaling = entropy_bonus_scaling

    # ppo.py:152-192 reads state, action, gae, lam_return and action_probs only
    _fields_unused_by_learn_batch = ("curr_available_actions", "next_available_actions", "next_action")

    # ------------------------------------------------------------------ flat views
    def _nets(self, batch_hint: int = 0, validate: bool = True):
        if not self._flat:
            self._flat["actor"] = FlatMlp(layers_of(self._actor.linear_layers()),
                                          self._actor_optimizer, max(self._batch_size, 1))
            self._flat["critic"] = FlatMlp(layers_of(self._critic.linear_layers()),
                                           self._critic_optimizer, max(self._batch_size, 1))
        if validate:
            return self._flat["actor"].ensure(batch_hint), self._flat["critic"].ensure(batch_hint)
        return self._flat["actor"].ready(batch_hint), self._flat["critic"].ready(batch_hint)

    @staticmethod
    def _f32(t: Tensor, dev: torch.device) -> Tensor:
        return t.to(device=dev, dtype=torch.float32).contiguous()

    # ------------------------------------------------------------------ losses (ppo.py:152-192)
    def _actor_update(self, batch: TransitionBatch) -> Tensor:
        assert isinstance(batch, PPOTransitionBatch) and batch.action_probs is not None
        actor, _ = self._nets(len(batch))
        dev = actor.device
        state = self._f32(batch.state, dev)
        B = state.shape[0]
        arep = self._f32(batch.action, dev).reshape(B, -1)
        A = actor.dims[-1]
        assert arep.shape[1] == A, "PPO needs the action representation the actor outputs"
        logits = actor.forward(state, keep=True)
        d_logits = torch.empty_like(logits)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        N.check(N.lib().pa_ppo_actor_loss(
            logits.data_ptr(), logits.stride(0), arep.data_ptr(), arep.stride(0),
            self._f32(batch.action_probs, dev).data_ptr(), self._f32(batch.gae, dev).data_ptr(),
            B, A, float(self._epsilon), float(self._entropy_bonus_scaling), d_logits.data_ptr(),
            d_logits.stride(0), loss.data_ptr(), N.stream_ptr(dev)))
        actor.backward(state, d_logits, want_dw=True, defer=True)
        actor.adam(reduce="sum")   # the surrogate is a SUM over the (global) minibatch
        return loss[0]

    def _critic_update(self, batch: TransitionBatch) -> Tensor:
        assert isinstance(batch, PPOTransitionBatch) and batch.lam_return is not None
        _, critic = self._nets(len(batch), validate=False)   # validated in _actor_update
        dev = critic.device
        state = self._f32(batch.state, dev)
        B = state.shape[0]
        v = critic.forward(state, keep=True)                      # (B, 1)
        dv = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        N.check(N.lib().pa_mse_head(v.data_ptr(), v.stride(0),
                                    self._f32(batch.lam_return, dev).data_ptr(), B, 2.0 / B, 1.0, 0,
                                    dv.data_ptr(), loss.data_ptr(), N.stream_ptr(dev)))
        critic.backward(state, dv, want_dw=True, defer=True)
        critic.adam()
        return loss[0]

    def _learn_batch_device(self, batch: TransitionBatch) -> Dict[str, Any]:
        """Actor and critic steps of one minibatch (actor_critic_base.py:309-366 order of effects:
        the two networks share nothing but the batch, so their passes run in lock-step — every
        layer of both in one launch — instead of one after the other)."""
        assert isinstance(batch, PPOTransitionBatch) and batch.action_probs is not None \
            and batch.lam_return is not None
        actor, critic = self._nets(len(batch))
        if actor.dims[0] != critic.dims[0] or len(actor.dims) != len(critic.dims):
            return super()._learn_batch_device(batch)
        dev = actor.device
        s = N.stream_ptr(dev)
        state = self._f32(batch.state, dev)
        B = state.shape[0]
        arep = self._f32(batch.action, dev).reshape(B, -1)
        A = actor.dims[-1]
        assert arep.shape[1] == A, "PPO needs the action representation the actor outputs"
        # Data parallel (BASELINE config 4; not in the reference): every rank steps on
        # its own minibatch of its rollout shard.  The surrogate is a SUM over the global minibatch
        # (ppo.py:176-183), the critic loss a MEAN (critic_utils.py:139-167): the critic's head is
        # scaled by 1 / world here, so that ONE SUM all-reduce of both gradient buffers is the
        # gradient of the reference learner on the concatenated minibatch.
        world = _comm.world_size() if getattr(self, "data_parallel", True) else 1
        dp = world > 1 or (os.environ.get("PEARL_AMD_FORCE_DP") == "1" and dist.is_available()
                           and dist.is_initialized())
        if FlatMlp.rowstep_supported(actor, critic, A):
            # forward, both heads and both backward passes in ONE launch (mlp_rowstep.hpp); the
            # critic's head is scaled by 1 / world there
            losses = FlatMlp.ppo_rowstep(
                actor, critic, state, arep, self._f32(batch.action_probs, dev),
                self._f32(batch.gae, dev), float(self._epsilon), float(self._entropy_bonus_scaling),
                self._f32(batch.lam_return, dev), 2.0 / (B * (world if dp else 1)))
            return self._ppo_optimizer_step(actor, critic, dp, losses)
        logits, v = FlatMlp.forward_pair(actor, critic, state, keep=True)
        d_logits = torch.empty_like(logits)
        dv = torch.empty(B, dtype=torch.float32, device=dev)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        # both heads in one launch: the value head's single workgroup runs beside the actor head
        N.check(N.lib().pa_ppo_heads(
            logits.data_ptr(), logits.stride(0), arep.data_ptr(), arep.stride(0),
            self._f32(batch.action_probs, dev).data_ptr(), self._f32(batch.gae, dev).data_ptr(),
            B, A, float(self._epsilon), float(self._entropy_bonus_scaling), d_logits.data_ptr(),
            d_logits.stride(0), v.data_ptr(), v.stride(0),
            self._f32(batch.lam_return, dev).data_ptr(), dv.data_ptr(), losses.data_ptr(), s))
        if dp and world > 1:
            dv.mul_(1.0 / world)      # (2 / B) (v - R) -> (2 / (B world)) (v - R), exact for world = 2^k
        FlatMlp.backward_pair(actor, critic, state, d_logits, dv, want_dw=True, defer=True)
        return self._ppo_optimizer_step(actor, critic, dp, losses)

    def _ppo_optimizer_step(self, actor: FlatMlp, critic: FlatMlp, dp: bool,
                            losses: torch.Tensor) -> Dict[str, Any]:
        """Weight gradients + AdamW of both networks (the backward passes left them pending)."""
        if dp:
            FlatMlp.adam_pair_data_parallel(actor, critic, force=True)
        elif os.environ.get("PEARL_AMD_PPO_PAIR", "1") == "1" and not (
                dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            # both networks' weight gradients + AdamW in one launch (same AdamW configuration only)
            FlatMlp.adam_pair(actor, critic, None)
        else:
            actor.adam(reduce="sum")   # the surrogate is a SUM over the (global) minibatch
            critic.adam()
        return {"actor_loss": losses[0], "critic_loss": losses[1]}

    # ------------------------------------------------------------------ learn (ppo.py:194-293)
    @perf_reported
    def learn(self, replay_buffer: ReplayBuffer) -> Dict[str, Any]:
        self.preprocess_replay_buffer(replay_buffer)
        return ActorCriticBase.learn(self, replay_buffer)

    def _learn_native_loop(self, replay_buffer: ReplayBuffer, batch_size: int
                           ) -> Optional[Dict[str, List[Any]]]:
        """The training rounds of learn() as ONE pa_ppo_learn call: the rounds the per-round loop
        would run — same index lists, same fused row step and weight-gradient / AdamW launches —
        with the batches of up to 16 rounds gathered by one launch that writes
        state || one-hot(action) rows directly (the per-round loop spent 112 us of interpreter
        time per 107 us round, plus a one-hot launch and a gather of fields learn_batch never
        reads).  None: this call takes the per-round loop (data parallel, another action
        representation / history summarisation / preprocess_batch, a replay buffer subclass)."""
        from ...action_representation_modules import OneHotActionTensorRepresentationModule
        from ..policy_learner import IdentityHistorySummarizationModule
        if os.environ.get("PEARL_AMD_AC_LOOP", "1") == "0":
            return None
        cls, base = type(self), ProximalPolicyOptimization
        rb = replay_buffer
        if cls._learn_batch_device is not base._learn_batch_device \
                or cls._ppo_optimizer_step is not base._ppo_optimizer_step \
                or cls.preprocess_batch is not ActorCriticBase.preprocess_batch \
                or cls._preprocess_for_learn is not ActorCriticBase._preprocess_for_learn \
                or cls.learn_batch is not ActorCriticBase.learn_batch \
                or type(rb) is not PPOReplayBuffer or rb.arena is None or not rb.extra:
            return None
        if (dist.is_available() and dist.is_initialized()) or os.environ.get("PEARL_AMD_FORCE_DP") == "1":
            return None
        arm = self.action_representation_module
        if type(arm) is not OneHotActionTensorRepresentationModule \
                or type(self._history_summarization_module) is not IdentityHistorySummarizationModule \
                or hasattr(getattr(self, "safety_module", None), "lambda_constraint"):
            return None
        actor, critic = self._nets(batch_size)
        dev = actor.device
        B, S, A = int(batch_size), actor.dims[0], actor.dims[-1]
        rounds = int(self._training_rounds)
        pre, z, arena = rb._presampled, rb._layout, rb.arena
        if pre is None or pre[1] != 0 or tuple(pre[0].shape) != (rounds, B) or rounds <= 0 \
                or arena.device != dev or rb._device_for_batches != dev \
                or len(z.state_shape) > 1 or z.state_dim != S or z.action_elems != 1 \
                or z.action_dtype.is_floating_point or arm.max_number_actions != A \
                or critic.dims[0] != S or len(actor.dims) != len(critic.dims) \
                or not FlatMlp.rowstep_supported(actor, critic, A) \
                or os.environ.get("PEARL_AMD_PPO_PAIR", "1") != "1":
            return None
        planes = rb._planes
        if planes.device != dev or planes.shape[1] != len(rb):
            return None
        G = max(1, min(rounds, self._LOOP_GATHER_BYTES // ((4 * (S + A) + 12) * B), len(rb) // B))
        ws = self._flat.get("loop_ws")
        key = (dev, B, S, A, G)
        if ws is None or ws["key"] != key:
            f32 = dict(dtype=torch.float32, device=dev)
            ws = {"key": key, "x": torch.empty(G * B, S + A, **f32),
                  "planes": torch.empty(3, G * B, **f32), "d_logits": torch.empty(B, A, **f32),
                  "dv": torch.empty(B, **f32), "args": N.PpoLearnArgs()}
            self._flat["loop_ws"] = ws
        losses = self._loop_losses(rounds, 2)
        a = ws["args"]
        a.actor, a.critic = actor.handle.value, critic.handle.value
        a.B, a.S, a.A, a.rounds, a.gather_rounds = B, S, A, rounds, G
        a.idx_lists = pre[0].data_ptr()
        a.planes, a.plane_stride = planes.data_ptr(), planes.stride(0)
        a.x, a.planes_ws = ws["x"].data_ptr(), ws["planes"].data_ptr()
        a.epsilon, a.entropy_scale = float(self._epsilon), float(self._entropy_bonus_scaling)
        a.value_grad_scale = 2.0 / B
        a.d_logits, a.d_value = ws["d_logits"].data_ptr(), ws["dv"].data_ptr()
        a.losses, a.losses_stride = losses.data_ptr(), 2
        a.actor_step, a.critic_step = actor.next_adam_step(), critic.next_adam_step()
        N.check(N.lib().pa_ppo_learn(C.byref(a), arena.handle, N.stream_ptr(dev)))
        for m in (actor, critic):
            m.stepped_natively(rounds)
        self._training_steps += rounds
        rb._presampled = (pre[0], rounds, len(rb))          # all consumed
        rb._last_idx = pre[0][rounds - 1]
        torch.cuda.current_stream(dev).synchronize()           # the single host sync of this call
        return {"actor_loss": losses[:, 0].tolist(), "critic_loss": losses[:, 1].tolist()}

    def preprocess_replay_buffer(self, replay_buffer: ReplayBuffer) -> None:
        assert isinstance(replay_buffer, PPOReplayBuffer), \
            "pearl_amd PPO needs a pearl_amd PPOReplayBuffer"
        n = len(replay_buffer)
        assert n > 0
        if type(replay_buffer).rollout is PPOReplayBuffer.rollout and replay_buffer._layout.has_next_state:
            r_state, r_action, r_reward, r_term, r_trunc, r_last_next = replay_buffer.rollout_inputs()
        else:       # (a subclass with its own notion of the rollout)
            roll = replay_buffer.rollout()
            r_state, r_action, r_reward, r_term, r_trunc = (roll.state, roll.action, roll.reward,
                                                           roll.terminated, roll.truncated)
            r_last_next = roll.next_state[n - 1:n]
        actor, critic = self._nets(n)
        dev = actor.device
        state = self._f32(self._history_summarization_module(r_state), dev)
        arep = self._f32(self.action_representation_module(r_action), dev).reshape(n, -1)
        # One pair launch over the whole rollout.  (It used to matter that this forward is learn_batch's
        # to the bit — with epsilon = 0 the clipped surrogate passes a gradient only where the ratio is
        # exactly 1.  That holds while learn_batch takes the fp32 forward; minibatches that fill the
        # chip (ceil(B / 16) * 2 > 256 tiles) take the fused row step's bf16x3 forward, whose logits
        # differ from these in the last bits — as the reference's own minibatch forward differs from
        # its rollout forward when MKL blocks the two shapes differently.  epsilon > 0 does not care.)
        if actor.dims[0] == critic.dims[0] and len(actor.dims) == len(critic.dims):
            logits, values = FlatMlp.forward_pair(actor, critic, state)
            values = values.reshape(n).contiguous()
        else:
            values = critic.forward(state).reshape(n).contiguous()
            logits = actor.forward(state)
        aprob = torch.empty(n, dtype=torch.float32, device=dev)
        s = N.stream_ptr(dev)
        N.check(N.lib().pa_softmax_action_prob(logits.data_ptr(), logits.stride(0), arep.data_ptr(),
                                               arep.stride(0), n, actor.dims[-1], None,
                                               aprob.data_ptr(), s))
        # value of the newest transition's next state bootstraps the recurrence (ppo.py:255-269)
        last_next = self._f32(self._history_summarization_module(r_last_next), dev)
        next_value = critic.forward(last_next).reshape(1)
        gae = torch.empty(n, dtype=torch.float32, device=dev)
        lam_return = torch.empty(n, dtype=torch.float32, device=dev)
        reward = self._f32(r_reward, dev).reshape(n)

        def as_u8(flags: Tensor) -> Tensor:      # a bool tensor is its bytes (no conversion launch)
            f = flags.to(dev).reshape(n).contiguous()
            return f.view(torch.uint8) if f.dtype == torch.bool else f.to(torch.uint8)
        term, trunc = as_u8(r_term), as_u8(r_trunc)
        N.check(N.lib().pa_ppo_gae(reward.data_ptr(), term.data_ptr(), trunc.data_ptr(),
                                   values.data_ptr(), next_value.data_ptr(),
                                   float(self._discount_factor), float(self._trace_decay_param), n,
                                   gae.data_ptr(), lam_return.data_ptr(), s))
        replay_buffer.set_extra(gae, lam_return, aprob)

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
        if not isinstance(other, ProximalPolicyOptimization):
            diffs.append("other is not an instance of ProximalPolicyOptimization")
        else:
            for attr in ("_epsilon", "_trace_decay_param", "_entropy_bonus_scaling"):
                if getattr(self, attr) != getattr(other, attr):
                    diffs.append(f"{attr} is different: {getattr(self, attr)} vs "
                                 f"{getattr(other, attr)}")
        return "\n".join(d for d in diffs if d)
