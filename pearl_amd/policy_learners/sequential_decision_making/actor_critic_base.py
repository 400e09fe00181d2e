"""``ActorCriticBase`` on the HIP MLP engine.

Mirror of pearl/policy_learners/sequential_decision_making/actor_critic_base.py:54-564: the same
constructor arguments, network construction (xavier-uniform actor with bias 0.01, critic through
``make_critic``, deep-copied targets), two ``optim.AdamW(amsgrad=True)`` optimizers whose state
dicts stay live (``get_extra_state`` / ``set_extra_state``), and the ``learn_batch`` sequencing
(:309-366): actor loss -> actor step, critic loss -> critic step, target soft updates.

What differs is where the arithmetic runs: subclasses implement ``_actor_update`` /
``_critic_update`` with ``pa_mlp_*`` and the fused heads of libpearl_amd on flat views of the
parameters (``FlatMlp``); nothing in the step is a torch op.
"""
from __future__ import annotations

import copy
import os
from abc import abstractmethod
from typing import Any, Dict, List, Optional

import torch
from torch import nn, optim

from ... import _comm
from ... import _native as N
from ...action_representation_modules import ActionRepresentationModule
from ...neural_networks.common.utils import xavier_init_weights
from ...neural_networks.common.value_networks import VanillaValueNetwork
from ...neural_networks.sequential_decision_making.actor_networks import (ActorNetwork,
                                                                         VanillaActorNetwork)
from ...neural_networks.sequential_decision_making.q_value_networks import VanillaQValueNetwork
from ...neural_networks.sequential_decision_making.twin_critic import TwinCritic
from ...replay_buffers.replay_buffer import ReplayBuffer
from ...replay_buffers.transition import TransitionBatch
from ..exploration import ExplorationModule
from ..policy_learner import PolicyLearner, _looks_like_batch, accept_optimizer, perf_reported
from .flat_mlp import FlatMlp


def make_critic(state_dim: int, hidden_dims: Optional[List[int]], use_twin_critic: bool,
                network_type: type, action_dim: Optional[int] = None) -> nn.Module:
    """pearl/utils/functional_utils/learning/critic_utils.py:39-100."""
    if use_twin_critic:
        assert action_dim is not None and hidden_dims is not None
        return TwinCritic(state_dim=state_dim, action_dim=action_dim, hidden_dims=hidden_dims,
                          network_type=network_type, init_fn=xavier_init_weights)
    if network_type is VanillaQValueNetwork:
        return network_type(state_dim=state_dim, action_dim=action_dim, hidden_dims=hidden_dims,
                            output_dim=1)
    if network_type is VanillaValueNetwork:
        return network_type(input_dim=state_dim, hidden_dims=hidden_dims, output_dim=1)
    raise NotImplementedError(f"Type {network_type} cannot be used to instantiate a critic network.")


class ActorCriticBase(PolicyLearner):
    def __init__(self, exploration_module: ExplorationModule, state_dim: Optional[int] = None,
                 actor_hidden_dims: Optional[List[int]] = None, use_critic: bool = True,
                 critic_hidden_dims: Optional[List[int]] = None, action_space: Any = None,
                 actor_learning_rate: float = 1e-3, critic_learning_rate: float = 1e-3,
                 history_summarization_learning_rate: float = 1e-3,
                 actor_network_type: type = VanillaActorNetwork,
                 critic_network_type: type = VanillaQValueNetwork,
                 use_actor_target: bool = False, use_critic_target: bool = False,
                 actor_soft_update_tau: float = 0.005, critic_soft_update_tau: float = 0.005,
                 use_twin_critic: bool = False, discount_factor: float = 0.99,
                 training_rounds: int = 1, batch_size: int = 256,
                 is_action_continuous: bool = False, on_policy: bool = False,
                 action_representation_module: Optional[ActionRepresentationModule] = None,
                 actor_network_instance: Optional[ActorNetwork] = None,
                 critic_network_instance: Optional[nn.Module] = None,
                 actor_optimizer: Optional[optim.Optimizer] = None,
                 critic_optimizer: Optional[optim.Optimizer] = None,
                 history_summarization_optimizer: Optional[optim.Optimizer] = None) -> None:
        super().__init__(on_policy=on_policy, is_action_continuous=is_action_continuous,
                         training_rounds=training_rounds, batch_size=batch_size,
                         exploration_module=exploration_module,
                         action_representation_module=action_representation_module,
                         action_space=action_space)
        self._state_dim = state_dim
        self._action_space = action_space
        self._use_actor_target = use_actor_target
        self._use_critic_target = use_critic_target
        self._use_twin_critic = use_twin_critic
        self._use_critic = use_critic
        rep = self.action_representation_module
        if actor_network_instance is not None:
            self._actor: nn.Module = actor_network_instance
        else:
            assert state_dim is not None and actor_hidden_dims is not None
            self._actor = actor_network_type(
                input_dim=state_dim, hidden_dims=actor_hidden_dims,
                output_dim=(rep.representation_dim if self._is_action_continuous
                            else rep.max_number_actions),
                action_space=action_space)
        self._actor.apply(xavier_init_weights)
        if actor_optimizer is not None:     # (actor_critic_base.py:159-167: used as handed over)
            self._actor_optimizer: optim.Optimizer = accept_optimizer(
                actor_optimizer, self._actor.parameters(), f"{type(self).__name__} (actor)")
        else:
            self._actor_optimizer = optim.AdamW(
                [{"params": self._actor.parameters(), "lr": actor_learning_rate, "amsgrad": True}])
        self._actor_soft_update_tau = actor_soft_update_tau
        self._critic_soft_update_tau = critic_soft_update_tau
        if self._use_actor_target:
            self._actor_target: nn.Module = copy.deepcopy(self._actor)
        if self._use_critic:
            if critic_network_instance is not None:
                self._critic: nn.Module = critic_network_instance
            else:
                assert state_dim is not None and critic_hidden_dims is not None
                self._critic = make_critic(state_dim=state_dim, action_dim=rep.representation_dim,
                                           hidden_dims=critic_hidden_dims,
                                           use_twin_critic=use_twin_critic,
                                           network_type=critic_network_type)
            if critic_optimizer is not None:     # (actor_critic_base.py:200-211)
                self._critic_optimizer: optim.Optimizer = accept_optimizer(
                    critic_optimizer, self._critic.parameters(), f"{type(self).__name__} (critic)")
            else:
                self._critic_optimizer = optim.AdamW(
                    [{"params": self._critic.parameters(), "lr": critic_learning_rate, "amsgrad": True}])
            if self._use_critic_target:
                self._critic_target: nn.Module = copy.deepcopy(self._critic)
        self._discount_factor = discount_factor
        self._history_summarization_optimizer = history_summarization_optimizer
        self._history_summarization_learning_rate = history_summarization_learning_rate
        self._actor_learning_rate: float = self._actor_optimizer.param_groups[0]["lr"]
        if self._use_critic:
            self._critic_learning_rate: float = self._critic_optimizer.param_groups[0]["lr"]
        self._flat: Dict[str, Any] = {}   # FlatMlp handles, built lazily on the device

    # ------------------------------------------------------------------ plumbing
    def set_history_summarization_module(self, value: nn.Module) -> None:
        if any(True for _ in value.parameters()):
            raise NotImplementedError("pearl_amd actor-critic learners: trainable history "
                                      "summarisation modules are not built")
        self._history_summarization_module = value

    def reset(self, action_space: Any) -> None:
        self._action_space = action_space

    def __deepcopy__(self, memo: dict) -> "ActorCriticBase":
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, {} if k == "_flat" else copy.deepcopy(v, memo))
        return new

    # ------------------------------------------------------------------ learn_batch (:309-366)
    def learn_batch(self, batch: TransitionBatch) -> Dict[str, Any]:
        report = self._learn_batch_device(batch)
        out = {k: (v.item() if isinstance(v, torch.Tensor) else v) for k, v in report.items()}
        _comm.check_exchange_after_sync()     # (data parallel: a P2P peer that never answered)
        return out

    def _learn_batch_device(self, batch: TransitionBatch) -> Dict[str, Any]:
        """learn_batch with the losses left on the device (no host synchronisation)."""
        report = {"actor_loss": self._actor_update(batch)}
        if self._use_critic:
            # the critics' soft target update follows their AdamW step with nothing in between:
            # twin critics take both in one launch (_step_twin_critics)
            self._target_update_follows = bool(self._use_critic_target)
            try:
                report["critic_loss"] = self._critic_update(batch)
            finally:
                self._target_update_follows = False
        if self._use_critic_target:
            self._update_critic_target()
        if self._use_actor_target:
            self._update_actor_target()
        return report

    @perf_reported
    def learn(self, replay_buffer: ReplayBuffer) -> Dict[str, Any]:
        """PolicyLearner.learn (policy_learner.py:190-231) with the same report — one list of
        floats per key — but ONE host synchronisation for the whole call instead of one `.item()`
        per loss per round: the host keeps enqueueing round r+1 while the device runs round r."""
        if len(replay_buffer) == 0:
            return {}
        batch_size = self._clamped_batch_size(replay_buffer)
        pending: Dict[str, List[Any]] = {}
        # device sampler: the index lists of every round in one launch (one workgroup each) instead
        # of a single-workgroup kernel on the critical path of every round
        presample = getattr(replay_buffer, "presample", None)
        if presample is not None:
            from ...replay_buffers.basic_replay_buffer import TensorBasedReplayBuffer
            if isinstance(replay_buffer, TensorBasedReplayBuffer) \
                    and os.environ.get("PEARL_AMD_PREGATHER", "1") != "0":
                # ... and one gather launch for the batches of as many rounds as fit the workspace
                presample(self._training_rounds, batch_size, pregather_bytes=self._LOOP_GATHER_BYTES)
            else:
                presample(self._training_rounds, batch_size)
        try:
            for m in self._flat.values():      # validated on the first step, trusted until the end
                if isinstance(m, FlatMlp):
                    m._loop_validated = False
                    # every learn() call rebuilds the library's derived weight copies from the
                    # parameters as they are now (one launch per network per CALL): torch's version
                    # counters — the per-step check — cannot see writes through `.data`
                    m.invalidate()
            native = self._learn_native_loop(replay_buffer, batch_size)
            if native is not None:
                return native
            FlatMlp.in_learn_loop = True
            self._begin_learn_loop(self._training_rounds, batch_size)
            for _ in range(self._training_rounds):
                self._training_steps += 1
                batch = replay_buffer.sample(batch_size)
                if not _looks_like_batch(batch):
                    continue
                for k, v in self._learn_batch_device(self._preprocess_for_learn(batch)).items():
                    pending.setdefault(k, []).append(v)
        finally:
            self._end_learn_loop()
            FlatMlp.leave_learn_loop()
            if presample is not None:
                replay_buffer.drop_presampled()
        report: Dict[str, List[Any]] = {}
        for k, vals in pending.items():
            dev_ix = [i for i, v in enumerate(vals) if isinstance(v, torch.Tensor)]
            if dev_ix:
                got = torch.stack([vals[i].reshape(()) for i in dev_ix]).tolist()
                for i, g in zip(dev_ix, got):
                    vals[i] = g
            report[k] = vals
        _comm.check_exchange_after_sync()     # (after the tolist() above synchronised the call)
        return report

    # batch fields this learner's learn_batch never reads: learn() — which owns the batches it
    # samples — does not spend launches on representing them (PPO: two (B, A, A) one-hot tensors
    # per minibatch, 9 us of a 190 us step).  preprocess_batch itself is unchanged.
    _fields_unused_by_learn_batch: tuple = ()

    def _preprocess_for_learn(self, batch: TransitionBatch) -> TransitionBatch:
        if self._fields_unused_by_learn_batch and \
                type(self).preprocess_batch is ActorCriticBase.preprocess_batch:
            for name in self._fields_unused_by_learn_batch:
                setattr(batch, name, None)
        return self.preprocess_batch(batch)

    def preprocess_batch(self, batch: TransitionBatch) -> TransitionBatch:
        safety = getattr(self, "safety_module", None)
        if safety is not None and hasattr(safety, "lambda_constraint"):
            batch.reward = batch.reward - safety.lambda_constraint * batch.cost
        return super().preprocess_batch(batch)

    def _learn_native_loop(self, replay_buffer: ReplayBuffer, batch_size: int
                           ) -> Optional[Dict[str, List[Any]]]:
        """Hook: the whole learn() call — every round's gather and step — sequenced by the library
        (pa_sac_learn / pa_ddpg_learn); None: this call takes the per-round loop below."""
        return None

    _LOOP_GATHER_BYTES = 64 << 20     # batch workspace of a native learn loop

    def _arena_loop_plan(self, replay_buffer: ReplayBuffer, batch_size: int, dev: torch.device,
                         S: int, A: int) -> Optional[Dict[str, Any]]:
        """What the native loops need from the replay buffer — its arena, the presampled index
        lists of this call, one batch of workspace — or None when a round of this call is not
        exactly `sample -> the library's own preprocess_batch -> learn_batch` on float32 rows."""
        from ...replay_buffers.basic_replay_buffer import TensorBasedReplayBuffer
        from ..policy_learner import IdentityHistorySummarizationModule
        if os.environ.get("PEARL_AMD_AC_LOOP", "1") == "0":
            return None
        rb = replay_buffer
        if not isinstance(rb, TensorBasedReplayBuffer) or rb.arena is None \
                or type(rb).sample is not TensorBasedReplayBuffer.sample \
                or type(rb)._gather_batch is not TensorBasedReplayBuffer._gather_batch:
            return None
        pre = rb._presampled
        rounds = int(self._training_rounds)
        if pre is None or pre[1] != 0 or tuple(pre[0].shape) != (rounds, batch_size) or rounds <= 0:
            return None
        z, arena = rb._layout, rb.arena
        if arena.device != dev or rb._device_for_batches != dev:
            return None
        if not (z.has_next_state and not z.has_cost and len(z.state_shape) <= 1 and z.state_dim == S
                and z.action_dtype == torch.float32 and z.action_elems == A
                and z.reward_dtype == torch.float32 and rb._is_action_continuous):
            return None
        if type(self).preprocess_batch is not ActorCriticBase.preprocess_batch \
                or type(self)._preprocess_for_learn is not ActorCriticBase._preprocess_for_learn \
                or type(self._history_summarization_module) is not IdentityHistorySummarizationModule \
                or type(self.action_representation_module).__name__ != "IdentityActionRepresentationModule":
            return None
        if hasattr(getattr(self, "safety_module", None), "lambda_constraint"):
            return None
        # workspace for the batches of G consecutive rounds: ONE gather launch fills them (5 us a
        # round as a launch of its own at B = 1024, ~0.3 us as a slice of a 64 MB gather)
        row_bytes = 4 * (2 * S + A + 1) + 2
        G = max(1, min(rounds, self._LOOP_GATHER_BYTES // (row_bytes * batch_size),
                       len(rb) // batch_size))
        ws = self._flat.get("loop_ws")
        key = (dev, batch_size, S, A, G)
        if ws is None or ws["key"] != key:
            n = G * batch_size

            def new(shape, dtype=torch.float32):
                return torch.empty(shape, dtype=dtype, device=dev)
            ws = {"key": key, "G": G, "state": new((n, S)), "action": new((n, A)),
                  "reward": new((n,)), "term": new((n,), torch.uint8),
                  "trunc": new((n,), torch.uint8), "next": new((n, S))}
            self._flat["loop_ws"] = ws
        out = N.BatchOut()
        out.state, out.action, out.reward = ws["state"].data_ptr(), ws["action"].data_ptr(), ws["reward"].data_ptr()
        out.terminated, out.truncated = ws["term"].data_ptr(), ws["trunc"].data_ptr()
        out.next_state = ws["next"].data_ptr()
        return {"arena": arena, "lists": pre[0], "rounds": rounds, "ws": ws, "out": out}

    def _loop_losses(self, rounds: int, width: int) -> torch.Tensor:
        """[rounds, width] zeros in pinned, device-mapped host memory: the step kernels store each
        round's losses straight into it (a few bytes per round over PCIe, fire-and-forget), so the
        end of a native loop is one stream synchronisation and no device-to-host copy (the same
        arrangement as DeepQLearning.learn's loss buffer)."""
        buf = self._flat.get("loss_host")
        if buf is None or buf.numel() < rounds * width:
            buf = torch.zeros(max(2 * rounds * width, 4096), dtype=torch.float32, pin_memory=True)
            self._flat["loss_host"] = buf
        out = buf[:rounds * width].view(rounds, width)
        out.zero_()
        return out

    def _begin_learn_loop(self, rounds: int, batch_size: int) -> None:
        """Hook: per-call preparation a learner can amortise over the rounds of one learn()."""

    def _end_learn_loop(self) -> None:
        """Hook: undo _begin_learn_loop."""

    _target_update_follows = False     # set around _critic_update by _learn_batch_device
    _critic_target_done = False        # the soft update already rode the critics' AdamW launch

    def _step_twin_critics(self, c1: FlatMlp, c2: FlatMlp) -> None:
        """AdamW of the twin critics (deferred weight gradients).  When this learn_batch goes on to
        soft-update their targets, that update rides the same launch (FlatMlp.adam_pair) and
        `_update_critic_target` finds it done."""
        tau = self._critic_soft_update_tau if self._target_update_follows else None
        self._critic_target_done = FlatMlp.adam_pair(c1, c2, tau)

    def _twin_target_update(self, c1: FlatMlp, c2: FlatMlp) -> None:
        if self._critic_target_done:
            self._critic_target_done = False
            return
        c1.soft_update(self._critic_soft_update_tau)
        c2.soft_update(self._critic_soft_update_tau)

    @abstractmethod
    def _actor_update(self, batch: TransitionBatch) -> torch.Tensor:
        """Actor loss, backward and AdamW step; returns the loss as a device scalar."""

    @abstractmethod
    def _critic_update(self, batch: TransitionBatch) -> torch.Tensor:
        """Critic loss, backward and AdamW step; returns the loss as a device scalar."""

    def _update_critic_target(self) -> None:
        raise NotImplementedError

    def _update_actor_target(self) -> None:
        raise NotImplementedError

    # ------------------------------------------------------------------ checkpoints (:411-428)
    def get_extra_state(self) -> Dict[str, Any]:
        state = {"actor_optimizer": self._actor_optimizer.state_dict()}
        if self._use_critic:
            state["critic_optimizer"] = self._critic_optimizer.state_dict()
        return state

    def set_extra_state(self, state: Dict[str, Any]) -> None:
        self._actor_optimizer.load_state_dict(state["actor_optimizer"])
        if self._use_critic and "critic_optimizer" in state:
            self._critic_optimizer.load_state_dict(state["critic_optimizer"])

    def compare(self, other: PolicyLearner) -> str:
        diffs = [super().compare(other)]
        if not isinstance(other, ActorCriticBase):
            diffs.append("other is not an instance of ActorCriticBase")
        else:
            for attr in ("_state_dim", "_use_actor_target", "_use_critic_target",
                         "_use_twin_critic", "_use_critic", "_actor_soft_update_tau",
                         "_critic_soft_update_tau", "_discount_factor", "_actor_learning_rate"):
                if getattr(self, attr) != getattr(other, attr):
                    diffs.append(f"{attr} is different: {getattr(self, attr)} vs "
                                 f"{getattr(other, attr)}")
            names = ["_actor"] + (["_critic"] if self._use_critic else []) + (
                ["_critic_target"] if self._use_critic_target else []) + (
                ["_actor_target"] if self._use_actor_target else [])
            for name in names:
                mine, theirs = getattr(self, name).state_dict(), getattr(other, name).state_dict()
                if mine.keys() != theirs.keys():
                    diffs.append(f"{name} is different: state_dict keys differ")
                    continue
                for k in mine:
                    if not torch.allclose(mine[k].cpu().float(), theirs[k].cpu().float(),
                                          rtol=1e-5, atol=1e-8):
                        diffs.append(f"{name} is different: key {k} differs")
        return "\n".join(d for d in diffs if d)
