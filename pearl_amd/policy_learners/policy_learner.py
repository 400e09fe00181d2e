"""The ``PolicyLearner`` plugin interface and its generic ``learn`` loop
(pearl/policy_learners/policy_learner.py:40-303).

``learn`` = ``training_rounds`` x (sample -> preprocess_batch -> learn_batch) with the
reference's batch-size clamp and report aggregation (:162-195); ``preprocess_batch`` applies the
history-summarisation and action-representation modules in place (:197-218).  Learners in this
package override ``learn`` with a fused device-side loop when the buffer is an HBM arena and keep
this generic loop for any other ``ReplayBuffer``.
"""
from __future__ import annotations

import functools
import os
import time
from abc import ABC, abstractmethod
from typing import Any, Callable, Dict, List, Optional

import torch
import torch.nn as nn

from ..action_representation_modules import (ActionRepresentationModule,
                                             IdentityActionRepresentationModule)
from ..replay_buffers.replay_buffer import ReplayBuffer
from ..replay_buffers.transition import TransitionBatch
from .exploration import ExplorationModule, NoExploration


class IdentityHistorySummarizationModule(nn.Module):
    """forward = x (pearl/history_summarization_modules/identity_history_summarization_module.py
    :41-42); the only history module on the configured hot path."""

    def __init__(self) -> None:
        super().__init__()
        self.history: Any = None

    def summarize_history(self, observation: Any, action: Any) -> Any:
        self.history = observation
        return observation

    def get_history(self) -> Any:
        return self.history

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def reset(self) -> None:
        self.history = None

    def compare(self, other: Any) -> str:
        return "" if isinstance(other, IdentityHistorySummarizationModule) else (
            "other is not an instance of IdentityHistorySummarizationModule")


def perf_reported(learn: Callable[..., Dict[str, Any]]) -> Callable[..., Dict[str, Any]]:
    """Opt-in performance fields of ``learn()``'s report (the dict the reference
    aggregates at policy_learner.py:181-195).  With ``learner.performance_report`` False — the
    default — the wrapped ``learn`` runs untouched: same keys as the reference, no extra
    synchronisation, no timers.  With it True the report of a non-empty call gains, as one-element
    lists (the report is a dict of lists):

      perf/transitions_per_s   batch_size x rounds of this call / its wall time
      perf/learn_wall_us       wall time of the call (host clock around it; the call's own final
                               synchronisation included, plus one device synchronisation after it)
      perf/rounds              rounds this call ran
      perf/kernel_us/<stage>   average duration of the stage's launches inside this call, from the
                               library's HIP-event timers (``pa_dqn_get_timing``, ``pa_mlp_timing_read``,
                               ``pa_sac_timing_read``): what rocprofv3's kernel trace shows for the
                               same launches, without a profiler attached
    """
    @functools.wraps(learn)
    def wrapper(self: "PolicyLearner", replay_buffer: ReplayBuffer) -> Dict[str, Any]:
        if not self.performance_report or self.__dict__.get("_perf_active"):
            return learn(self, replay_buffer)
        self.__dict__["_perf_active"] = True
        try:
            steps0 = self._training_steps
            self._perf_begin()
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            t0 = time.perf_counter()
            report = learn(self, replay_buffer)
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            stages = self._perf_end()
            rounds = self._training_steps - steps0
            if report and rounds > 0:
                bs = self._clamped_batch_size(replay_buffer) if len(replay_buffer) else self._batch_size
                report["perf/transitions_per_s"] = [bs * rounds / dt]
                report["perf/learn_wall_us"] = [dt * 1e6]
                report["perf/rounds"] = [rounds]
                for k, v in stages.items():
                    report[f"perf/kernel_us/{k}"] = [v]
            return report
        finally:
            self.__dict__["_perf_active"] = False
    return wrapper


def _looks_like_batch(obj: Any) -> bool:
    return isinstance(obj, TransitionBatch) or all(
        hasattr(obj, k) for k in ("state", "action", "reward", "terminated", "next_state"))


class PolicyLearner(nn.Module, ABC):
    def __init__(self, on_policy: bool, is_action_continuous: bool, action_space: Any = None,
                 training_rounds: int = 100, batch_size: int = 1, requires_tensors: bool = True,
                 action_representation_module: Optional[ActionRepresentationModule] = None,
                 **options: Any) -> None:
        super().__init__()
        self.exploration_module = options.get("exploration_module") or NoExploration()
        if action_representation_module is None:
            if action_space is not None:
                action_representation_module = IdentityActionRepresentationModule(
                    max_number_actions=getattr(action_space, "n", None),
                    representation_dim=action_space.action_dim)
            else:
                action_representation_module = IdentityActionRepresentationModule()
        else:
            assert action_representation_module.representation_dim is not None
        self.action_representation_module = action_representation_module
        self._history_summarization_module: nn.Module = IdentityHistorySummarizationModule()
        self._training_rounds = training_rounds
        self._batch_size = batch_size
        self._training_steps = 0
        # opt-in performance fields of learn()'s report (perf_reported above); PEARL_AMD_PERF_REPORT=1
        # switches them on for every learner of the process
        self.performance_report: bool = os.environ.get("PEARL_AMD_PERF_REPORT") == "1"
        self.on_policy = on_policy
        self._is_action_continuous = is_action_continuous
        self.distribution_enabled: bool = (torch.distributed.is_available()
                                           and torch.distributed.is_initialized())
        self.requires_tensors = requires_tensors

    @property
    def batch_size(self) -> int:
        return self._batch_size

    def get_action_representation_module(self) -> ActionRepresentationModule:
        return self.action_representation_module

    @abstractmethod
    def set_history_summarization_module(self, value: nn.Module) -> None:
        ...

    def reset(self, action_space: Any) -> None:
        pass

    @abstractmethod
    def act(self, subjective_state: Any, available_action_space: Any, exploit: bool = False) -> Any:
        ...

    def _clamped_batch_size(self, replay_buffer: ReplayBuffer) -> int:
        n = len(replay_buffer)
        return n if (self._batch_size == -1 or n < self._batch_size) else self._batch_size

    def invalidate_native_copies(self) -> None:
        """Tell the HIP library that parameters were written behind its back.

        The learners keep derived copies of the weights (MFMA fragment-major layouts, bf16 split
        planes).  ``learn()`` rebuilds them from the parameters at the start of every call, and
        ``learn_batch()`` notices writes that bump torch's version counters (``load_state_dict``,
        in-place ops, a torch optimizer).  Writes through ``param.data`` — the reference's own
        ``update_target_network`` idiom (common/utils.py:214-226) — are invisible to those counters:
        after such a write, call this before the next ``learn_batch()``.  (No reference counterpart.)"""
        flat = getattr(self, "_flat", None)
        if isinstance(flat, dict):
            for m in flat.values():
                if hasattr(m, "invalidate"):
                    m.invalidate()
        nat = getattr(self, "_native", None)
        if nat is not None and getattr(nat, "handle", None) is not None:
            from .. import _native as N
            N.check(N.lib().pa_dqn_invalidate(nat.handle))

    # -- hooks of the opt-in performance report: switch the library's event timers on / read them
    def _perf_begin(self) -> None:
        """Default: the generic MLP engine's timers (fused row step, weight gradients + AdamW)."""
        from .. import _native as N
        if torch.cuda.is_available():
            N.check(N.lib().pa_mlp_timing(1))

    def _perf_end(self) -> Dict[str, float]:
        import ctypes as C
        from .. import _native as N
        out: Dict[str, float] = {}
        if not torch.cuda.is_available():
            return out
        for which, name in ((0, "row_step"), (1, "weight_grad_adamw")):
            us, n = C.c_double(), C.c_int64()
            N.check(N.lib().pa_mlp_timing_read(which, C.byref(us), C.byref(n)))
            if n.value:
                out[name] = us.value
        N.check(N.lib().pa_mlp_timing(0))
        return out

    @perf_reported
    def learn(self, replay_buffer: ReplayBuffer) -> Dict[str, Any]:
        if len(replay_buffer) == 0:
            return {}
        batch_size = self._clamped_batch_size(replay_buffer)
        report: Dict[str, List[Any]] = {}
        for _ in range(self._training_rounds):
            self._training_steps += 1
            batch = replay_buffer.sample(batch_size)
            single: Dict[str, Any] = {}
            if _looks_like_batch(batch):
                single = self.learn_batch(self.preprocess_batch(batch))
            for k, v in single.items():
                report.setdefault(k, []).append(v)
        return report

    def preprocess_batch(self, batch: TransitionBatch) -> TransitionBatch:
        batch.state = self._history_summarization_module(batch.state)
        with torch.no_grad():
            batch.next_state = self._history_summarization_module(batch.next_state)
        rep = self.action_representation_module
        batch.action = rep(batch.action)
        for name in ("next_action", "curr_available_actions", "next_available_actions"):
            value = getattr(batch, name, None)
            if value is not None:
                setattr(batch, name, rep(value))
        return batch

    @abstractmethod
    def learn_batch(self, batch: TransitionBatch) -> Dict[str, Any]:
        ...

    def __str__(self) -> str:
        return self.__class__.__name__

    def compare(self, other: "PolicyLearner") -> str:
        if not isinstance(other, PolicyLearner):
            return "other is not an instance of PolicyLearner"
        diffs: List[str] = []
        for attr in ("_training_rounds", "_batch_size", "on_policy", "_is_action_continuous"):
            a, b = getattr(self, attr), getattr(other, attr)
            if a != b:
                diffs.append(f"{attr} is different: {a} vs {b}")
        for label, mine, theirs in (
                ("exploration_module", self.exploration_module, other.exploration_module),
                ("action_representation_module", self.action_representation_module,
                 other.action_representation_module),
                ("history summarization module", self._history_summarization_module,
                 other._history_summarization_module)):
            reason = mine.compare(theirs)
            if reason:
                diffs.append(f"{label} is different: {reason}")
        return "\n".join(diffs)


def accept_optimizer(optimizer: Any, params: Any, who: str) -> Any:
    """A caller-supplied optimizer (deep_td_learning.py:183-185, actor_critic_base.py:159-211: the
    reference uses whatever it is handed).  The HIP learner steps the parameters itself, with
    torch.optim.AdamW's arithmetic and the hyper-parameters read from the optimizer's parameter group
    (lr, betas, eps, weight_decay, amsgrad) — so what can be honoured is exactly: ``optim.AdamW``, or
    ``optim.Adam`` with weight_decay 0 (the same update), over the learner network's own parameters in
    ONE group, without maximize / capturable / differentiable.  The optimizer object stays the owner of
    the state (``state_dict`` / ``load_state_dict`` / per-parameter ``step`` keep working); anything
    else is refused, loudly — never silently replaced by AdamW."""
    import torch.optim as optim
    params = list(params)
    ok_type = type(optimizer) is optim.AdamW or (
        type(optimizer) is optim.Adam and all(g["weight_decay"] == 0 for g in optimizer.param_groups))
    if not ok_type:
        raise NotImplementedError(
            f"pearl_amd {who}: the HIP step implements torch.optim.AdamW (or Adam with weight_decay 0); "
            f"got {type(optimizer).__name__}")
    if len(optimizer.param_groups) != 1:
        raise NotImplementedError(f"pearl_amd {who}: one parameter group expected, got "
                                  f"{len(optimizer.param_groups)}")
    g = optimizer.param_groups[0]
    theirs = list(g["params"])
    if len(theirs) != len(params) or any(a is not b for a, b in zip(theirs, params)):
        raise NotImplementedError(
            f"pearl_amd {who}: the optimizer must own exactly the learner network's parameters, in "
            "their order (build it from network.parameters())")
    for flag in ("maximize", "capturable", "differentiable"):
        if g.get(flag, False):
            raise NotImplementedError(f"pearl_amd {who}: optimizer option {flag}=True is not built")
    g.setdefault("amsgrad", False)
    return optimizer
