"""Batch wire format between the replay arena and the learner.

Mirrors the contract of the reference's ``Transition`` / ``TransitionBatch``
(pearl/replay_buffers/transition.py:21-72, :89-239): same field names, same
defaults (terminated = ones, truncated = zeros when unset), same shape checks,
``.to(device)`` and ``len()``.  It is the output contract of the gather kernel
(SURVEY.md §8a row a5).
"""
from __future__ import annotations

from typing import Iterator, Optional, Tuple

import torch
from torch import Tensor

_TENSOR_FIELDS: Tuple[str, ...] = (
    "state", "action", "reward", "terminated", "truncated", "next_state", "next_action",
    "curr_available_actions", "curr_unavailable_actions_mask", "next_available_actions",
    "next_unavailable_actions_mask", "weight", "cost",
)


class _FieldBag:
    """Shared plumbing: iterate the populated tensor fields, move them between devices."""

    _fields: Tuple[str, ...] = ()

    def _items(self) -> Iterator[Tuple[str, Tensor]]:
        for name in self._fields:
            value = getattr(self, name)
            if value is not None:
                yield name, value

    def to(self, device: torch.device):
        for name, value in list(self._items()):
            setattr(self, name, torch.as_tensor(value, device=device))
        return self

    @property
    def device(self) -> torch.device:
        return self.state.device

    def __repr__(self) -> str:
        body = ", ".join(f"{k}={tuple(v.shape)}:{str(v.dtype).replace('torch.', '')}"
                         for k, v in self._items())
        return f"{type(self).__name__}({body})"


class Transition(_FieldBag):
    """One stored transition; every tensor carries a leading dim of 1 (transition.py:21-72)."""

    _fields = _TENSOR_FIELDS

    def __init__(self, state: Tensor, action: Tensor, reward: Tensor,
                 terminated: Optional[Tensor] = None, truncated: Optional[Tensor] = None,
                 next_state: Optional[Tensor] = None, next_action: Optional[Tensor] = None,
                 curr_available_actions: Optional[Tensor] = None,
                 curr_unavailable_actions_mask: Optional[Tensor] = None,
                 next_available_actions: Optional[Tensor] = None,
                 next_unavailable_actions_mask: Optional[Tensor] = None,
                 weight: Optional[Tensor] = None, cost: Optional[Tensor] = None) -> None:
        self.state, self.action, self.reward = state, action, reward
        # bandit-friendly defaults of the reference (:51-52)
        self.terminated = torch.tensor(True) if terminated is None else terminated
        self.truncated = torch.tensor(False) if truncated is None else truncated
        self.next_state, self.next_action = next_state, next_action
        self.curr_available_actions = curr_available_actions
        self.curr_unavailable_actions_mask = curr_unavailable_actions_mask
        self.next_available_actions = next_available_actions
        self.next_unavailable_actions_mask = next_unavailable_actions_mask
        self.weight, self.cost = weight, cost


class TransitionBatch(_FieldBag):
    """A batch of transitions, leading dim = batch size (transition.py:89-239)."""

    _fields = _TENSOR_FIELDS + ("time_diff",)

    def __init__(self, state: Tensor, action: Tensor, reward: Tensor,
                 terminated: Optional[Tensor] = None, truncated: Optional[Tensor] = None,
                 next_state: Optional[Tensor] = None, next_action: Optional[Tensor] = None,
                 curr_available_actions: Optional[Tensor] = None,
                 curr_unavailable_actions_mask: Optional[Tensor] = None,
                 next_available_actions: Optional[Tensor] = None,
                 next_unavailable_actions_mask: Optional[Tensor] = None,
                 weight: Optional[Tensor] = None, time_diff: Optional[Tensor] = None,
                 cost: Optional[Tensor] = None) -> None:
        self.state, self.action, self.reward = state, action, reward
        self.terminated, self.truncated = terminated, truncated
        self.next_state, self.next_action = next_state, next_action
        self.curr_available_actions = curr_available_actions
        self.curr_unavailable_actions_mask = curr_unavailable_actions_mask
        self.next_available_actions = next_available_actions
        self.next_unavailable_actions_mask = next_unavailable_actions_mask
        self.weight, self.time_diff, self.cost = weight, time_diff, cost
        self._validate()

    def _validate(self) -> None:
        # the reference's __post_init__ checks (:133-217), same messages in spirit
        assert self.state.ndim >= 2, (
            f"state has shape {tuple(self.state.shape)}, but must have at least 2 dimensions "
            "(batch_size, ...)")
        assert self.action.ndim >= 1, (
            f"action has shape {tuple(self.action.shape)}, but must be (batch_size,) or "
            "(batch_size, ...)")
        assert self.reward.ndim >= 1, (
            f"reward has shape {tuple(self.reward.shape)}, but must be (batch_size,) or "
            "(batch_size, ...)")
        n = self.reward.shape[0]
        assert self.state.shape[0] == n, (
            f"state has shape {tuple(self.state.shape)}, but reward has shape "
            f"{tuple(self.reward.shape)}, and they must have the same batch_size dimension")
        for name, fill in (("terminated", True), ("truncated", False)):
            flag = getattr(self, name)
            if flag is None:
                setattr(self, name, torch.full((n,), fill, dtype=torch.bool,
                                               device=self.reward.device))
                continue
            ok = (flag.ndim == 1 and flag.shape[0] == n) or (
                flag.ndim == 2 and tuple(flag.shape) == (n, 1))
            assert ok, (f"{name} has shape {tuple(flag.shape)} but it should be equal to either "
                        f"({n},) or ({n}, 1) (since batch_size is {n})")
        if self.next_state is not None:
            assert self.next_state.ndim >= 2, (
                f"next_state has shape {tuple(self.next_state.shape)}, but must have at least 2 "
                f"dimensions ({n}, ...)")
        if self.next_action is not None:
            assert self.next_action.ndim >= 1, (
                f"next_action has shape {tuple(self.next_action.shape)}, but must be ({n},) or "
                f"({n}, ...)")

    def __len__(self) -> int:
        return self.reward.shape[0]


class TransitionWithBootstrapMask(Transition):
    """``Transition`` + the (1, ensemble_size) Bernoulli mask of Bootstrapped DQN
    (transition.py:242-244)."""

    _fields = _TENSOR_FIELDS + ("bootstrap_mask",)

    def __init__(self, *args, bootstrap_mask: Optional[Tensor] = None, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.bootstrap_mask = bootstrap_mask


class TransitionWithBootstrapMaskBatch(TransitionBatch):
    """``TransitionBatch`` + ``bootstrap_mask`` (batch_size, ensemble_size) (transition.py:247-249)."""

    _fields = TransitionBatch._fields + ("bootstrap_mask",)

    def __init__(self, *args, bootstrap_mask: Optional[Tensor] = None, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        self.bootstrap_mask = bootstrap_mask


def filter_batch_by_bootstrap_mask(batch: TransitionWithBootstrapMaskBatch, z) -> TransitionBatch:
    """The transitions of `batch` whose mask is active for ensemble member `z`
    (transition.py:252-301); fields the batch does not carry stay None, ``weight`` / ``cost`` /
    ``time_diff`` are dropped exactly as the reference drops them."""
    mask = batch.bootstrap_mask

    def keep(x: Optional[Tensor]) -> Optional[Tensor]:
        if x is None or mask is None:
            return None
        return x[mask[:, z] == 1]

    required = [keep(getattr(batch, k)) for k in ("state", "action", "reward", "terminated",
                                                  "truncated")]
    assert all(v is not None for v in required)
    return TransitionBatch(
        state=required[0], action=required[1], reward=required[2], terminated=required[3],
        truncated=required[4], next_state=keep(batch.next_state),
        next_action=keep(batch.next_action),
        curr_available_actions=keep(batch.curr_available_actions),
        curr_unavailable_actions_mask=keep(batch.curr_unavailable_actions_mask),
        next_available_actions=keep(batch.next_available_actions),
        next_unavailable_actions_mask=keep(batch.next_unavailable_actions_mask))
