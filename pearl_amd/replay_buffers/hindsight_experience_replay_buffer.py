"""HindsightExperienceReplayBuffer
(reference: pearl/replay_buffers/sequential_decision_making/
hindsight_experience_replay_buffer.py:19-160; Andrychowicz et al. 2017, "final" strategy).

Every push is stored as it comes and remembered; when the episode ends (terminated or truncated)
the whole trajectory is pushed AGAIN with the goal slot of ``state`` / ``next_state`` (their last
``goal_dim`` entries) overwritten by ``next_state[:-goal_dim]`` of the final transition and the
reward — optionally also ``terminated`` — recomputed by the caller's functions (:117-160).  Like the
reference, the relabelling writes into the caller's state tensors in place.

Storage is ``BasicReplayBuffer``'s HBM arena; the relabelling is host-side bookkeeping on the
``push`` path (outside the learner hot path) and adds no kernels.
"""
from __future__ import annotations

from typing import Any, Callable, List, Optional, Tuple

import torch

from .basic_replay_buffer import BasicReplayBuffer


class HindsightExperienceReplayBuffer(BasicReplayBuffer):
    def __init__(self, capacity: int, goal_dim: int, reward_fn: Callable[[Any, Any], Any],
                 terminated_fn: Optional[Callable[[Any, Any], bool]] = None,
                 sampler: str = "device", staging_rows: int = 0) -> None:
        super().__init__(capacity, sampler=sampler, staging_rows=staging_rows)
        self._goal_dim = goal_dim
        self._reward_fn = reward_fn
        self._terminated_fn = terminated_fn
        self._trajectory: List[Tuple] = []

    def push(self, state: Any, action: Any, reward: Any, terminated: bool, truncated: bool,
             curr_available_actions: Any = None, next_state: Any = None,
             next_available_actions: Any = None, max_number_actions: Optional[int] = None,
             cost: Optional[float] = None) -> None:
        assert isinstance(next_state, torch.Tensor), "next_state must be a tensor"
        super().push(state, action, reward, terminated, truncated, curr_available_actions,
                     next_state, next_available_actions, max_number_actions, cost)
        if curr_available_actions is None:
            raise ValueError(f"{type(self)} requires curr_available_actions not to be None")
        if next_available_actions is None:
            raise ValueError(f"{type(self)} requires next_available_actions not to be None")
        self._trajectory.append((state, action, next_state, curr_available_actions,
                                 next_available_actions, terminated, truncated,
                                 max_number_actions, cost))
        if terminated or truncated:
            additional_goal = next_state[: -self._goal_dim]      # "final" strategy (:117)
            for (st, act, nst, ca, na, term, trunc, mna, c) in self._trajectory:
                assert isinstance(st, torch.Tensor) and isinstance(nst, torch.Tensor)
                st[-self._goal_dim:] = additional_goal
                nst[-self._goal_dim:] = additional_goal
                super().push(st, act, self._reward_fn(st, act),
                             term if self._terminated_fn is None else self._terminated_fn(st, act),
                             trunc, ca, nst, na, mna, c)
            self._trajectory = []
