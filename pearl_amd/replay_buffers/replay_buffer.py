"""The ``ReplayBuffer`` plugin interface (pearl/replay_buffers/replay_buffer.py:18-91).

Same abstract surface as the reference: ``device_for_batches`` (get/set), ``push``,
``sample``, ``clear``, ``__len__`` and ``is_action_continuous`` (also written directly as
``_is_action_continuous`` by the agent, pearl_agent.py:114-116).  One deliberate
difference in the *implementations* below this ABC: storage lives in HBM, not on the CPU
(SURVEY.md §7 "hard parts", test_replay_buffer_not_stored_on_gpu).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Any, Optional

import torch


class ReplayBuffer(ABC):
    def __init__(self) -> None:
        super().__init__()
        self._is_action_continuous: bool = False

    @property
    @abstractmethod
    def device_for_batches(self) -> torch.device:
        """Device on which sampled batches are returned."""

    @device_for_batches.setter
    @abstractmethod
    def device_for_batches(self, new_device_for_batches: torch.device) -> None:
        ...

    @abstractmethod
    def push(self, state: Any, action: Any, reward: Any, terminated: bool, truncated: bool,
             curr_available_actions: Any = None, next_state: Any = None,
             next_available_actions: Any = None, max_number_actions: Optional[int] = None,
             cost: Optional[float] = None) -> None:
        """Saves a transition."""

    @abstractmethod
    def sample(self, batch_size: int) -> object:
        ...

    @abstractmethod
    def clear(self) -> None:
        """Empties the buffer."""

    @abstractmethod
    def __len__(self) -> int:
        ...

    def __str__(self) -> str:
        return self.__class__.__name__

    @property
    def is_action_continuous(self) -> bool:
        return self._is_action_continuous

    @is_action_continuous.setter
    def is_action_continuous(self, value: bool) -> None:
        self._is_action_continuous = value
