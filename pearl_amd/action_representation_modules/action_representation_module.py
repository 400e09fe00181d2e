"""Action-representation plugin interface
(pearl/action_representation_modules/action_representation_module.py:17-50)."""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Optional

import torch
import torch.nn as nn


class ActionRepresentationModule(ABC, nn.Module):
    @property
    @abstractmethod
    def max_number_actions(self) -> Optional[int]:
        ...

    @property
    @abstractmethod
    def representation_dim(self) -> Optional[int]:
        ...

    @abstractmethod
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        ...

    @abstractmethod
    def compare(self, other: "ActionRepresentationModule") -> str:
        """'' when equal, else a description of the differences."""
