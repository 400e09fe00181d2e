"""Identity action representation
(pearl/action_representation_modules/identity_action_representation_module.py:17-73)."""
from __future__ import annotations

from typing import Optional

import torch

from .action_representation_module import ActionRepresentationModule


class IdentityActionRepresentationModule(ActionRepresentationModule):
    def __init__(self, max_number_actions: Optional[int] = None,
                 representation_dim: Optional[int] = None) -> None:
        super().__init__()
        self._max_number_actions = max_number_actions
        self._representation_dim = representation_dim

    @property
    def max_number_actions(self) -> Optional[int]:
        return self._max_number_actions

    @property
    def representation_dim(self) -> Optional[int]:
        return self._representation_dim

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x

    def compare(self, other: ActionRepresentationModule) -> str:
        if not isinstance(other, IdentityActionRepresentationModule):
            return "other is not an instance of IdentityActionRepresentationModule"
        diffs = []
        if self.max_number_actions != other.max_number_actions:
            diffs.append(f"max_number_actions is different: {self.max_number_actions} vs "
                         f"{other.max_number_actions}")
        if self.representation_dim != other.representation_dim:
            diffs.append(f"representation_dim is different: {self.representation_dim} vs "
                         f"{other.representation_dim}")
        return "\n".join(diffs)
