"""A finite set of action tensors
(pearl/utils/instantiations/spaces/discrete_action.py:32-111), without the gym dependency.

Only what the replay/learner path reads is kept: ``n``, ``actions``, ``actions_batch``,
``action_dim``, ``sample``, ``to`` and item access.  Any object with ``n`` / ``actions_batch`` /
``action_dim`` (e.g. the reference's own DiscreteActionSpace) is accepted wherever this type
is used.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch
from torch import Tensor


class DiscreteActionSpace:
    def __init__(self, actions: List[Tensor], seed: Optional[int] = None) -> None:
        if len(actions) == 0:
            raise ValueError("`DiscreteActionSpace` requires at least one action.")
        flat = [torch.as_tensor(a).reshape(-1) for a in actions]
        want = flat[0].shape
        for a in flat:
            if a.shape != want:
                raise ValueError(
                    f"All actions must have the same shape. Expected {want}, but got {a.shape}.")
        self.elements: List[Tensor] = flat
        self._rng = seed if isinstance(seed, np.random.Generator) else np.random.default_rng(seed)

    @property
    def n(self) -> int:
        return len(self.elements)

    @property
    def actions(self) -> List[Tensor]:
        return self.elements

    @property
    def actions_batch(self) -> Tensor:
        return torch.stack(self.elements, dim=0)

    @property
    def action_dim(self) -> int:
        return int(self.elements[0].shape[0])

    @property
    def shape(self) -> torch.Size:
        return self.elements[0].shape

    @property
    def is_continuous(self) -> bool:
        return False

    def __getitem__(self, index: int) -> Tensor:
        return self.elements[index]

    def __len__(self) -> int:
        return self.n

    def __iter__(self):
        return iter(self.elements)

    def sample(self, mask: Optional[Tensor] = None) -> Tensor:
        if mask is not None:
            valid = np.flatnonzero(np.asarray(torch.as_tensor(mask).cpu()) == 1)
            k = int(self._rng.choice(valid)) if len(valid) else 0
        else:
            k = int(self._rng.integers(self.n))
        return self.elements[k]

    def to(self, device: torch.device) -> None:
        self.elements = [a.to(device) for a in self.elements]
