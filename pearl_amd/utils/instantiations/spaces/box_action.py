"""A continuous box action space
(pearl/utils/instantiations/spaces/box_action.py:32-120), without the gym dependency: what the
continuous actor-critic learner path reads — ``low`` / ``high`` (1-d float tensors),
``action_dim``, ``shape``, ``sample`` and ``is_continuous``."""
from __future__ import annotations

from typing import Optional, Union

import numpy as np
import torch
from torch import Tensor


class BoxActionSpace:
    def __init__(self, low: Union[float, Tensor], high: Union[float, Tensor],
                 seed: Optional[int] = None) -> None:
        lo = torch.as_tensor(low, dtype=torch.float32).reshape(-1)
        hi = torch.as_tensor(high, dtype=torch.float32).reshape(-1)
        if lo.shape != hi.shape:
            raise ValueError(f"low and high differ in shape: {tuple(lo.shape)} vs {tuple(hi.shape)}")
        self._low, self._high = lo, hi
        self._rng = seed if isinstance(seed, np.random.Generator) else np.random.default_rng(seed)

    @property
    def low(self) -> Tensor:
        return self._low

    @property
    def high(self) -> Tensor:
        return self._high

    @property
    def shape(self) -> torch.Size:
        return self._low.shape

    @property
    def action_dim(self) -> int:
        return int(self._low.shape[0])

    @property
    def is_continuous(self) -> bool:
        return True

    def sample(self, mask: Optional[Tensor] = None) -> Tensor:
        u = self._rng.uniform(size=self._low.shape[0]).astype(np.float32)
        return self._low + (self._high - self._low) * torch.from_numpy(u)

    def to(self, device: torch.device) -> None:
        self._low, self._high = self._low.to(device), self._high.to(device)
