"""State-value network container (pearl/neural_networks/common/value_networks.py:35-59).

Owns the parameters (same ``state_dict`` keys and default init as the reference); the learner
step runs through ``pa_mlp_*`` on a flat view of them, ``forward`` is the torch expression of the
same function for act-time / inspection."""
from __future__ import annotations

from typing import Any, List, Optional

import torch.nn as nn
from torch import Tensor

from .utils import mlp_block, linear_layers_of_plain


class ValueNetwork(nn.Module):
    """Umbrella type of all value networks."""


class VanillaValueNetwork(ValueNetwork):
    def __init__(self, input_dim: int, hidden_dims: Optional[List[int]], output_dim: int = 1,
                 **kwargs: Any) -> None:
        super().__init__()
        self._model: nn.Module = mlp_block(input_dim=input_dim, hidden_dims=hidden_dims,
                                           output_dim=output_dim, **kwargs)

    def forward(self, x: Tensor) -> Tensor:
        return self._model(x)

    def linear_layers(self) -> List[nn.Linear]:
        return linear_layers_of_plain(self._model, type(self).__name__)
