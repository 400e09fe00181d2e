"""Q(s, a) network container
(pearl/neural_networks/sequential_decision_making/q_value_networks.py:34-69, :124-182).

``VanillaQValueNetwork`` is the nn.Module that owns the parameters: ``state_dict`` keys
(``_model.0.0.weight`` ... ``_model.2.0.bias``) and default initialisation are the reference's,
so checkpoints are interchangeable.  The learner step does NOT call ``forward``: the HIP
kernels read and update these parameters in place through a flat view
(pearl_amd/policy_learners/sequential_decision_making/deep_q_learning.py).  ``get_q_values`` is
kept as the torch expression of the same function for ``act()`` (outside the measured path).
"""
from __future__ import annotations

from abc import ABC, abstractmethod
from typing import List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from ..common.utils import mlp_block


class QValueNetwork(ABC, nn.Module):
    @property
    @abstractmethod
    def state_dim(self) -> int:
        ...

    @property
    @abstractmethod
    def action_dim(self) -> int:
        ...

    @abstractmethod
    def get_q_values(self, state_batch: Tensor, action_batch: Tensor,
                     curr_available_actions_batch: Optional[Tensor] = None) -> Tensor:
        ...


class VanillaQValueNetwork(QValueNetwork):
    def __init__(self, state_dim: int, action_dim: int, hidden_dims: List[int], output_dim: int,
                 use_layer_norm: bool = False) -> None:
        super().__init__()
        self._state_dim = int(state_dim)
        self._action_dim = int(action_dim)
        self._model: nn.Module = mlp_block(input_dim=state_dim + action_dim,
                                           hidden_dims=hidden_dims, output_dim=output_dim,
                                           use_layer_norm=use_layer_norm)

    def forward(self, x: Tensor) -> Tensor:
        return self._model(x)

    def get_q_values(self, state_batch: Tensor, action_batch: Tensor,
                     curr_available_actions_batch: Optional[Tensor] = None) -> Tensor:
        # (B, S) x (B, A, d) -> (B, A);  (B, S) x (B, d) -> (B,)     (:152-174)
        assert state_batch.ndim == 2 and action_batch.ndim in (2, 3)
        acts = action_batch if action_batch.ndim == 3 else action_batch.unsqueeze(1)
        states = state_batch.unsqueeze(1).expand(-1, acts.shape[1], -1)
        q = self.forward(torch.cat([states, acts], dim=-1)).squeeze(-1)
        return q if action_batch.ndim == 3 else q.squeeze(-1)

    @property
    def state_dim(self) -> int:
        return self._state_dim

    @property
    def action_dim(self) -> int:
        return self._action_dim

    def linear_layers(self) -> List[nn.Linear]:
        return [m for m in self._model.modules() if isinstance(m, nn.Linear)]
