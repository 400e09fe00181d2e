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

from ..common.utils import mlp_block, linear_layers_of_plain


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
        return linear_layers_of_plain(self._model, type(self).__name__)


class VanillaQValueMultiHeadNetwork(QValueNetwork):
    """One output per action: f(s) in R^A, Q(s, a) = onehot(a) . f(s)
    (q_value_networks.py:185-249).  No (B, A, S + AD) expansion: the learner evaluates the trunk
    once per state (generic_q.MultiHeadOps)."""

    def __init__(self, state_dim: int, action_dim: int, hidden_dims: List[int], output_dim: int,
                 use_layer_norm: bool = False) -> None:
        super().__init__()
        self._state_dim, self._action_dim, self._output_dim = int(state_dim), int(action_dim), int(output_dim)
        self._model: nn.Module = mlp_block(input_dim=state_dim, hidden_dims=hidden_dims,
                                           output_dim=output_dim, use_layer_norm=use_layer_norm)

    def forward(self, x: Tensor) -> Tensor:
        return self._model(x)

    def get_q_values(self, state_batch: Tensor, action_batch: Tensor,
                     curr_available_actions_batch: Optional[Tensor] = None) -> Tensor:
        assert self._output_dim == action_batch.shape[-1]
        assert state_batch.ndim == 2 and action_batch.ndim in (2, 3)
        acts = action_batch if action_batch.ndim == 3 else action_batch.unsqueeze(1)
        q = torch.bmm(acts, self.forward(state_batch).unsqueeze(-1)).squeeze(-1)
        return q if action_batch.ndim == 3 else q.squeeze(-1)

    @property
    def state_dim(self) -> int:
        return self._state_dim

    @property
    def action_dim(self) -> int:
        return self._action_dim

    def linear_layers(self) -> List[nn.Linear]:
        return linear_layers_of_plain(self._model, type(self).__name__)


class DuelingQValueNetwork(QValueNetwork):
    """state -> state_arch -> features; value_arch(features) = V(s);
    advantage_arch([features | action]) = A(s, a); Q = V + A - mean(A)
    (q_value_networks.py:352-508; same sub-module names, hence the same ``state_dict`` keys)."""

    def __init__(self, state_dim: int, action_dim: int, hidden_dims: List[int], output_dim: int,
                 value_hidden_dims: Optional[List[int]] = None,
                 advantage_hidden_dims: Optional[List[int]] = None,
                 state_hidden_dims: Optional[List[int]] = None) -> None:
        super().__init__()
        from ..common.value_networks import VanillaValueNetwork
        self._state_dim, self._action_dim = int(state_dim), int(action_dim)
        self.state_arch = VanillaValueNetwork(
            input_dim=state_dim,
            hidden_dims=hidden_dims if state_hidden_dims is None else state_hidden_dims,
            output_dim=hidden_dims[-1])
        self.value_arch = VanillaValueNetwork(
            input_dim=hidden_dims[-1],
            hidden_dims=hidden_dims if value_hidden_dims is None else value_hidden_dims,
            output_dim=output_dim)
        self.advantage_arch = VanillaValueNetwork(
            input_dim=hidden_dims[-1] + action_dim,
            hidden_dims=hidden_dims if advantage_hidden_dims is None else advantage_hidden_dims,
            output_dim=output_dim)

    @property
    def state_dim(self) -> int:
        return self._state_dim

    @property
    def action_dim(self) -> int:
        return self._action_dim

    def get_q_values(self, state_batch: Tensor, action_batch: Tensor,
                     curr_available_actions_batch: Optional[Tensor] = None) -> Tensor:
        assert state_batch.ndim == 2 and action_batch.ndim in (2, 3)
        acts = action_batch if action_batch.ndim == 3 else action_batch.unsqueeze(1)
        feats = self.state_arch(state_batch)
        value = self.value_arch(feats)                                    # (B, 1)

        def adv(actions: Tensor) -> Tensor:
            f = feats.unsqueeze(1).expand(-1, actions.shape[1], -1)
            return self.advantage_arch(torch.cat([f, actions], dim=-1)).squeeze(-1)

        advantage = adv(acts)
        if curr_available_actions_batch is None:
            mean = advantage.mean(dim=-1, keepdim=True)
        else:
            mean = adv(curr_available_actions_batch).mean(dim=-1, keepdim=True)
        q = value + advantage - mean
        return q if action_batch.ndim == 3 else q.squeeze(-1)
