"""Actor network containers
(pearl/neural_networks/sequential_decision_making/actor_networks.py:29-73, :107-176, :448-485,
:488-629).

Same constructors, ``state_dict`` keys and torch semantics as the reference.  The learner step
does not call these ``forward``s: it runs ``pa_mlp_forward`` on a flat view of the parameters and
the softmax / tanh-Gaussian heads of libpearl_amd (``pa_softmax_action_prob``,
``pa_gauss_sample``).  The torch expressions below serve ``act()`` (outside the measured path)."""
from __future__ import annotations

from typing import Any, List, Optional, Tuple, Union

import torch
import torch.nn as nn
from torch import Tensor
from torch.distributions import Normal

from ..common.utils import mlp_block, linear_layers_of_plain


def action_scaling(action_space: Any, input_action: Tensor) -> Tensor:
    """[-1, 1]^d -> [low, high]^d (actor_networks.py:29-51)."""
    low = action_space.low.clone().detach().to(input_action.device)
    high = action_space.high.clone().detach().to(input_action.device)
    return (((high - low) * (input_action + 1.0)) / 2) + low


def action_unscaling(action_space: Any, input_action: Tensor) -> Tensor:
    """[low, high]^d -> [-1, 1]^d (actor_networks.py:54-73)."""
    low = action_space.low.clone().detach().to(input_action.device)
    high = action_space.high.clone().detach().to(input_action.device)
    return ((input_action - low) / (high - low)) * 2.0 - 1.0


class ActorNetwork(nn.Module):
    def __init__(self, input_dim: int, hidden_dims: Optional[List[int]], output_dim: int,
                 action_space: Any = None) -> None:
        super().__init__()


class VanillaActorNetwork(ActorNetwork):
    """Softmax policy over a discrete action set (:107-176)."""

    def __init__(self, input_dim: int, hidden_dims: Optional[List[int]], output_dim: int,
                 action_space: Any = None) -> None:
        super().__init__(input_dim, hidden_dims, output_dim, action_space)
        self._model: nn.Module = mlp_block(input_dim=input_dim, hidden_dims=hidden_dims,
                                           output_dim=output_dim, last_activation="softmax")

    def forward(self, x: Tensor) -> Tensor:
        return self._model(x)

    def get_policy_distribution(self, state_batch: Tensor, available_actions: Optional[Tensor] = None,
                                unavailable_actions_mask: Optional[Tensor] = None) -> Tensor:
        return self.forward(state_batch)

    def get_action_prob(self, state_batch: Tensor, action_batch: Tensor,
                        available_actions: Optional[Tensor] = None,
                        unavailable_actions_mask: Optional[Tensor] = None) -> Tensor:
        all_action_probs = self.forward(state_batch)
        return torch.sum(all_action_probs * action_batch, dim=1, keepdim=True).view(-1)

    def linear_layers(self) -> List[nn.Linear]:
        return linear_layers_of_plain(self._model, type(self).__name__)


class VanillaContinuousActorNetwork(ActorNetwork):
    """Deterministic policy: mlp with a tanh output, scaled to the action box (:448-485)."""

    def __init__(self, input_dim: int, hidden_dims: Optional[List[int]], output_dim: int,
                 action_space: Any) -> None:
        super().__init__(input_dim, hidden_dims, output_dim, action_space)
        self._model: nn.Module = mlp_block(input_dim=input_dim, hidden_dims=hidden_dims,
                                           output_dim=output_dim, last_activation="tanh")
        self._action_space = action_space

    def forward(self, x: Tensor) -> Tensor:
        return self._model(x)

    def sample_action(self, x: Tensor) -> Tensor:
        return action_scaling(self._action_space, self._model(x))

    def linear_layers(self) -> List[nn.Linear]:
        return linear_layers_of_plain(self._model, type(self).__name__)


class GaussianActorNetwork(ActorNetwork):
    """tanh-squashed diagonal Gaussian policy (:488-629)."""

    def __init__(self, input_dim: int, hidden_dims: List[int], output_dim: int,
                 action_space: Any) -> None:
        super().__init__(input_dim, hidden_dims, output_dim, action_space)
        if len(hidden_dims) < 1:
            raise ValueError("The hidden dims cannot be empty for a gaussian actor network.")
        self._model: nn.Module = mlp_block(input_dim=input_dim, hidden_dims=hidden_dims[:-1],
                                           output_dim=hidden_dims[-1], last_activation="relu")
        self.fc_mu = nn.Linear(hidden_dims[-1], output_dim)
        self.fc_std = nn.Linear(hidden_dims[-1], output_dim)
        self._action_space = action_space
        assert hasattr(action_space, "low") and hasattr(action_space, "high"), \
            "GaussianActorNetwork needs a box action space"
        self.register_buffer("_action_bound",
                             (action_space.high.clone().detach() - action_space.low.clone().detach()) / 2)
        self._log_std_min = -5
        self._log_std_max = 2

    def forward(self, x: Tensor) -> Tuple[Tensor, Tensor]:
        x = self._model(x)
        mean = self.fc_mu(x)
        log_std = torch.tanh(self.fc_std(x))
        log_std = self._log_std_min + 0.5 * (self._log_std_max - self._log_std_min) * (log_std + 1)
        return mean, log_std

    def sample_action(self, state_batch: Tensor, get_log_prob: bool = False
                      ) -> Union[Tensor, Tuple[Tensor, Tensor]]:
        epsilon = 1e-6
        mean, log_std = self.forward(state_batch)
        normal = Normal(mean, log_std.exp())
        sample = normal.rsample()
        normalized_action = torch.tanh(sample)
        action = action_scaling(self._action_space, normalized_action)
        log_prob = normal.log_prob(sample)
        log_prob = log_prob - torch.log(self._action_bound * (1 - normalized_action.pow(2)) + epsilon)
        if log_prob.dim() == 2:
            log_prob = log_prob.sum(dim=1, keepdim=True)
        return (action, log_prob) if get_log_prob else action

    def get_log_probability(self, state_batch: Tensor, action_batch: Tensor) -> Tensor:
        """log pi(a | s) of given actions (:593-629)."""
        epsilon = 1e-6
        mean, log_std = self.forward(state_batch)
        normal = Normal(mean, log_std.exp())
        normalized = torch.clip(action_unscaling(self._action_space, action_batch), -1 + epsilon,
                                1 - epsilon)
        log_prob = normal.log_prob(torch.atanh(normalized))
        log_prob = log_prob - torch.log(self._action_bound * (1 - normalized.pow(2)) + epsilon)
        if log_prob.dim() == 2:
            log_prob = log_prob.sum(dim=1, keepdim=True)
        return log_prob

    def trunk_layers(self) -> List[nn.Linear]:
        return linear_layers_of_plain(self._model, type(self).__name__)
