"""Two independently initialised Q networks
(pearl/neural_networks/sequential_decision_making/twin_critic.py:22-91): same constructor,
``state_dict`` keys (``_critic_1.*``, ``_critic_2.*``, ``_critic_networks_combined.*``) and init
order as the reference."""
from __future__ import annotations

from typing import Callable, Iterable, Optional, Tuple

import torch.nn as nn
from torch import Tensor

from ..common.utils import xavier_init_weights
from .q_value_networks import QValueNetwork, VanillaQValueNetwork


class TwinCritic(nn.Module):
    def __init__(self, state_dim: Optional[int] = None, action_dim: Optional[int] = None,
                 hidden_dims: Optional[Iterable[int]] = None,
                 init_fn: Callable[[nn.Module], None] = xavier_init_weights,
                 network_type: type = VanillaQValueNetwork, output_dim: int = 1,
                 network_instance_1: Optional[QValueNetwork] = None,
                 network_instance_2: Optional[QValueNetwork] = None) -> None:
        super().__init__()
        if network_instance_1 is not None and network_instance_2 is not None:
            self._critic_1, self._critic_2 = network_instance_1, network_instance_2
        else:
            assert state_dim is not None and action_dim is not None and hidden_dims is not None
            if network_type is not VanillaQValueNetwork:
                raise NotImplementedError("pearl_amd TwinCritic: only VanillaQValueNetwork critics "
                                          "have HIP kernels")
            self._critic_1 = network_type(state_dim=state_dim, action_dim=action_dim,
                                          hidden_dims=list(hidden_dims), output_dim=output_dim)
            self._critic_2 = network_type(state_dim=state_dim, action_dim=action_dim,
                                          hidden_dims=list(hidden_dims), output_dim=output_dim)
        self._critic_networks_combined = nn.ModuleList([self._critic_1, self._critic_2])
        self._critic_networks_combined.apply(init_fn)

    def get_q_values(self, state_batch: Tensor, action_batch: Tensor) -> Tuple[Tensor, Tensor]:
        return (self._critic_1.get_q_values(state_batch, action_batch),
                self._critic_2.get_q_values(state_batch, action_batch))
