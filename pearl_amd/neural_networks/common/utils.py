"""Network construction helpers on the learner path
(pearl/neural_networks/common/utils.py:75-152 mlp_block, :201-205 xavier init).

Only the plain configuration the DQN/actor-critic configs use is built here (Linear + ReLU
hidden layers, optional last activation); the reference's layer-norm / batch-norm / dropout /
residual options change the math of the fused kernels and are rejected loudly.
"""
from __future__ import annotations

from typing import List, Optional

import torch.nn as nn

class _Softmax(nn.Softmax):
    def __init__(self) -> None:
        super().__init__(dim=-1)


_ACTIVATIONS = {"relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "linear": nn.Identity,
                "softmax": _Softmax}


def mlp_block(input_dim: int, hidden_dims: Optional[List[int]], output_dim: int = 1,
              use_batch_norm: bool = False, use_layer_norm: bool = False,
              hidden_activation: str = "relu", last_activation: Optional[str] = None,
              dropout_ratio: float = 0.0, use_skip_connections: bool = False) -> nn.Module:
    """nn.Sequential of [Sequential(Linear, act)] * len(hidden) + [Sequential(Linear[, act])].

    The nesting (and therefore the ``state_dict`` keys ``0.0.weight``, ``1.0.weight`` ...) and
    the order in which nn.Linear layers draw their default init are those of the reference, so
    the same ``torch.manual_seed`` yields the same initial weights.
    """
    if use_batch_norm or use_layer_norm or dropout_ratio > 0 or use_skip_connections:
        raise NotImplementedError(
            "pearl_amd.mlp_block: batch/layer norm, dropout and skip connections are not part of "
            "the HIP learner path")
    dims = [input_dim] + list(hidden_dims or []) + [output_dim]
    layers = []
    for d_in, d_out in zip(dims[:-2], dims[1:-1]):
        layers.append(nn.Sequential(nn.Linear(d_in, d_out), _ACTIVATIONS[hidden_activation]()))
    last = [nn.Linear(dims[-2], dims[-1])]
    if last_activation is not None:
        last.append(_ACTIVATIONS[last_activation]())
    layers.append(nn.Sequential(*last))
    return nn.Sequential(*layers)


def xavier_init_weights(m: nn.Module) -> None:
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        m.bias.data.fill_(0.01)
