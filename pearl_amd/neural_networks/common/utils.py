"""Network construction helpers on the learner path
(pearl/neural_networks/common/utils.py:75-152 mlp_block, :201-205 xavier init).

The plain configuration the DQN / actor-critic configs use (Linear + ReLU hidden layers, optional
last activation) is what the fused kernels compute.  ``use_layer_norm`` and the other hidden
activations of ``ActivationType`` (utils.py:29-56: leaky_relu, tanh, softplus, sigmoid, linear) are
built too — such networks train through the generic ``pa_mlp`` engine layer by layer
(``mlp_norm_act.hpp``; ``generic_q.mlp_spec`` is what recognises them).  Since round 6 so are batch
norm, dropout and skip connections (utils.py:113-131, :142-150), in the same layer-by-layer path.
"""
from __future__ import annotations

import logging
from typing import List, Optional

import torch.nn as nn

from .residual_wrapper import ResidualWrapper

class _Softmax(nn.Softmax):
    def __init__(self) -> None:
        super().__init__(dim=-1)


_ACTIVATIONS = {"relu": nn.ReLU, "tanh": nn.Tanh, "sigmoid": nn.Sigmoid, "linear": nn.Identity,
                "softmax": _Softmax, "leaky_relu": nn.LeakyReLU, "softplus": nn.Softplus}
# hidden activations with HIP kernels (pa_mlp_desc.hidden_act; "linear" = FlatMlp.identity_layers)
HIDDEN_ACTIVATIONS = ("relu", "leaky_relu", "tanh", "softplus", "sigmoid", "linear")


def mlp_block(input_dim: int, hidden_dims: Optional[List[int]], output_dim: int = 1,
              use_batch_norm: bool = False, use_layer_norm: bool = False,
              hidden_activation: str = "relu", last_activation: Optional[str] = None,
              dropout_ratio: float = 0.0, use_skip_connections: bool = False) -> nn.Module:
    """nn.Sequential of [Sequential(Linear, act)] * len(hidden) + [Sequential(Linear[, act])].

    The nesting (and therefore the ``state_dict`` keys ``0.0.weight``, ``1.0.weight`` ...) and
    the order in which nn.Linear layers draw their default init are those of the reference, so
    the same ``torch.manual_seed`` yields the same initial weights.
    """
    if hidden_activation not in HIDDEN_ACTIVATIONS:
        raise NotImplementedError(
            f"pearl_amd.mlp_block: hidden_activation {hidden_activation!r} has no HIP kernel "
            f"(built: {', '.join(HIDDEN_ACTIVATIONS)})")
    dims = [input_dim] + list(hidden_dims or []) + [output_dim]
    layers = []
    for d_in, d_out in zip(dims[:-2], dims[1:-1]):
        single = [nn.Linear(d_in, d_out)]
        if use_layer_norm:
            single.append(nn.LayerNorm(d_out))          # (utils.py:110-113: between Linear and activation)
        if dropout_ratio > 0:
            single.append(nn.Dropout(p=dropout_ratio))  # (:114-116)
        single.append(_ACTIVATIONS[hidden_activation]())
        if use_batch_norm:
            single.append(nn.BatchNorm1d(d_out))        # (:119-121: AFTER the activation)
        block: nn.Module = nn.Sequential(*single)
        if use_skip_connections:                        # (:122-131)
            if d_in == d_out:
                block = ResidualWrapper(block)
            else:
                logging.warning("Skip connections are enabled, but layer in_dim (%d) != out_dim (%d). "
                                "Skip connection will not be added for this layer", d_in, d_out)
        layers.append(block)
    last = [nn.Linear(dims[-2], dims[-1])]
    if last_activation is not None:
        last.append(_ACTIVATIONS[last_activation]())
    last_block: nn.Module = nn.Sequential(*last)
    if use_skip_connections:                            # (:142-150)
        if dims[-2] == dims[-1]:
            last_block = ResidualWrapper(last_block)
        else:
            logging.warning("Skip connections are enabled, but layer in_dim (%d) != out_dim (%d). "
                            "Skip connection will not be added for this layer", dims[-2], dims[-1])
    layers.append(last_block)
    return nn.Sequential(*layers)


def linear_layers_of_plain(model: nn.Module, owner: str) -> List[nn.Linear]:
    """The nn.Linear modules of an mlp_block in its PLAIN form — hidden blocks ``Sequential(Linear,
    ReLU)`` — for the learners whose fused kernels hard-wire that form (the actor-critic family, the
    fused DQN step).  A network with LayerNorm or another hidden activation is refused here, loudly:
    it must never be trained as if it were ReLU.  (Such Q networks train through the generic TD
    engine, which reads the blocks itself: generic_q.mlp_spec.)"""
    blocks = list(model) if isinstance(model, nn.Sequential) else []
    for blk in blocks[:-1]:
        if not (isinstance(blk, nn.Sequential) and len(blk) == 2 and isinstance(blk[0], nn.Linear)
                and type(blk[1]) is nn.ReLU):
            raise NotImplementedError(
                f"pearl_amd: {owner} has hidden layers that are not Linear + ReLU (LayerNorm / other "
                "activations / batch norm / dropout / skip connections): this learner's fused HIP "
                "kernels compute the plain form only")
    if blocks and not (isinstance(blocks[-1], nn.Sequential) and isinstance(blocks[-1][0], nn.Linear)):
        raise NotImplementedError(
            f"pearl_amd: {owner}'s last layer is wrapped (skip connection): this learner's fused HIP "
            "kernels compute the plain form only")
    return [m for m in model.modules() if isinstance(m, nn.Linear)]


def xavier_init_weights(m: nn.Module) -> None:
    if isinstance(m, nn.Linear):
        nn.init.xavier_uniform_(m.weight)
        m.bias.data.fill_(0.01)
