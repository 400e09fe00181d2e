"""ResidualWrapper (pearl/neural_networks/common/residual_wrapper.py:13-29): out = x + module(x).
mlp_block wraps a layer's block in it when ``use_skip_connections`` and d_in == d_out
(neural_networks/common/utils.py:122-131, :142-150).  The attribute is called ``module`` as in the
reference, so ``state_dict`` keys (``0.module.0.weight`` ...) are the reference's."""
import torch
import torch.nn as nn


class ResidualWrapper(nn.Module):
    def __init__(self, module: nn.Module) -> None:
        super().__init__()
        self.module = module

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return x + self.module(x)
