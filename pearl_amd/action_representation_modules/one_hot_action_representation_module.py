"""One-hot action representation
(pearl/action_representation_modules/one_hot_action_representation_module.py:18-70).

``forward`` is ``F.one_hot(x.long(), n).squeeze(-2).float()`` in the reference (:27-34).  For
tensors in HBM it runs the ``pa_one_hot`` HIP kernel; inside the fused ``learn()`` path the
expansion is folded into the gather kernel and this module is not called at all.
"""
from __future__ import annotations

import torch

from .. import _native as N
from .action_representation_module import ActionRepresentationModule


class OneHotActionTensorRepresentationModule(ActionRepresentationModule):
    def __init__(self, max_number_actions: int) -> None:
        super().__init__()
        self._max_number_actions = int(max_number_actions)

    @property
    def max_number_actions(self) -> int:
        return self._max_number_actions

    @property
    def representation_dim(self) -> int:
        return self._max_number_actions

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        n = self._max_number_actions
        if x.ndim == 1:
            x = x.unsqueeze(-1)
        # (.., 1) indices -> (.., n); a trailing dim != 1 keeps the reference's (.., d, n)
        lead = x.shape[:-1] if x.shape[-1] == 1 else x.shape
        if not x.is_cuda:
            # host-side act() plumbing only (not on the learner hot path)
            return torch.nn.functional.one_hot(x.long(), num_classes=n).squeeze(dim=-2).float()
        src = x.contiguous()
        out = torch.empty(tuple(lead) + (n,), dtype=torch.float32, device=x.device)
        N.check(N.lib().pa_one_hot(src.data_ptr(), N.pa_dtype_of(src.dtype), src.numel(), n,
                                   out.data_ptr(), N.stream_ptr(x.device)))
        return out

    def compare(self, other: ActionRepresentationModule) -> str:
        if not isinstance(other, OneHotActionTensorRepresentationModule):
            return "other is not an instance of OneHotActionTensorRepresentationModule"
        if self.max_number_actions != other.max_number_actions:
            return (f"max_number_actions is different: {self.max_number_actions} vs "
                    f"{other.max_number_actions}")
        return ""
