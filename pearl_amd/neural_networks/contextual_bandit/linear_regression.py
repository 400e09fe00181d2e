"""LinUCB regression layer and the neural-linear model
(pearl/neural_networks/contextual_bandit/linear_regression.py:21-290,
 neural_linear_regression.py:25-157): containers of the buffers / parameters with the reference's
names (``_A``, ``_b``, ``_sum_weight``, ``_inv_A``, ``_coefs``; ``_nn_layers``,
``linear_layer_e2e``) so ``state_dict`` round-trips.  The learning step runs in libpearl_amd
(``pa_linreg_delta`` / ``_apply`` / ``_solve``); the torch expressions below serve act-time."""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch import Tensor

from ..common.utils import _ACTIVATIONS
from ..common.value_networks import VanillaValueNetwork


def _join_before_state_dict(module, prefix, keep_vars) -> None:
    module.join_solve()


class LinearRegression(nn.Module):
    def __init__(self, feature_dim: int, l2_reg_lambda: float = 1.0, gamma: float = 1.0,
                 force_pinv: bool = False) -> None:
        super().__init__()
        assert 0 < gamma <= 1, f"gamma should be in (0, 1]. Got gamma={gamma} instead"
        # force_pinv (linear_regression.py:138-157): torch.linalg.pinv of A + lambda I instead of inv.
        # Every such system goes through pa_linreg_pinv — eigenvalues by Jacobi rotations in fp64,
        # torch's default cut-off (eigenvalues at or below D * eps(float32) of the largest are dropped)
        # — whatever lambda is: with lambda > 0 the matrix is positive definite and its pseudo-inverse
        # is its inverse only while max eig(A) < lambda / (D eps), i.e. up to ~1e5 weighted samples
        # at lambda = 1; beyond that torch zeroes the directions near lambda where an inverse keeps
        # 1 / lambda, and inv_A / coefs / the UCB sigma would part from the reference.
        # The kernel holds the system in one workgroup's LDS: order <= 72.
        if force_pinv and feature_dim + 1 > 72:
            raise NotImplementedError(
                "pearl_amd LinearRegression: force_pinv is built for feature_dim <= 71 "
                f"(pa_linreg_pinv holds the system in one workgroup's LDS; got {feature_dim})")
        self._feature_dim = feature_dim
        self.gamma, self.l2_reg_lambda, self.force_pinv = gamma, l2_reg_lambda, force_pinv
        self.register_buffer("_A", torch.zeros(feature_dim + 1, feature_dim + 1))
        self.register_buffer("_b", torch.zeros(feature_dim + 1))
        self.register_buffer("_sum_weight", torch.zeros(1))
        self.register_buffer("_inv_A", torch.zeros(feature_dim + 1, feature_dim + 1))
        self.register_buffer("_coefs", torch.zeros(feature_dim + 1))
        # The learner refreshes `_inv_A` / `_coefs` on side streams (the fp64 solve is one serial
        # workgroup and nothing in the next learn_batch reads its result; two solves may be in
        # flight, each writing its own pair of result buffers): `_solve_done` is (event, inv_A,
        # coefs) of the latest refresh, and every read of the two buffers — attribute access,
        # state_dict — first makes torch's current stream wait for the event and copies the
        # results in.
        self.__dict__["_solve_done"] = None
        self.register_state_dict_pre_hook(_join_before_state_dict)
        # writers of the buffers join it too: a solve still in flight on the side stream
        # would otherwise finish AFTER load_state_dict / .to() and overwrite the restored
        # `_inv_A` / `_coefs` with values computed from the pre-restore A and b
        self.register_load_state_dict_pre_hook(self._join_before_load)

    def _join_before_load(self, *args, **kwargs) -> None:
        self.drain_solve()

    def drain_solve(self) -> None:
        """Host-side wait for the latest solve (before anything rewrites the buffers it writes)."""
        pend = self.__dict__.get("_solve_done")
        if pend is not None:
            pend[0].synchronize()
            self._take_solve(pend)

    def _apply(self, fn, *args, **kwargs):
        self.drain_solve()
        return super()._apply(fn, *args, **kwargs)

    def __getstate__(self):
        self.join_solve()                 # (an event is neither picklable nor deep-copyable)
        return self.__dict__

    def join_solve(self) -> None:
        pend = self.__dict__.get("_solve_done")
        if pend is not None:
            torch.cuda.current_stream(pend[1].device).wait_event(pend[0])
            self._take_solve(pend)

    def _take_solve(self, pend) -> None:
        """The finished (as far as the current stream is concerned) solve's results become the
        buffers' contents."""
        self.__dict__["_solve_done"] = None
        _, inv_A, coefs = pend
        with torch.no_grad():
            if self._buffers["_inv_A"].device == inv_A.device:
                self._buffers["_inv_A"].copy_(inv_A)
                self._buffers["_coefs"].copy_(coefs)
            else:       # (buffers moved since the solve was enqueued)
                self._buffers["_inv_A"] = inv_A.clone()
                self._buffers["_coefs"] = coefs.clone()

    def __getattr__(self, name: str):
        if name in ("_inv_A", "_coefs"):
            self.join_solve()
        return super().__getattr__(name)

    @property
    def uses_pinv(self) -> bool:
        """The pseudo-inverse kernel instead of the SPD solve (force_pinv, any lambda)."""
        return bool(self.force_pinv)

    @property
    def A(self) -> Tensor:
        return self._A + self.l2_reg_lambda * torch.eye(self._feature_dim + 1, device=self._A.device)

    @property
    def coefs(self) -> Tensor:
        return self._coefs

    @staticmethod
    def append_ones(x: Tensor) -> Tensor:
        ones = torch.ones_like(torch.select(x, dim=-1, index=0).unsqueeze(-1))
        return torch.cat((ones, x), dim=-1)

    def forward(self, x: Tensor) -> Tensor:
        batch_size = x.shape[0]
        x = self.append_ones(x.reshape(-1, x.shape[-1]))
        return torch.matmul(x, self.coefs.t()).reshape(batch_size, -1)

    def calculate_sigma(self, x: Tensor) -> Tensor:
        batch_size = x.shape[0]
        x = self.append_ones(x.reshape(-1, x.shape[-1]))
        return torch.sqrt((torch.matmul(x, self._inv_A) * x).sum(-1).unsqueeze(-1)).reshape(batch_size, -1)


class NeuralLinearRegression(nn.Module):
    def __init__(self, feature_dim: int, hidden_dims: List[int], l2_reg_lambda_linear: float = 1.0,
                 gamma: float = 1.0, force_pinv: bool = False,
                 output_activation_name: str = "linear", nn_e2e: bool = True, **mlp_kwargs) -> None:
        super().__init__()
        if output_activation_name not in ("linear", "sigmoid"):
            raise NotImplementedError("pearl_amd NeuralLinearRegression: the linear and sigmoid "
                                      "output activations have HIP kernels "
                                      f"(got {output_activation_name!r})")
        self._feature_dim = feature_dim
        self._nn_layers = VanillaValueNetwork(input_dim=feature_dim, hidden_dims=hidden_dims,
                                              output_dim=hidden_dims[-1], **mlp_kwargs)
        self._linear_regression_layer = LinearRegression(feature_dim=hidden_dims[-1],
                                                         l2_reg_lambda=l2_reg_lambda_linear,
                                                         gamma=gamma, force_pinv=force_pinv)
        self.output_activation: nn.Module = _ACTIVATIONS[output_activation_name]()
        self.linear_layer_e2e = nn.Linear(in_features=hidden_dims[-1], out_features=1, bias=False)
        self.nn_e2e = nn_e2e

    def forward_with_intermediate_values(self, x: Tensor) -> Dict[str, Tensor]:
        batch_size = x.shape[0]
        nn_output = self._nn_layers(x.reshape(-1, x.shape[-1]))
        # nn_e2e (neural_linear_regression.py:100-105, :140-147): mu from the end-to-end linear layer,
        # or from the LinUCB regression's coefficients on the same features
        out = self.linear_layer_e2e(nn_output) if self.nn_e2e else self._linear_regression_layer(nn_output)
        return {"pred_label_pre_activation": out.reshape(batch_size, -1),
                "pred_label": self.output_activation(out).reshape(batch_size, -1),
                "nn_output": nn_output}

    def forward(self, x: Tensor) -> Tensor:
        return self.forward_with_intermediate_values(x)["pred_label"]

    def calculate_sigma(self, x: Tensor) -> Tensor:
        batch_size = x.shape[0]
        return self._linear_regression_layer.calculate_sigma(
            self._nn_layers(x.reshape(-1, x.shape[-1]))).reshape(batch_size, -1)
