"""``NeuralLinearBandit`` (+ ``SquareCBExploration``) on HIP.

Mirror of pearl/policy_learners/contextual_bandits/neural_linear_bandit.py:44-330 and
.../exploration_modules/contextual_bandits/squarecb_exploration.py:22-115.

One ``learn_batch`` (:159-225) on the device:
  features = _nn_layers(x) ; mu = linear_layer_e2e(features)               pa_mlp_forward (the e2e layer
                                                                             is the bias-free last layer)
  loss = sum w (mu - r)^2 / sum w ; backward ; AdamW(amsgrad)                pa_weighted_mse_head, pa_mlp_*
  A += sym([1|f]^T ([1|f] w)) ; b += [1|f]^T (r w) ; sum_weight += sum w     pa_linreg_delta / _apply
      (one packed all-reduce of the deltas when torch.distributed is up — the reference's three
       all_reduce calls, linear_regression.py:207-210)
  inv_A = inv(A + lambda I) ; coefs = inv_A b                                 pa_linreg_solve
The regression operates on the features computed BEFORE the optimizer step, as in the reference.
``act`` / SquareCB sampling are act-time (outside the learner path) and stay torch expressions.
"""
from __future__ import annotations

import copy
import ctypes as C
import os

from typing import Any, Dict, List, Optional

import torch
import torch.distributed as dist
from torch import Tensor, optim
from torch.distributions import Categorical

from ... import _native as N
from ...action_representation_modules import ActionRepresentationModule
from ...neural_networks.contextual_bandit.linear_regression import NeuralLinearRegression
from ...replay_buffers.transition import TransitionBatch
from ..exploration import ExplorationModule
from ..policy_learner import PolicyLearner
from ..sequential_decision_making.flat_mlp import FlatMlp, layers_of


class SquareCBExploration(ExplorationModule):
    """p_a = 1 / (K + gamma * gap_a) for a != argmax, the arg-max takes the remainder
    (squarecb_exploration.py:59-115).  Rows are normalised one by one: identical to the reference
    for one context per call — the only case its whole-matrix ``complementary_sum`` (:90) and its
    (B,) - (B, A) broadcast (:84) yield a distribution for — and well-defined for batches.

    The probability table is ``pa_squarecb_probs`` (one launch for the whole batch of contexts)
    when the values live on a HIP device; the draw itself is the reference's — one
    ``Categorical(row).sample()`` per row on torch's global CPU generator — so a seeded run picks
    the reference's actions.  Values on the CPU (unit tests of host logic) take the same rule
    through torch."""

    def __init__(self, gamma: float, reward_lb: float = 0.0, reward_ub: float = 1.0,
                 clamp_values: bool = False, randomized_tiebreaking: bool = False) -> None:
        super().__init__()
        self._gamma, self.reward_lb, self.reward_ub = gamma, reward_lb, reward_ub
        self.clamp_values = clamp_values

    def clamp(self, values: Tensor) -> Tensor:
        return torch.clamp(values, min=self.reward_lb, max=self.reward_ub) if self.clamp_values else values

    def get_unnormalize_prob(self, empirical_gaps: Tensor, max_val: Any, action_num: int) -> Tensor:
        return torch.div(1.0, action_num + self._gamma * empirical_gaps)

    def probabilities(self, values: Tensor, n_actions: int) -> Tensor:
        """(B, n_actions) table the actions are drawn from (:71-92)."""
        values = values.reshape(-1, n_actions)
        B = values.shape[0]
        if values.is_cuda:
            v = values.to(torch.float32).contiguous()
            prob = torch.empty(B, n_actions, dtype=torch.float32, device=v.device)
            arg = torch.empty(B, dtype=torch.int32, device=v.device)
            N.check(N.lib().pa_squarecb_probs(v.data_ptr(), v.stride(0), B, n_actions,
                                              float(self._gamma), int(self.clamp_values),
                                              float(self.reward_lb), float(self.reward_ub),
                                              prob.data_ptr(), arg.data_ptr(), N.stream_ptr(v.device)))
            return prob
        values = self.clamp(values)
        max_val, max_indices = torch.max(values, dim=1)
        prob = self.get_unnormalize_prob(max_val.unsqueeze(1) - values, max_val, n_actions)
        for i in range(B):
            prob[i, max_indices[i]] = 0.0
            prob[i, max_indices[i]] = 1.0 - torch.sum(prob[i])
        return prob

    def act(self, subjective_state: Any, action_space: Any, values: Optional[Tensor] = None,
            representation: Any = None, exploit_action: Any = None,
            action_availability_mask: Any = None, **kwargs: Any) -> Tensor:
        assert values is not None
        prob = self.probabilities(values, action_space.n).cpu()
        selected = torch.zeros((prob.size(0),), dtype=torch.int)
        for i in range(prob.size(0)):
            selected[i] = Categorical(prob[i, :]).sample()
        return selected.squeeze(-1)

    def get_scores(self, subjective_state: Any, action_space: Any, values: Tensor,
                   exploit_action: Any = None, representation: Any = None) -> Tensor:
        return values.view(-1, action_space.n)           # (:117-126)


class UCBExploration(ExplorationModule):
    """score = value + alpha * sigma, the arg-max arm is played (ucb_exploration.py:27-95 on
    score_exploration_base.py:47-101 with NO_TIEBREAKING).  ``sigma`` comes from the regression head
    (``representation.calculate_sigma``: sqrt(x^T A^-1 x), pa_linreg_sigma on a HIP device); NaNs count
    as zero like the reference's.  An availability mask (1 = present) restricts the arg-max."""

    def __init__(self, alpha: float, randomized_tiebreaking: Any = None) -> None:
        super().__init__()
        if randomized_tiebreaking not in (None, False, 0) and \
                getattr(randomized_tiebreaking, "name", "") != "NO_TIEBREAKING":
            raise NotImplementedError("pearl_amd UCBExploration: randomized tie-breaking is not built")
        self._alpha = alpha

    def sigma(self, subjective_state: Tensor, representation: Any) -> Tensor:
        sigma = representation.calculate_sigma(subjective_state)
        return torch.where(torch.isnan(sigma), torch.zeros_like(sigma), sigma)

    def get_scores(self, subjective_state: Any, action_space: Any, values: Tensor,
                   exploit_action: Any = None, representation: Any = None) -> Tensor:
        n = int(action_space.n)
        values = values.view(-1, n)
        sigma = self.sigma(subjective_state, representation).view(values.shape)
        return (values + self._alpha * sigma).view(-1, n)

    def act(self, subjective_state: Any, action_space: Any, values: Optional[Tensor] = None,
            representation: Any = None, exploit_action: Any = None,
            action_availability_mask: Optional[Tensor] = None, **kwargs: Any) -> Tensor:
        assert values is not None and values.shape[-1] == action_space.n
        scores = self.get_scores(subjective_state, action_space, values, representation=representation)
        if action_availability_mask is not None:
            present = action_availability_mask.to(scores.device).bool().view(scores.shape)
            scores = scores.masked_fill(~present, float("-inf"))
        index = torch.argmax(scores, dim=1)
        return torch.nn.functional.embedding(index, action_space.actions_batch.to(index.device))

    def compare(self, other: Any) -> str:
        if not isinstance(other, UCBExploration):
            return "other is not an instance of UCBExploration"
        return "" if self._alpha == other._alpha else f"_alpha is different: {self._alpha} vs {other._alpha}"


_LOSS_KINDS = {"mse": 0, "mae": 1, "cross_entropy": 2}        # PA_LOSS_* (include/pearl_amd.h)
_OUT_ACTS = {"linear": 0, "sigmoid": 1}                        # PA_OUT_*


class NeuralLinearBandit(PolicyLearner):
    def __init__(self, feature_dim: int, hidden_dims: List[int],
                 exploration_module: Optional[ExplorationModule] = None,
                 action_representation_module: Optional[ActionRepresentationModule] = None,
                 training_rounds: int = 100, batch_size: int = 128, learning_rate: float = 0.0003,
                 l2_reg_lambda_linear: float = 1.0, gamma: float = 1.0,
                 apply_discounting_interval: float = 0.0, force_pinv: bool = False,
                 state_features_only: bool = True, loss_type: str = "mse",
                 output_activation_name: str = "linear", nn_e2e: bool = True,
                 separate_uncertainty: bool = False, **mlp_kwargs: Any) -> None:
        assert len(hidden_dims) >= 1
        # LossType (neural_networks/common/utils.py:60-72): "mse" | "mae" | "cross_entropy", a str or
        # an enum member with that value
        self.loss_type = str(getattr(loss_type, "value", loss_type)).lower()
        if self.loss_type not in _LOSS_KINDS:
            raise ValueError(f"{loss_type!r} is not a valid LossType")
        if self.loss_type == "cross_entropy":
            # the reference asserts this at learn time (:186); a linear output fails its [0, 1] check
            assert output_activation_name == "sigmoid", \
                "the cross-entropy loss needs output_activation_name='sigmoid'"
        super().__init__(training_rounds=training_rounds, batch_size=batch_size,
                         exploration_module=exploration_module, on_policy=False,
                         is_action_continuous=False,
                         action_representation_module=action_representation_module)
        self._feature_dim = feature_dim
        self.model = NeuralLinearRegression(
            feature_dim=feature_dim, hidden_dims=hidden_dims,
            l2_reg_lambda_linear=l2_reg_lambda_linear, gamma=gamma, force_pinv=force_pinv,
            output_activation_name=output_activation_name, nn_e2e=nn_e2e, **mlp_kwargs)
        self._optimizer: optim.Optimizer = optim.AdamW(self.model.parameters(), lr=learning_rate,
                                                       amsgrad=True)
        self._state_features_only = state_features_only
        self.apply_discounting_interval = apply_discounting_interval
        self.last_sum_weight_when_discounted = 0.0
        self.separate_uncertainty = separate_uncertainty
        self._out_act = _OUT_ACTS[output_activation_name]
        self._flat: Dict[str, Any] = {}

    # Native handles (`_flat`) and the side stream / events of the asynchronous solve
    # (`_solve_state`: torch.cuda.Event is neither picklable nor deep-copyable) are per-object
    # run-time state: a copy, a pickle or torch.save(agent) leaves them behind and the next
    # learn_batch rebuilds them (copy.deepcopy(learner) raised after the first step).
    def _portable_state(self, memo=None) -> Dict[str, Any]:
        self.model._linear_regression_layer.join_solve()
        out = {}
        for k, v in self.__dict__.items():
            if k == "_solve_state":
                continue
            out[k] = {} if k == "_flat" else (copy.deepcopy(v, memo) if memo is not None else v)
        return out

    def __deepcopy__(self, memo: dict) -> "NeuralLinearBandit":
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__dict__.update(self._portable_state(memo))
        return new

    def __getstate__(self) -> Dict[str, Any]:
        return self._portable_state()

    @property
    def feature_dim(self) -> int:
        return self._feature_dim

    @property
    def optimizer(self) -> optim.Optimizer:
        return self._optimizer

    def set_history_summarization_module(self, value: torch.nn.Module) -> None:
        if any(True for _ in value.parameters()):
            raise NotImplementedError("trainable history summarisation modules are not built")
        self._history_summarization_module = value

    def _net(self, batch_hint: int = 0) -> FlatMlp:
        if "head" in self._flat and "net" in self._flat and \
                self._flat["head"][0].device != self.model.linear_layer_e2e.weight.device:
            self._flat.pop("net").close()          # the model moved: the head tensors follow it
        if "net" not in self._flat:
            # the trunk in any of mlp_block's forms the engine computes (use_layer_norm /
            # hidden_activation of neural_linear_bandit.py:84-85, :110-111; generic_q.mlp_spec)
            from ..sequential_decision_making.generic_q import plain_or_spec
            spec = plain_or_spec(self.model._nn_layers._model, "NeuralLinearBandit's _nn_layers")
            layers = layers_of(spec["linears"])
            frozen = not self.model.nn_e2e
            if frozen:
                # nn_e2e=False (neural_linear_regression.py:100-105): mu = [1 | features] coefs, the
                # LinUCB regression's coefficients — buffers, not parameters: the loss reaches the trunk
                # THROUGH them, they get no gradient step, and `linear_layer_e2e` is left alone (its
                # .grad stays None in the reference, which AdamW skips).  To the engine that is a last
                # layer with weight coefs[1:] and bias coefs[0] that the optimizer does not own
                # (FlatMlp.frozen_last); `_load_head` rewrites it from the current coefs before a step.
                d = self.model._linear_regression_layer._feature_dim
                dev0 = spec["linears"][0].weight.device
                self._flat["head"] = (torch.zeros(1, d, device=dev0), torch.zeros(1, device=dev0))
                layers.append(([self._flat["head"][0]], [self._flat["head"][1]]))
            else:
                layers.append(([self.model.linear_layer_e2e.weight], []))     # bias-free last layer
            # the trunk's own output layer (index len-2) has no activation (last_activation=None) and
            # no LayerNorm; hidden_activation="linear" leaves every hidden layer without one
            n_hidden = len(layers) - 1
            ident = ((1 << n_hidden) - 1) if spec["identity"] else (1 << (n_hidden - 1))
            norms = (list(spec["norms"]) + [None]) if spec["norms"] else None
            # (round 6) batch norm / dropout sit on the trunk's hidden layers only; a skip connection
            # may also wrap the trunk's own output layer (index n_hidden - 1 here)
            bnorms = (list(spec["bnorms"]) + [None]) if spec["bnorms"] else None
            drops = [spec["dropout"]] * (n_hidden - 1) + [0.0]
            self._flat["net"] = FlatMlp(layers, self._optimizer, max(self._batch_size, 1),
                                        identity_layers=ident, norms=norms,
                                        hidden_act=spec["hidden_act"], frozen_last=frozen,
                                        bnorms=bnorms, dropout=drops, residual=spec["residual"])
            if spec["dropout_modules"]:
                self._flat["net"].dropout_modules = (list(spec["dropout_modules"]) + [None], None)
        if self.model.nn_e2e:
            return self._flat["net"].ensure(batch_hint)
        self._load_head()
        net = self._flat["net"].ensure(batch_hint)
        net.invalidate()      # (inside a learn() loop `ensure` does not look at version counters)
        return net

    def _load_head(self) -> None:
        """nn_e2e=False: the engine's last layer <- the regression's current coefficients (reading
        them joins the solve of the previous step: this mode's forward depends on it, as the
        reference's does).  In-place torch copies: `ensure` sees the version change and has the
        engine rebuild its derived copies of the weights."""
        w, b = self._flat["head"]
        coefs = self.model._linear_regression_layer._coefs
        with torch.no_grad():
            w.copy_(coefs[1:].to(w.device).view(1, -1))
            b.copy_(coefs[:1].to(b.device))
        net = self._flat.get("net")
        if net is not None:
            net.frozen_reloaded()

    def learn_batch(self, batch: TransitionBatch) -> Dict[str, Any]:
        net = self._net(len(batch))
        dev = net.device
        lib, s = N.lib(), N.stream_ptr(dev)
        x = batch.state if self._state_features_only else torch.cat([batch.state, batch.action], dim=1)
        x = x.to(dev, torch.float32).contiguous()
        B = x.shape[0]
        y = batch.reward.to(dev, torch.float32).reshape(B).contiguous()
        w = None if batch.weight is None else batch.weight.to(dev, torch.float32).reshape(B).contiguous()
        lr = self.model._linear_regression_layer
        d = lr._feature_dim
        D = d + 1
        kind, oact = _LOSS_KINDS[self.loss_type], self._out_act
        if kind == 2:
            # the reference's own checks (:181-186); predictions of a sigmoid are in [0, 1] already
            assert bool(torch.all(y >= 0)) and bool(torch.all(y <= 1)), \
                "cross-entropy needs labels in [0, 1]"
        for name in ("_A", "_b", "_sum_weight", "_inv_A", "_coefs"):
            buf = lr._buffers[name]       # (not getattr: reading _inv_A / _coefs joins the solve)
            if buf.device != dev or not buf.is_contiguous():
                lr.join_solve()
                setattr(lr, name, buf.to(dev).contiguous())
        # (nn_e2e=False steps through the general sequence: forward, loss head, backward, AdamW)
        fused = w is None and self.model.nn_e2e and self._rowstep_ok(net)
        if fused and not lr.uses_pinv and not (dist.is_available() and dist.is_initialized()) \
                and os.environ.get("PEARL_AMD_BANDIT_ONE_CALL", "1") != "0":
            return self._learn_batch_one_call(net, x, y, lr, kind, oact)
        dpred = torch.empty(B, dtype=torch.float32, device=dev)
        loss = torch.empty(1, dtype=torch.float32, device=dev)
        wsum = torch.empty(1, dtype=torch.float32, device=dev)
        pred = torch.empty(B, 1, dtype=torch.float32, device=dev)      # act(network output)
        if fused:
            # unit weights: forward (kept), the loss gradient and the backward pass in one launch
            # (mlp_rowstep.hpp); the weight gradients stay pending for net.adam()
            net.ready(B)
            N.check(lib.pa_wloss_rowstep(net.handle, x.data_ptr(), x.stride(0), B, y.data_ptr(),
                                         kind, oact, None, pred.data_ptr(), dpred.data_ptr(),
                                         loss.data_ptr(), s))
            net._pending_x = (x, dpred)
        else:
            z = net.forward(x, keep=True)                      # (B, 1); features stay in the engine
            N.check(lib.pa_weighted_loss_head(z.data_ptr(), z.stride(0), y.data_ptr(), N.ptr(w), B,
                                              kind, oact, pred.data_ptr(), dpred.data_ptr(),
                                              loss.data_ptr(), wsum.data_ptr(), s))
        # features of this forward (before the optimizer step) feed the regression: read in place
        # (the kept activation is not touched by the optimizer step; a copy launch otherwise)
        feats_ptr, feats_ld = C.c_void_p(), C.c_int32()
        N.check(lib.pa_mlp_activation(net.handle, len(net.layers) - 2, C.byref(feats_ptr),
                                      C.byref(feats_ld)))
        # all-zero weights: skip the optimizer (:171-175).  Data parallel: the decision is taken on
        # the GLOBAL weight sum, so every rank enters (or skips) the gradient all-reduce of
        # net.adam() together — a rank-local decision left the other ranks hanging in it.
        skip = False
        if w is not None:
            ws = wsum
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                ws = wsum.clone()
                dist.all_reduce(ws)
            skip = float(ws.item()) == 0.0
        if not skip:
            if not fused:
                net.backward(x, dpred, want_dw=True, defer=True)
            net.adam()
        # ---- LinUCB update on the detached features
        xs = torch.empty(B * ((D + 3) & ~3) + D, dtype=torch.float32, device=dev)     # rows of whole
        rs = torch.empty(B * ((D + 2) & ~1), dtype=torch.float32, device=dev)         # 16- / 8-byte vectors
        delta = torch.empty(D * (D + 1) + 1, dtype=torch.float32, device=dev)
        N.check(lib.pa_linreg_delta2(feats_ptr, feats_ld.value, y.data_ptr(), N.ptr(w), B, d,
                                    xs.data_ptr(), rs.data_ptr(), delta.data_ptr(), s))
        if dist.is_available() and dist.is_initialized():
            dist.all_reduce(delta)          # delta_A | delta_b | delta_sum_weight in ONE message
        # A, b, sum_weight updated in place; the same launch also writes the (A, b) snapshot the
        # asynchronous solve reads (two copy launches otherwise)
        snap = self._solve_slot(lr, dev)
        N.check(lib.pa_linreg_apply2(delta.data_ptr(), d, lr._A.data_ptr(), lr._b.data_ptr(),
                                     lr._sum_weight.data_ptr(),
                                     None if snap is None else snap[0].data_ptr(),
                                     None if snap is None else snap[1].data_ptr(), s))
        self._solve(lr, dev, snap)
        self._maybe_apply_discounting()
        p = pred.detach().reshape(B, -1)
        return {"label": y, "prediction": p, "weight": w if w is not None else torch.ones_like(y),
                "loss": loss[0], "mu_scores": p.mean()}

    def _rowstep_ok(self, net: FlatMlp) -> bool:
        """pa_rowstep_supported for this network's handle (a shape property: asked once per handle)."""
        memo = self._flat.get("rowstep_ok")
        key = net.handle.value
        if memo is None or memo[0] != key:
            memo = (key, bool(N.lib().pa_rowstep_supported(net.handle, None, 0)))
            self._flat["rowstep_ok"] = memo
        return memo[1]

    def _workspace(self, B: int, D: int, dev: torch.device, stream: int) -> Dict[str, Any]:
        """Scratch of the one-call step, kept between steps: everything in it is written and read by
        launches of ONE stream (the key), so the next step's launches are ordered behind this
        step's.  The unit weights of the report live here too (read-only by convention; the
        reference returns a fresh ``ones_like`` every step)."""
        ws = self._flat.get("ws")
        key = (B, D, dev.index, stream)
        if ws is None or ws["key"] != key:
            f32 = dict(dtype=torch.float32, device=dev)
            ws = {"key": key,
                  "dpred": torch.empty(B, **f32),
                  "xs": torch.empty(B * ((D + 3) & ~3) + D, **f32),      # rows of whole 16- /
                  "rs": torch.empty(B * ((D + 2) & ~1), **f32),          # 8-byte vectors
                  "delta": torch.empty(D * (D + 1) + 1, **f32),
                  "ones": torch.ones(B, **f32),
                  "args": N.BanditStepArgs()}
            a = ws["args"]
            a.B, a.d = B, D - 1
            a.d_pred, a.x_scratch, a.r_scratch, a.delta = (ws[k].data_ptr() for k in
                                                           ("dpred", "xs", "rs", "delta"))
            self._flat["ws"] = ws
        return ws

    def _learn_batch_one_call(self, net: FlatMlp, x: Tensor, y: Tensor, lr: Any, kind: int,
                              oact: int) -> Dict[str, Any]:
        """Unit weights, single process, a network the fused row step takes: the whole step is
        ``pa_bandit_step`` — four launches, the LinUCB moment update riding the weight-gradient /
        AdamW launch — and the solve on its side stream.  The report's tensors are views of one
        fresh allocation per step (predictions, loss, mean prediction)."""
        dev = x.device
        B = x.shape[0]
        s = N.stream_ptr(dev)
        ws = self._workspace(B, lr._feature_dim + 1, dev, s)
        out = torch.empty(B + 2, dtype=torch.float32, device=dev)
        step = net.next_adam_step()
        a = ws["args"]
        a.net = net.handle.value
        a.x, a.ldx, a.y = x.data_ptr(), x.stride(0), y.data_ptr()
        a.loss_kind, a.out_act, a.adam_step = kind, oact, step
        a.pred = out.data_ptr()
        a.scalars = a.pred + 4 * B
        bufs = lr._buffers
        a.A, a.b, a.sum_weight = bufs["_A"].data_ptr(), bufs["_b"].data_ptr(), bufs["_sum_weight"].data_ptr()
        if os.environ.get("PEARL_AMD_BANDIT_ASYNC_SOLVE", "1") == "0":
            a.A_snap = a.b_snap = a.side_stream = None
            N.check(N.lib().pa_bandit_step(C.byref(a), s))
            self._solve(lr, dev, None)
        else:
            # the solve of this step's (A, b) snapshot on the side stream, enqueued by the same call
            # (_solve / _solve_slot: the same protocol through torch's stream API)
            st = self._solve_state_for(lr, dev)
            i = st["slot"]
            st["slot"] = 1 - i
            st["cur"] = i
            snap, (h_ready, h_done) = st["snap"][i], st["evh"][i]
            a.A_snap, a.b_snap = snap[0].data_ptr(), snap[1].data_ptr()
            a.side_stream = st["side"][i].cuda_stream
            a.ev_slot_free = h_done if st["busy"][i] is not None else None
            a.ev_ready, a.ev_done = h_ready, h_done
            a.l2_reg_lambda = float(lr.l2_reg_lambda)
            a.work = st["work"][i].data_ptr()
            inv_i, coefs_i = st["out"][i]
            a.inv_A, a.coefs = inv_i.data_ptr(), coefs_i.data_ptr()
            a.singular = st["flag"].data_ptr() + 4 * i
            N.check(N.lib().pa_bandit_step(C.byref(a), s))
            done = st["ev"][i][1]
            st["busy"][i] = done
            lr.__dict__["_solve_done"] = (done, inv_i, coefs_i)
        net.stepped(step)
        self._maybe_apply_discounting()
        return {"label": y, "prediction": out[:B].view(B, 1), "weight": ws["ones"], "loss": out[B],
                "mu_scores": out[B + 1]}

    def _solve_state_for(self, lr: Any, dev: torch.device) -> Dict[str, Any]:
        D = lr._feature_dim + 1
        st = self.__dict__.get("_solve_state")
        if st is None or st["dev"] != dev or st["D"] != D:
            # two slots, each with its own stream, (A, b) snapshot, work area and RESULT pair: two
            # solves can be in flight (one serial workgroup each, 45 - 100 us beside the learner's
            # launches against an ~80 us step); the regression layer copies the latest result into
            # its buffers when something reads them (LinearRegression.join_solve)
            st = {"dev": dev, "D": D, "side": [torch.cuda.Stream(dev) for _ in range(2)], "slot": 0,
                  "out": [(torch.zeros(D, D, dtype=torch.float32, device=dev),
                           torch.zeros(D, dtype=torch.float32, device=dev)) for _ in range(2)],
                  "snap": [(torch.empty(D, D, dtype=torch.float32, device=dev),
                            torch.empty(D, dtype=torch.float32, device=dev)) for _ in range(2)],
                  "work": [torch.empty(D * 2 * D, dtype=torch.float64, device=dev) for _ in range(2)],
                  "flag": torch.zeros(2, dtype=torch.int32, device=dev),
                  # per slot: "snapshot written" and "solve finished", recorded anew every use
                  "ev": [(torch.cuda.Event(), torch.cuda.Event()) for _ in range(2)],
                  "busy": [None, None], "cur": None}
            # (an event has no handle before its first record: the raw handles go to pa_bandit_step)
            for pair, side in zip(st["ev"], st["side"]):
                for ev in pair:
                    ev.record(side)
            st["evh"] = [tuple(ev.cuda_event for ev in pair) for pair in st["ev"]]
            self.__dict__["_solve_state"] = st
        return st

    def _solve_slot(self, lr: Any, dev: torch.device):
        """The (A, b) snapshot pair the NEXT asynchronous solve reads, made safe to overwrite (the
        solve that last read it — two steps back — has finished as far as the learner stream is
        concerned); None when the solve runs in-stream (PEARL_AMD_BANDIT_ASYNC_SOLVE=0)."""
        if os.environ.get("PEARL_AMD_BANDIT_ASYNC_SOLVE", "1") == "0":
            return None
        st = self._solve_state_for(lr, dev)
        i = st["slot"]
        st["slot"] = 1 - i
        st["cur"] = i
        if st["busy"][i] is not None:
            torch.cuda.current_stream(dev).wait_event(st["busy"][i])
        return st["snap"][i]

    def _solve(self, lr: Any, dev: torch.device, snap=None) -> None:
        """inv(A + lambda I) and coefs = inv_A b (linear_regression.py:252-270 calculate_coefs) — OFF
        the learner's critical path: the fp64 Gauss-Jordan solve is ONE serial workgroup (45 us — 97 us in round 4 — of a
        240 us step) and nothing in the next learn_batch reads `_inv_A` / `_coefs`; they are read at
        act time.  So the step leaves a snapshot of (A, b) on the learner stream (written by the
        apply launch itself: `snap`, from _solve_slot) and the solve runs on the slot's side stream,
        writing the slot's result pair; readers of the two buffers join it and take the result
        (LinearRegression.join_solve: attribute access, state_dict).
        PEARL_AMD_BANDIT_ASYNC_SOLVE=0: in-stream as before."""
        d = lr._feature_dim
        st = self._solve_state_for(lr, dev)
        pinv = lr.uses_pinv     # force_pinv on an unregularised system: pa_linreg_pinv (same slots / streams)
        if os.environ.get("PEARL_AMD_BANDIT_ASYNC_SOLVE", "1") == "0":
            lr.join_solve()
            inv_A, coefs = lr._buffers["_inv_A"], lr._buffers["_coefs"]
            if pinv:
                N.check(N.lib().pa_linreg_pinv(lr._A.data_ptr(), lr._b.data_ptr(), float(lr.l2_reg_lambda),
                                               d, inv_A.data_ptr(), coefs.data_ptr(),
                                               st["flag"].data_ptr(), N.stream_ptr(dev)))
                return
            N.check(N.lib().pa_linreg_solve(lr._A.data_ptr(), lr._b.data_ptr(), float(lr.l2_reg_lambda),
                                            d, st["work"][0].data_ptr(), inv_A.data_ptr(),
                                            coefs.data_ptr(), st["flag"].data_ptr(), N.stream_ptr(dev)))
            return
        main = torch.cuda.current_stream(dev)
        if snap is None:
            # (a caller that changed A / b itself — discounting: take the snapshot with two copies)
            snap = self._solve_slot(lr, dev)
            snap[0].copy_(lr._A)
            snap[1].copy_(lr._b)
        i = st["cur"]
        A_s, b_s = snap
        ready, done = st["ev"][i]
        ready.record(main)
        side = st["side"][i]
        side.wait_event(ready)
        inv_i, coefs_i = st["out"][i]
        if pinv:
            N.check(N.lib().pa_linreg_pinv(A_s.data_ptr(), b_s.data_ptr(), float(lr.l2_reg_lambda), d,
                                           inv_i.data_ptr(), coefs_i.data_ptr(),
                                           st["flag"][i:].data_ptr(), side.cuda_stream))
        else:
            N.check(N.lib().pa_linreg_solve(A_s.data_ptr(), b_s.data_ptr(), float(lr.l2_reg_lambda), d,
                                            st["work"][i].data_ptr(), inv_i.data_ptr(), coefs_i.data_ptr(),
                                            st["flag"][i:].data_ptr(), side.cuda_stream))
        done.record(side)
        st["busy"][i] = done
        lr.__dict__["_solve_done"] = (done, inv_i, coefs_i)

    def _maybe_apply_discounting(self) -> None:
        lr = self.model._linear_regression_layer
        if self.apply_discounting_interval > 0:
            sw = float(lr._sum_weight.item())
            if sw - self.last_sum_weight_when_discounted >= self.apply_discounting_interval:
                if lr.gamma < 1:
                    lr._A *= lr.gamma
                    lr._b *= lr.gamma
                self._solve(lr, lr._A.device)
                self.last_sum_weight_when_discounted = sw

    def act(self, subjective_state: Tensor, available_action_space: Any,
            action_availability_mask: Optional[Tensor] = None, exploit: bool = False) -> Any:
        feats = self._concat_actions(subjective_state, available_action_space)
        with torch.no_grad():
            ret = self.model.forward_with_intermediate_values(feats)
        return self.exploration_module.act(
            subjective_state=ret["nn_output"], action_space=available_action_space,
            values=ret["pred_label_pre_activation"], action_availability_mask=action_availability_mask,
            representation=self.model._linear_regression_layer)

    def _concat_actions(self, state: Tensor, space: Any) -> Tensor:
        """utils/functional_utils/learning/action_utils.py concatenate_actions_to_state."""
        if state.ndim == 1:
            state = state.unsqueeze(0)
        B, n = state.shape[0], space.n
        exp = state.unsqueeze(1).repeat(1, n, 1)
        if self._state_features_only:
            return exp
        acts = self.action_representation_module(space.actions_batch.to(state)).unsqueeze(0).repeat(B, 1, 1)
        return torch.cat([exp, acts], dim=2)

    @torch.no_grad()
    def get_scores(self, subjective_state: Tensor, action_space_to_score: Any,
                   exploit: bool = False) -> Tensor:
        """Scores of every arm (neural_linear_bandit.py:260-311): features -> model ->
        exploration_module.get_scores(values) -> output activation (pre-activation values unless
        ``separate_uncertainty``).  Act-time torch, like ``act``."""
        assert not exploit, "exploit=True is not yet implemented for NeuralLinearBandit.get_scores"
        assert hasattr(self.exploration_module, "get_scores"), \
            "get_scores needs a score-based exploration module"
        feature = self._concat_actions(subjective_state, action_space_to_score)
        batch_size = feature.shape[0]
        ret = self.model.forward_with_intermediate_values(feature.reshape(-1, feature.shape[-1]))
        if self.separate_uncertainty is False:
            scores = self.exploration_module.get_scores(
                subjective_state=ret["nn_output"], values=ret["pred_label_pre_activation"],
                action_space=action_space_to_score,
                representation=self.model._linear_regression_layer)
            scores = self.model.output_activation(scores)
        else:
            scores = self.exploration_module.get_scores(
                subjective_state=ret["nn_output"], values=ret["pred_label"],
                action_space=action_space_to_score,
                representation=self.model._linear_regression_layer)
        return scores.reshape(batch_size, -1).squeeze(-1)

    def compare(self, other: PolicyLearner) -> str:
        diffs = [super().compare(other)]
        if not isinstance(other, NeuralLinearBandit):
            diffs.append("other is not an instance of NeuralLinearBandit")
        else:
            a, b = self.model.state_dict(), other.model.state_dict()
            for k in a:
                if k not in b or not torch.allclose(a[k].cpu(), b[k].cpu(), rtol=1e-5, atol=1e-8):
                    diffs.append(f"model is different: key {k} differs")
        return "\n".join(d for d in diffs if d)
