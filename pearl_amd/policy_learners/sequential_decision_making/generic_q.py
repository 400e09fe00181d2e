"""Generic TD(0) step for Q-network architectures the fused ``pa_dqn_*`` kernels do not cover.

``DeepQLearning`` / ``DoubleDQN`` / ``DeepSARSA`` route here when ``_Q`` is not "VanillaQValueNetwork
with two ReLU hidden layers of at most 256 units" (the shape of the fused MI355X path):

* ``VanillaQValueNetwork`` of any depth / width (common/utils.py:75-152 ``mlp_block``),
* ``VanillaQValueMultiHeadNetwork`` (q_value_networks.py:185-249): the trunk runs ONCE per state
  and Q(s, a) is a row-dot with the one-hot action — no ``(B, A, S + AD)`` expansion at all,
* ``DuelingQValueNetwork`` (q_value_networks.py:352-508): state tower -> value tower + advantage
  tower, ``Q = V + A - mean(A)``.

Same arithmetic as ``DeepTDLearning.learn_batch`` (deep_td_learning.py:269-360): conditional soft
target update, Q(s, a), next-state values of the subclass (max / double / SARSA), Bellman target,
``MSELoss`` backward, one ``AdamW(amsgrad)`` step, report ``mean |Q - target|``.  Every arithmetic
step is a libpearl_amd call on flat parameter views (``FlatMlp`` = the ``pa_mlp_*`` engine with fp32
MFMA GEMMs, ``qheads.hip`` for the row-local heads); there is no torch fallback.

The engine keeps ONE forward's activations per network, so all no-grad passes (targets) run before
the kept online forward.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional

import torch
from torch import Tensor, nn

from ... import _native as N
from ...neural_networks.sequential_decision_making.q_value_networks import (
    DuelingQValueNetwork, VanillaQValueMultiHeadNetwork, VanillaQValueNetwork)
from .flat_mlp import FlatMlp, layers_of


def plain_relu_mlp(model: nn.Module) -> bool:
    """True iff `model` is mlp_block's plain form: [Sequential(Linear, ReLU)] * k + [Sequential(Linear)]
    — what the pa_mlp engine computes.  Anything else (other activations, norms, dropout, residual
    blocks) must be refused loudly, not trained as if it were ReLU."""
    if not isinstance(model, nn.Sequential) or len(model) == 0:
        return False
    blocks = list(model)
    for blk in blocks[:-1]:
        if not (isinstance(blk, nn.Sequential) and len(blk) == 2 and isinstance(blk[0], nn.Linear)
                and type(blk[1]) is nn.ReLU):
            return False
    last = blocks[-1]
    return isinstance(last, nn.Sequential) and len(last) == 1 and isinstance(last[0], nn.Linear)


_ACT_KIND = {nn.ReLU: 0, nn.LeakyReLU: 1, nn.Tanh: 2, nn.Softplus: 3, nn.Sigmoid: 4}


def mlp_spec(model: nn.Module) -> Optional[Dict[str, Any]]:
    """What the pa_mlp engine needs to know about an mlp_block (common/utils.py:75-152), or None when
    the model is something else: hidden blocks ``Sequential(Linear[, LayerNorm][, Dropout], activation
    [, BatchNorm1d])`` that all have the same form, each possibly inside a ``ResidualWrapper``, then
    ``Sequential(Linear)`` (possibly wrapped too).  Activations: ReLU, LeakyReLU (slope 0.01), Tanh,
    Softplus (beta 1, threshold 20), Sigmoid, Identity.  Returns ``linears``, ``norms`` (the LayerNorm
    modules or None), ``bnorms`` (the BatchNorm1d modules or None), ``dropout`` (p of the hidden
    layers' nn.Dropout, 0.0 without), ``residual`` (bit l: layer l is wrapped), ``hidden_act``
    (pa_mlp_desc.hidden_act), ``identity`` (hidden layers have no activation), ``plain`` (Linear +
    ReLU only: the fused kernels' form)."""
    from ...neural_networks.common.residual_wrapper import ResidualWrapper
    if not isinstance(model, nn.Sequential) or len(model) == 0:
        return None
    blocks = list(model)
    linears, norms, bnorms, drops, kinds, drop_mods = [], [], [], [], [], []
    residual = 0
    for li, blk in enumerate(blocks[:-1]):
        if isinstance(blk, ResidualWrapper):
            residual |= 1 << li
            blk = blk.module
        if not (isinstance(blk, nn.Sequential) and 2 <= len(blk) <= 5 and isinstance(blk[0], nn.Linear)):
            return None
        rest = list(blk)[1:]
        ln = rest.pop(0) if rest and type(rest[0]) is nn.LayerNorm else None
        dr = rest.pop(0) if rest and type(rest[0]) is nn.Dropout else None
        if not rest:
            return None
        act = rest.pop(0)
        bn = rest.pop(0) if rest and type(rest[0]) is nn.BatchNorm1d else None
        if rest:
            return None
        d_out = blk[0].out_features
        if ln is not None and not (ln.elementwise_affine and ln.bias is not None
                                   and tuple(ln.normalized_shape) == (d_out,) and ln.eps == 1e-5):
            return None
        if bn is not None and not (bn.affine and bn.track_running_stats and bn.num_features == d_out
                                   and bn.eps == 1e-5 and bn.momentum == 0.1):
            return None
        if dr is not None and not (0.0 < dr.p < 1.0 and not dr.inplace):
            return None
        if type(act) is nn.Identity:
            kind = -1
        elif type(act) in _ACT_KIND:
            kind = _ACT_KIND[type(act)]
            if kind == 1 and act.negative_slope != 0.01:
                return None
            if kind == 3 and (act.beta != 1 or act.threshold != 20):
                return None
        else:
            return None
        linears.append(blk[0]); norms.append(ln); bnorms.append(bn); kinds.append(kind)
        drops.append(float(dr.p) if dr is not None else 0.0)
        drop_mods.append(dr)
    last = blocks[-1]
    if isinstance(last, ResidualWrapper):
        residual |= 1 << (len(blocks) - 1)
        last = last.module
    if not (isinstance(last, nn.Sequential) and len(last) == 1 and isinstance(last[0], nn.Linear)):
        return None
    linears.append(last[0])
    if len(set(kinds)) > 1 or len({n is None for n in norms}) > 1 or len({b is None for b in bnorms}) > 1 \
            or len(set(drops)) > 1:
        return None                     # mlp_block gives every hidden layer the same form
    for l in range(len(linears)):
        if (residual >> l) & 1 and linears[l].in_features != linears[l].out_features:
            return None
    has_ln = bool(norms) and norms[0] is not None
    has_bn = bool(bnorms) and bnorms[0] is not None
    p_drop = drops[0] if drops else 0.0
    kind = kinds[0] if kinds else 0
    return {"linears": linears, "norms": norms if has_ln else None, "bnorms": bnorms if has_bn else None,
            "dropout": p_drop, "dropout_modules": drop_mods if p_drop > 0 else None,
            "residual": residual, "hidden_act": max(kind, 0), "identity": kind == -1,
            "plain": (not has_ln) and (not has_bn) and p_drop == 0.0 and residual == 0 and kind == 0}


def plain_or_spec(model: nn.Module, what: str) -> Dict[str, Any]:
    spec = mlp_spec(model)
    if spec is None:
        raise NotImplementedError(
            f"pearl_amd: {what} is not an mlp_block the HIP engine computes (Linear [+ LayerNorm] [+ Dropout] "
            "+ relu / leaky_relu / tanh / softplus / sigmoid / linear [+ BatchNorm1d] hidden layers of one "
            "form, optionally with skip connections)")
    return spec


def flat_mlp_of(model: nn.Module, target_model: Optional[nn.Module], optimizer: Any, max_batch: int,
                what: str) -> FlatMlp:
    """FlatMlp over an mlp_block in any of the forms mlp_spec recognises (target_model: its copy)."""
    spec = plain_or_spec(model, what)
    tspec = plain_or_spec(target_model, what + " (target)") if target_model is not None else None
    L = len(spec["linears"])
    net = FlatMlp(layers_of(spec["linears"]), optimizer, max_batch,
                   target_layers=layers_of(tspec["linears"]) if tspec is not None else None,
                   identity_layers=((1 << (L - 1)) - 1) if spec["identity"] else 0,
                   norms=spec["norms"], target_norms=tspec["norms"] if tspec is not None else None,
                   hidden_act=spec["hidden_act"], bnorms=spec["bnorms"],
                   target_bnorms=tspec["bnorms"] if tspec is not None else None,
                   dropout=spec["dropout"], residual=spec["residual"])
    if spec["dropout_modules"]:
        net.dropout_modules = (spec["dropout_modules"], tspec["dropout_modules"] if tspec is not None else None)
    return net


def _f32(t: Tensor, dev: torch.device) -> Tensor:
    return t.to(device=dev, dtype=torch.float32).contiguous()


def _new(dev: torch.device, *shape: int) -> Tensor:
    return torch.empty(*shape, dtype=torch.float32, device=dev)


class _Ops:
    """What an architecture provides to the TD step."""
    nets: List[FlatMlp]

    def ensure(self, batch_hint: int) -> None:
        for m in self.nets:
            m.ensure(batch_hint)

    @property
    def device(self) -> torch.device:
        return self.nets[0].device

    def adam(self) -> None:
        for m in self.nets:
            m.adam(reduce="mean")

    def soft_update(self, tau: float) -> None:
        for m in self.nets:
            m.soft_update(tau)

    # q_taken(state, action_rep, curr_avail_rep) -> (B,), keeps activations; backward(dq)
    # q_all(state, rep[B, A, AD] | [A, AD], use_target) -> (B, A);  q_one(state, action_rep, use_target) -> (B,)


def _expand(state: Tensor, rep: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """x[b * A + i] = state[b] || rep[b, i]  (extend_state_feature.py:12-47 + cat)."""
    B, S = state.shape
    A, AD = int(rep.shape[-2]), int(rep.shape[-1])
    x = _new(state.device, B * A, S + AD) if out is None else out
    N.check(N.lib().pa_expand_state_actions(state.data_ptr(), state.stride(0), rep.data_ptr(),
                                            A * AD if rep.ndim == 3 else 0, B, A, S, AD,
                                            x.data_ptr(), N.stream_ptr(state.device)))
    return x


def _concat(left: Tensor, right: Tensor, out: Optional[Tensor] = None) -> Tensor:
    B = left.shape[0]
    x = _new(left.device, B, left.shape[1] + right.shape[1]) if out is None else out
    N.check(N.lib().pa_concat_cols(left.data_ptr(), left.stride(0), right.data_ptr(),
                                   right.stride(0), x.data_ptr(), B, left.shape[1], right.shape[1],
                                   N.stream_ptr(left.device)))
    return x


class VanillaOps(_Ops):
    def __init__(self, q: VanillaQValueNetwork, q_target: VanillaQValueNetwork, optimizer: Any,
                 max_batch: int) -> None:
        self.net = flat_mlp_of(q._model, q_target._model, optimizer, max_batch,
                               f"{type(q).__name__}._model")
        self.nets = [self.net]
        self._x: Optional[Tensor] = None

    def q_taken(self, state: Tensor, action: Tensor, curr_avail: Optional[Tensor]) -> Tensor:
        self._x = _concat(state, action)
        return self.net.forward(self._x, keep=True).view(-1)

    def backward(self, dq: Tensor) -> None:
        self.net.backward(self._x, dq, want_dw=True, defer=True)

    def cql_rows(self, state: Tensor, action: Tensor, rep: Tensor) -> Tensor:
        """Q of the B taken (state, action) rows followed by every (state, available action) row —
        B + B A rows in ONE kept forward: what the MSE term and compute_cql_loss's all-actions table
        (loss_fn_utils.py:52-58) need; `backward` then takes pa_cql_head's row gradients."""
        B, S = state.shape
        A, AD = int(rep.shape[-2]), int(rep.shape[-1])
        X = _new(state.device, B + B * A, S + AD)
        _concat(state, action, out=X[:B])
        _expand(state, rep, out=X[B:])
        self._x = X
        return self.net.forward(X, keep=True).view(-1)

    def q_all(self, state: Tensor, rep: Tensor, use_target: bool) -> Tensor:
        B, A = state.shape[0], int(rep.shape[-2])
        if self.net.supports_q_all(A):
            return self.net.q_all(state, rep, use_target=use_target).view(B, A)
        return self.net.forward(_expand(state, rep), use_target=use_target).view(B, A)

    def q_one(self, state: Tensor, action: Tensor, use_target: bool) -> Tensor:
        return self.net.forward(_concat(state, action), use_target=use_target).view(-1)


class MultiHeadOps(_Ops):
    def __init__(self, q: VanillaQValueMultiHeadNetwork, q_target: VanillaQValueMultiHeadNetwork,
                 optimizer: Any, max_batch: int) -> None:
        self.net = flat_mlp_of(q._model, q_target._model, optimizer, max_batch,
                               f"{type(q).__name__}._model")
        self.nets = [self.net]
        self.A = int(self.net.dims[-1])
        self._state: Optional[Tensor] = None
        self._action: Optional[Tensor] = None

    def _dot(self, f: Tensor, rep: Tensor) -> Tensor:
        B = f.shape[0]
        assert rep.shape == (B, self.A), (
            f"multi-head Q network: one-hot actions of width {self.A} expected, got {tuple(rep.shape)}")
        out = _new(f.device, B)
        N.check(N.lib().pa_rows_dot(f.data_ptr(), f.stride(0), rep.data_ptr(), rep.stride(0), B,
                                    self.A, out.data_ptr(), N.stream_ptr(f.device)))
        return out

    def q_taken(self, state: Tensor, action: Tensor, curr_avail: Optional[Tensor]) -> Tensor:
        self._state, self._action = state, action
        return self._dot(self.net.forward(state, keep=True), action)

    def backward(self, dq: Tensor) -> None:
        st, act = self._state, self._action
        B = st.shape[0]
        df = _new(st.device, B, self.A)
        N.check(N.lib().pa_rows_scale(dq.data_ptr(), act.data_ptr(), act.stride(0), B, self.A,
                                      df.data_ptr(), df.stride(0), N.stream_ptr(st.device)))
        rep, self._cql_rep = self._cql_rep, None
        if rep is not None:
            # dq = [B taken | B Q table]: the table's gradient reaches f through the transpose of its bmm
            Q = int(rep.shape[-2])
            assert dq.numel() == B + B * Q
            N.check(N.lib().pa_rows_bmm_t(dq[B:].data_ptr(), rep.data_ptr(),
                                          Q * self.A if rep.ndim == 3 else 0, B, Q, self.A, 1,
                                          df.data_ptr(), df.stride(0), N.stream_ptr(st.device)))
        self.net.backward(st, df, want_dw=True, defer=True)

    _cql_rep: Optional[Tensor] = None

    def cql_rows(self, state: Tensor, action: Tensor, rep: Tensor) -> Tensor:
        """Q of the B taken actions followed by the (B, Q) all-actions table, both from ONE kept
        forward f(s) (q_value_networks.py:211-238: the reference evaluates f twice — once for the
        MSE term, once inside compute_cql_loss — with the same values; the gradients add)."""
        B, Q = state.shape[0], int(rep.shape[-2])
        assert rep.shape[-1] == self.A
        self._state, self._action, self._cql_rep = state, action, rep
        f = self.net.forward(state, keep=True)
        rows = _new(state.device, B + B * Q)
        lib, s = N.lib(), N.stream_ptr(state.device)
        N.check(lib.pa_rows_dot(f.data_ptr(), f.stride(0), action.data_ptr(), action.stride(0), B,
                                self.A, rows[:B].data_ptr(), s))
        N.check(lib.pa_rows_bmm(rep.data_ptr(), Q * self.A if rep.ndim == 3 else 0, f.data_ptr(),
                                f.stride(0), B, Q, self.A, rows[B:].data_ptr(), s))
        return rows

    def q_all(self, state: Tensor, rep: Tensor, use_target: bool) -> Tensor:
        B, Q = state.shape[0], int(rep.shape[-2])
        assert rep.shape[-1] == self.A
        f = self.net.forward(state, use_target=use_target)
        out = _new(state.device, B, Q)
        N.check(N.lib().pa_rows_bmm(rep.data_ptr(), Q * self.A if rep.ndim == 3 else 0, f.data_ptr(),
                                    f.stride(0), B, Q, self.A, out.data_ptr(),
                                    N.stream_ptr(state.device)))
        return out

    def q_one(self, state: Tensor, action: Tensor, use_target: bool) -> Tensor:
        return self._dot(self.net.forward(state, use_target=use_target), action)


class DuelingOps(_Ops):
    def __init__(self, q: DuelingQValueNetwork, q_target: DuelingQValueNetwork, optimizer: Any,
                 max_batch: int) -> None:
        def mk(a: nn.Module, b: nn.Module) -> FlatMlp:
            return FlatMlp(layers_of(a.linear_layers()), optimizer, max_batch,
                           target_layers=layers_of(b.linear_layers()))
        self.state_net = mk(q.state_arch, q_target.state_arch)
        self.value_net = mk(q.value_arch, q_target.value_arch)
        self.adv_net = mk(q.advantage_arch, q_target.advantage_arch)
        self.nets = [self.state_net, self.value_net, self.adv_net]
        self.H = int(self.state_net.dims[-1])
        self._kept: Dict[str, Any] = {}

    def _combine(self, v: Tensor, adv_q: Tensor, Q: int, adv_mean: Optional[Tensor], M: int) -> Tensor:
        B = v.shape[0]
        out = _new(v.device, B, Q)
        N.check(N.lib().pa_dueling_q(v.data_ptr(), adv_q.data_ptr(), Q, N.ptr(adv_mean), M, B,
                                     out.data_ptr(), N.stream_ptr(v.device)))
        return out

    def q_taken(self, state: Tensor, action: Tensor, curr_avail: Optional[Tensor]) -> Tensor:
        """get_q_values(state, action (B, AD), curr_available_actions) (:424-506): the advantage
        tower sees B taken-action rows followed by B * M available-action rows."""
        B = state.shape[0]
        feats = self.state_net.forward(state, keep=True)                      # (B, H)
        v = self.value_net.forward(feats, keep=True).view(-1)                 # (B,)
        M = 0 if curr_avail is None else int(curr_avail.shape[-2])
        x_adv = _new(state.device, B * (1 + M), self.H + action.shape[1])
        _concat(feats, action, out=x_adv[:B])
        if M:
            _expand(feats, curr_avail, out=x_adv[B:])
        adv = self.adv_net.forward(x_adv, keep=True).view(-1)                 # (B + B M,)
        self._kept = dict(state=state, feats=feats, x_adv=x_adv, B=B, M=M, v=v, adv=adv)
        return self._combine(v, adv[:B], 1, adv[B:] if M else None, M).view(-1)

    def cql_rows(self, state: Tensor, action: Tensor, rep: Tensor) -> Tensor:
        """Q of the B taken actions (mean over the available actions' advantages) followed by the
        (B, M) all-actions table compute_cql_loss asks for — get_q_values(state, curr_available
        actions) with no separate available set, i.e. the mean over those same M rows
        (q_value_networks.py:474-479) — both from the taken-action forward's kept advantage rows."""
        q = self.q_taken(state, action, rep)
        k = self._kept
        B, M = k["B"], k["M"]
        assert M > 0
        rows = _new(state.device, B + B * M)
        rows[:B].copy_(q)
        N.check(N.lib().pa_dueling_q(k["v"].data_ptr(), k["adv"][B:].data_ptr(), M, None, 0, B,
                                     rows[B:].data_ptr(), N.stream_ptr(state.device)))
        k["cql"] = True
        return rows

    def backward(self, dq: Tensor) -> None:
        k = self._kept
        B, M, dev = k["B"], k["M"], dq.device
        lib, s = N.lib(), N.stream_ptr(dev)
        d_adv = _new(dev, B * (1 + M))
        if k.get("cql"):
            assert dq.numel() == B + B * M
            d_v = _new(dev, B)
            N.check(lib.pa_dueling_cql_grad(dq[:B].data_ptr(), dq[B:].data_ptr(), B, M,
                                            d_adv.data_ptr(), d_v.data_ptr(), s))
            dq = d_v
        else:
            N.check(lib.pa_dueling_grad(dq.data_ptr(), B, M, d_adv.data_ptr(), s))
        dx_adv = self.adv_net.backward(k["x_adv"], d_adv, want_dw=True, want_dx=True, defer=True)
        dfeat = self.value_net.backward(k["feats"], dq, want_dw=True, want_dx=True, defer=True)   # (B, H)
        N.check(lib.pa_dueling_feat_grad(dx_adv.data_ptr(), dx_adv.stride(0), B, M, self.H, 1,
                                         dfeat.data_ptr(), dfeat.stride(0), s))
        self.state_net.backward(k["state"], dfeat, want_dw=True, defer=True)

    def q_all(self, state: Tensor, rep: Tensor, use_target: bool) -> Tensor:
        """get_q_values(state, actions (B, Q, AD)) with no separate available-action set: the mean
        runs over the Q query actions themselves, padded ones included (:474-479)."""
        B, Q = state.shape[0], int(rep.shape[-2])
        feats = self.state_net.forward(state, use_target=use_target)
        v = self.value_net.forward(feats, use_target=use_target).view(-1)
        adv = self.adv_net.forward(_expand(feats, rep), use_target=use_target).view(-1)
        return self._combine(v, adv, Q, None, 0)

    def q_one(self, state: Tensor, action: Tensor, use_target: bool) -> Tensor:
        feats = self.state_net.forward(state, use_target=use_target)
        v = self.value_net.forward(feats, use_target=use_target).view(-1)
        adv = self.adv_net.forward(_concat(feats, action), use_target=use_target).view(-1)
        return self._combine(v, adv, 1, None, 0).view(-1)     # (v + a) - a, as the reference rounds it


def make_ops(q: nn.Module, q_target: nn.Module, optimizer: Any, max_batch: int) -> _Ops:
    if isinstance(q, DuelingQValueNetwork):
        towers = (q.state_arch._model, q.value_arch._model, q.advantage_arch._model)
        if not all(plain_relu_mlp(m) for m in towers):
            raise NotImplementedError("pearl_amd: DuelingQValueNetwork towers must be plain "
                                      "Linear + ReLU mlp_blocks")
        return DuelingOps(q, q_target, optimizer, max_batch)
    plain_or_spec(getattr(q, "_model", None), f"{type(q).__name__}._model")
    if isinstance(q, VanillaQValueMultiHeadNetwork):
        return MultiHeadOps(q, q_target, optimizer, max_batch)
    if isinstance(q, VanillaQValueNetwork):
        return VanillaOps(q, q_target, optimizer, max_batch)
    raise NotImplementedError(f"pearl_amd: no HIP path for Q network type {type(q).__name__}")


class GenericTd:
    """The TD(0) step over an `_Ops`; `rule`: 0 max (DeepQLearning), 1 double (DoubleDQN),
    2 SARSA (DeepSARSA) — pa_dqn_desc.double_q's encoding."""

    def __init__(self, ops: _Ops, rule: int, gamma: float, tau: float,
                 cql_alpha: Optional[float] = None) -> None:
        self.ops, self.rule, self.gamma, self.tau = ops, int(rule), float(gamma), float(tau)
        # is_conservative (deep_td_learning.py:323-327): loss += alpha * compute_cql_loss
        self.cql_alpha = None if cql_alpha is None else float(cql_alpha)
        if self.cql_alpha is not None and not hasattr(ops, "cql_rows"):
            raise NotImplementedError(
                f"pearl_amd: the CQL term has no kernel path for {type(ops).__name__[:-3]} networks")

    def targets(self, b: Dict[str, Any], want_next_v: bool = False):
        ops, dev = self.ops, b["state"].device
        B = b["state"].shape[0]
        lib, s = N.lib(), N.stream_ptr(dev)
        y = _new(dev, B)
        nv = _new(dev, B) if want_next_v else None
        if self.rule == 2:       # Q_target(s', committed next action) (deep_sarsa.py:59-97)
            v = ops.q_one(b["next_state"], b["next_action"], use_target=True)
            N.check(lib.pa_td_target(v.data_ptr(), 1, None, 0, None, 0, b["reward"].data_ptr(),
                                     b["terminated"].data_ptr(), self.gamma, B, 1, N.ptr(nv),
                                     y.data_ptr(), s))
            return nv, y
        nav, mask = b["next_avail"], b["next_mask"]
        A, AD = int(nav.shape[-2]), int(nav.shape[-1])
        if self.rule == 1:
            # DoubleDQN (double_dqn.py:29-57): a' = argmax over the available next actions of the
            # ONLINE network, valued by Q_target.get_q_values(s', a') on that ONE action — for a
            # dueling network that is V + A - A, not an entry of the all-actions table
            q_sel = ops.q_all(b["next_state"], nav, use_target=False)
            chosen = _new(dev, B, AD)
            N.check(lib.pa_argmax_rows(q_sel.data_ptr(), q_sel.stride(0), N.ptr(mask), A,
                                       nav.data_ptr(), A * AD if nav.ndim == 3 else 0, B, A, AD,
                                       None, chosen.data_ptr(), s))
            v = ops.q_one(b["next_state"], chosen, use_target=True)
            N.check(lib.pa_td_target(v.data_ptr(), 1, None, 0, None, 0, b["reward"].data_ptr(),
                                     b["terminated"].data_ptr(), self.gamma, B, 1, N.ptr(nv),
                                     y.data_ptr(), s))
            return nv, y
        q_val = ops.q_all(b["next_state"], nav, use_target=True)                     # (B, A)
        N.check(lib.pa_td_target(q_val.data_ptr(), q_val.stride(0), None, 0, N.ptr(mask), A,
                                 b["reward"].data_ptr(), b["terminated"].data_ptr(), self.gamma, B,
                                 A, N.ptr(nv), y.data_ptr(), s))
        return nv, y

    def q_values(self, b: Dict[str, Any]) -> Tensor:
        """Q(s, a) of the online network as forward() computes it (with curr_available_actions)."""
        return self.ops.q_taken(b["state"], b["action"], b["curr_avail"])

    def step(self, b: Dict[str, Any], do_target_update: bool, losses: Tensor) -> None:
        """One learn_batch; `losses` (2,) receives mean |Q - y| and the MSE."""
        ops, dev = self.ops, b["state"].device
        B = b["state"].shape[0]
        if do_target_update:
            ops.soft_update(self.tau)                  # before the forward (deep_td_learning.py:283-284)
        _, y = self.targets(b)
        if self.cql_alpha is not None:
            # mse(Q(s, a), y) + alpha (mean_b logsumexp_i Q(s_b, a_i) - mean of the reference's gather)
            # (loss_fn_utils.py:17-72) over B + B A rows of one kept forward
            rep = b["curr_avail"]
            assert rep is not None and rep.ndim == 3, "the CQL term needs curr_available_actions"
            A, AD = int(rep.shape[-2]), int(rep.shape[-1])
            q_rows = ops.cql_rows(b["state"], b["action"], rep)
            dq_rows = _new(dev, B + B * A)
            N.check(N.lib().pa_cql_head(q_rows.data_ptr(), y.data_ptr(), b["action"].data_ptr(),
                                        b["action"].stride(0), B, A, AD, self.cql_alpha,
                                        dq_rows.data_ptr(), losses.data_ptr(), N.stream_ptr(dev)))
            ops.backward(dq_rows)
            ops.adam()
            return
        q = ops.q_taken(b["state"], b["action"], b["curr_avail"])
        dq = _new(dev, B)
        N.check(N.lib().pa_td_head(q.data_ptr(), 1, y.data_ptr(), B, 2.0 / B, dq.data_ptr(),
                                   losses.data_ptr(), N.stream_ptr(dev)))
        ops.backward(dq)
        ops.adam()
