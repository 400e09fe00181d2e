"""``FlatMlp``: an nn.Module's Linear layers re-pointed into flat fp32 buffers + a ``pa_mlp`` handle.

The nn.Parameters stay the source of truth for ``state_dict`` / ``compare`` / checkpoints
(SURVEY.md §5), but their storage becomes slices of one flat buffer per role (parameters, target
parameters, gradient, AdamW ``exp_avg`` / ``exp_avg_sq`` / ``max_exp_avg_sq``) that libpearl_amd
reads and updates in place — the same ownership rule as ``DeepQLearning`` (deep_q_learning.py).

A "layer" is a list of weight parameters stacked along the output dimension plus the matching
list of biases: one nn.Linear normally, two for the Gaussian actor head where ``fc_mu`` and
``fc_std`` (actor_networks.py:516-517) form ONE last layer of 2A output rows.
"""
from __future__ import annotations

import ctypes as C
from typing import Any, Dict, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import nn, optim

from ... import _native as N

Layer = Tuple[List[nn.Parameter], List[nn.Parameter]]  # (weights stacked by rows, biases)


def layers_of(linears: Sequence[nn.Linear]) -> List[Layer]:
    return [([l.weight], [l.bias]) for l in linears]



def reduce_gradient_(flat_grad: torch.Tensor, reduce: str = "mean") -> torch.Tensor:
    """In-place data-parallel reduction of a flat gradient buffer; identity without a process
    group or with a single rank.  "mean": sum over ranks / world; "sum": sum over ranks."""
    assert reduce in ("mean", "sum"), reduce
    if not (dist.is_available() and dist.is_initialized()):
        return flat_grad
    world = dist.get_world_size()
    if world == 1:
        return flat_grad
    if reduce == "mean":
        flat_grad.mul_(1.0 / world)
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return flat_grad


class FlatMlp:
    def __init__(self, layers: List[Layer], optimizer: Optional[optim.Optimizer], max_batch: int,
                 target_layers: Optional[List[Layer]] = None, identity_layers: int = 0,
                 norms: Optional[Sequence[nn.LayerNorm]] = None,
                 target_norms: Optional[Sequence[nn.LayerNorm]] = None, hidden_act: int = 0,
                 frozen_last: bool = False, bnorms: Optional[Sequence[Optional[nn.BatchNorm1d]]] = None,
                 target_bnorms: Optional[Sequence[Optional[nn.BatchNorm1d]]] = None,
                 dropout: Any = 0.0, residual: int = 0) -> None:
        self.layers = layers
        # the last layer's tensors are NOT parameters of the caller's optimizer (the bandit's
        # nn_e2e=False head: the regression's coefficients, rewritten by the caller before every
        # step): they live in the flat buffers like any layer, but get no optimizer state entry and
        # no .grad, and whatever the engine's AdamW does to their slots is the caller's to overwrite
        self.frozen_last = bool(frozen_last)
        # the engine's AdamW launch also writes the frozen layer's slots (it knows no per-layer
        # ranges): after a step they hold optimizer-perturbed values until the owner reloads them
        # (`frozen_reloaded`).  `ensure` — the entry every use of the network goes through — refuses
        # a network whose frozen slots are stale instead of computing with them.
        self._frozen_stale = False
        self.identity_layers = int(identity_layers)   # bit l: hidden layer l has no ReLU
        self.target_layers = target_layers
        # mlp_block's other forms (pa_mlp_desc.hidden_act / layer_norm): the hidden layers' nn.LayerNorm
        # modules (one per hidden layer; their weight / bias join the flat buffers behind W / b) and
        # the activation kind (0 relu, 1 leaky_relu, 2 tanh, 3 softplus, 4 sigmoid)
        # (entries may be None: a hidden layer without a LayerNorm — the bandit trunk's output layer)
        self.norms = list(norms) if norms and any(n is not None for n in norms) else None
        self.target_norms = list(target_norms) if (target_norms and self.norms) else None
        self.hidden_act = int(hidden_act)
        assert self.norms is None or len(self.norms) == len(layers) - 1
        assert (self.target_norms is None) == (self.norms is None or target_layers is None)
        self.norm_mask = sum(1 << i for i, n in enumerate(self.norms or []) if n is not None)
        # round 6 — the remaining options of mlp_block (pa_mlp_desc.batch_norm / dropout / residual):
        # the hidden layers' nn.BatchNorm1d modules (weight / bias join the flat buffers behind the
        # LayerNorm block; running_mean / running_var / num_batches_tracked stay the modules' buffers
        # and are updated by the kernels), the dropout probability per hidden layer, and the mask of
        # layers wrapped in a ResidualWrapper
        n_hidden = len(layers) - 1
        self.bnorms = list(bnorms) if bnorms and any(b is not None for b in bnorms) else None
        self.target_bnorms = list(target_bnorms) if (target_bnorms and self.bnorms) else None
        assert self.bnorms is None or len(self.bnorms) == n_hidden
        assert (self.target_bnorms is None) == (self.bnorms is None or target_layers is None)
        self.bn_mask = sum(1 << i for i, b in enumerate(self.bnorms or []) if b is not None)
        ps = list(dropout) if isinstance(dropout, (list, tuple)) else [float(dropout)] * n_hidden
        assert len(ps) == n_hidden and all(0.0 <= p < 1.0 for p in ps)
        self.dropout = [float(p) for p in ps]
        self.drop_mask = sum(1 << i for i, p in enumerate(self.dropout) if p > 0.0)
        self.residual = int(residual)
        # dropout keep masks: callable(layer, use_target, B, d, device) -> [B, d] float tensor of 0 / 1
        # (None: torch's generator on the device); parity tests replay the reference's draws.
        # `dropout_modules` (the nn.Dropout modules, online then target, when the owner hands them
        # over): a module in eval() mode draws nothing, as in torch
        self.dropout_source: Optional[Any] = None
        self.dropout_modules: Optional[Tuple[Sequence[Any], Optional[Sequence[Any]]]] = None
        self._drop_live: Dict[Tuple[int, bool], torch.Tensor] = {}
        self.optimizer = optimizer
        self.max_batch = int(max_batch)
        self.dims = [int(layers[0][0][0].shape[1])] + [
            int(sum(w.shape[0] for w in ws)) for ws, _ in layers]
        self.handle: Optional[C.c_void_p] = None
        self.flat: Dict[str, torch.Tensor] = {}
        self._sig: Tuple = ()
        self._desc_key: Tuple = ()

    # ------------------------------------------------------------------ lifetime
    def close(self) -> None:
        h, self.handle = self.handle, None
        if h:
            N.lib().pa_mlp_destroy(h)

    def __del__(self) -> None:  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def __deepcopy__(self, memo: dict) -> None:
        return None  # learners rebuild their FlatMlp lazily

    # ------------------------------------------------------------------ binding
    def _params(self) -> List[nn.Parameter]:
        ps = [p for ws, bs in self.layers for p in (*ws, *bs)]
        if self.norms:
            ps += [p for ln in self.norms if ln is not None for p in (ln.weight, ln.bias)]
        if self.bnorms:
            ps += [p for bn in self.bnorms if bn is not None for p in (bn.weight, bn.bias)]
        return ps

    def _frozen(self) -> List[torch.Tensor]:
        return [p for p in (*self.layers[-1][0], *self.layers[-1][1])] if self.frozen_last else []

    def _optimized(self) -> List[nn.Parameter]:
        """The tensors the caller's optimizer owns (everything but a frozen last layer)."""
        fz = self._frozen()
        return [p for p in self._params() if not any(p is q for q in fz)]

    def _target_params(self) -> List[nn.Parameter]:
        if self.target_layers is None:
            return []
        ps = [p for ws, bs in self.target_layers for p in (*ws, *bs)]
        if self.target_norms:
            ps += [p for ln in self.target_norms if ln is not None for p in (ln.weight, ln.bias)]
        if self.target_bnorms:
            ps += [p for bn in self.target_bnorms if bn is not None for p in (bn.weight, bn.bias)]
        return ps

    @property
    def plain(self) -> bool:
        """Linear + ReLU (+ identity layers): the form the fused row kernels compute."""
        return (self.norms is None and self.hidden_act == 0 and self.bnorms is None
                and self.drop_mask == 0 and self.residual == 0)

    def _bn_buffers(self) -> List[torch.Tensor]:
        out = []
        for group in (self.bnorms, self.target_bnorms):
            for bn in (group or []):
                if bn is not None:
                    out += [bn.running_mean, bn.running_var, bn.num_batches_tracked]
        return out

    def _group(self) -> dict:
        if self.optimizer is None:
            return dict(lr=0.0, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False)
        first = self._params()[0]
        for g in self.optimizer.param_groups:
            if any(p is first for p in g["params"]):
                return g
        raise AssertionError("the optimizer does not own this network's parameters")

    def adam_steps(self) -> int:
        if self.optimizer is None:
            return 0
        for p in self._optimized():
            st = self.optimizer.state.get(p)
            if st and "step" in st:
                return int(float(st["step"]))
        return 0

    # While a learner's learn() loop owns the networks (FlatMlp.in_learn_loop), the torch-side step
    # counters — one 0-d tensor per parameter, 6 fill_() calls per network per step — are written
    # once at the end of the loop instead of after every step, and the per-step re-validation of the
    # parameter / optimizer-state aliasing is skipped (nothing else runs between the steps).
    in_learn_loop = False

    def _set_adam_steps(self, n: int) -> None:
        if FlatMlp.in_learn_loop:
            self._steps_dirty = n
            FlatMlp._dirty.add(self)
            return
        self._steps_dirty = None
        bank = getattr(self, "_step_bank", None)
        if bank is not None and self._steps_are_banked(bank):
            bank.fill_(float(n))      # every parameter's 0-d "step" tensor is a view of it
            return
        for p in self._optimized():
            st = self.optimizer.state.get(p)
            if st is not None and "step" in st:
                st["step"].fill_(float(n))

    def _steps_are_banked(self, bank: torch.Tensor) -> bool:
        """The optimizer's per-parameter "step" tensors are still the views of the bank this object
        handed out (optimizer.load_state_dict replaces them while the moment tensors may stay aliased:
        _signature() does not see that) — otherwise they are filled one by one."""
        lo, hi = bank.data_ptr(), bank.data_ptr() + bank.numel() * bank.element_size()
        for p in self._optimized():
            st = self.optimizer.state.get(p)
            if st is None or "step" not in st or not (lo <= st["step"].data_ptr() < hi):
                return False
        return True

    _dirty: set = set()

    @staticmethod
    def leave_learn_loop() -> None:
        """End of a learn() loop: write the deferred step counters."""
        FlatMlp.in_learn_loop = False
        pending, FlatMlp._dirty = FlatMlp._dirty, set()
        for m in pending:
            n = getattr(m, "_steps_dirty", None)
            if n is not None and m.optimizer is not None:
                m._set_adam_steps(n)

    def _signature(self) -> Tuple:
        sig = []
        frozen = self._frozen()
        for p in self._params():
            st = self.optimizer.state.get(p, {}) if (self.optimizer is not None and not any(
                p is q for q in frozen)) else {}
            sig.append((p.data_ptr(), tuple(st[k].data_ptr() if k in st else 0
                                            for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"))))
        sig.extend(p.data_ptr() for p in self._target_params())
        sig.extend(b.data_ptr() for b in self._bn_buffers())
        return tuple(sig)

    def frozen_reloaded(self) -> None:
        """The owner rewrote the frozen last layer's tensors since the last optimizer step."""
        self._frozen_stale = False

    def ensure(self, batch_hint: int = 0) -> "FlatMlp":
        for group in (self.bnorms, self.target_bnorms):
            for bn in (group or []):
                if bn is not None and not bn.training:
                    # the engine's BatchNorm1d is the TRAINING-mode one (statistics of the batch at hand:
                    # what the reference's learners run, they never call eval()); learning on a network
                    # that was switched to eval() would silently use other statistics than torch
                    raise NotImplementedError(
                        "pearl_amd FlatMlp: a BatchNorm1d of this network is in eval() mode; the HIP "
                        "learner step computes training-mode batch statistics (call .train() first)")
        if self.frozen_last and self._frozen_stale:
            raise RuntimeError(
                "pearl_amd FlatMlp: the frozen last layer was not reloaded after the last optimizer "
                "step (its slots hold AdamW-perturbed values): reload it and call frozen_reloaded()")
        if FlatMlp.in_learn_loop and self.handle is not None and self._sig \
                and batch_hint <= self.max_batch and getattr(self, "_loop_validated", False):
            return self
        self._loop_validated = FlatMlp.in_learn_loop
        if getattr(self, "_steps_dirty", None) is not None and not FlatMlp.in_learn_loop:
            self._set_adam_steps(self._steps_dirty)
        p0 = self._params()[0]
        if not p0.is_cuda:
            N.require_gpu()
            raise N.NativeError("pearl_amd: network parameters are on the CPU; move the learner to "
                                "a HIP device first — there is no CPU learner path")
        dev = p0.device
        g = self._group()
        max_b = max(self.max_batch, int(batch_hint), 1)
        key = (dev.index, tuple(self.dims), max_b, g["lr"], tuple(g["betas"]), g["eps"],
               g["weight_decay"], bool(g.get("amsgrad", False)), self.hidden_act, self.norm_mask,
               self.bn_mask, self.drop_mask, self.residual)
        if self.handle is not None and key != self._desc_key:
            torch.cuda.synchronize(dev)
            self.close()
        if self.handle is None:
            desc = N.MlpDesc(device=dev.index, n_layers=len(self.layers), max_batch=max_b,
                             lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"],
                             weight_decay=g["weight_decay"], amsgrad=int(bool(g.get("amsgrad", False))),
                             no_last_bias=int(len(self.layers[-1][1]) == 0),
                             identity_layers=self.identity_layers, hidden_act=self.hidden_act,
                             layer_norm=self.norm_mask, batch_norm=self.bn_mask, dropout=self.drop_mask,
                             residual=self.residual)
            for i, d in enumerate(self.dims):
                desc.dims[i] = d
            self._desc = desc
            h = C.c_void_p()
            N.check(N.lib().pa_mlp_create(C.byref(h), C.byref(desc)))
            self.handle, self._desc_key, self._sig = h, key, ()
            self.max_batch = max_b
        if self._sig and self._sig == self._signature():
            if getattr(self, "_steps_dirty", None) is None:
                self._steps = self.adam_steps()
            # same storage, but its CONTENTS may have been written through torch since the last
            # step (load_state_dict copies in place; every in-place torch op bumps the tensor's
            # version counter, our own kernels do not): derived copies are rebuilt on next use
            if self._flat_versions() != getattr(self, "_versions", None):
                N.check(N.lib().pa_mlp_invalidate(self.handle))
                self._versions = self._flat_versions()
            return self
        # ---- flatten
        P = int(N.lib().pa_mlp_param_count(C.byref(self._desc)))
        offs = (C.c_int64 * (2 * len(self.layers)))()
        N.check(N.lib().pa_mlp_param_offsets(C.byref(self._desc), offs))
        names = ["p", "grad"]
        if self.optimizer is not None:
            names += ["exp_avg", "exp_avg_sq", "max_exp_avg_sq"]
        if self.target_layers is not None:
            names.append("p_target")
        flat = {k: torch.zeros(P, dtype=torch.float32, device=dev) for k in names}
        steps = self.adam_steps()
        # torch keeps one 0-d host "step" tensor per parameter: views of ONE host tensor here, so the
        # bookkeeping after every native optimizer step is one fill_ (torch's own AdamW steps them
        # in place, which views take)
        n_params = len(self._params())
        self._step_bank = torch.full((n_params,), float(steps), dtype=torch.float32) \
            if self.optimizer is not None else None
        bank_i = 0
        # (parameter lists, flat offset, the matching target parameters) in _params() order: every
        # layer's weights and biases, then every LayerNorm's weight and bias
        groups = []
        for li, (ws, bs) in enumerate(self.layers):
            for kind, plist in ((0, ws), (1, bs)):
                groups.append((plist, int(offs[2 * li + kind]),
                               self.target_layers[li][kind] if self.target_layers is not None else None))
        if self.norms:
            noffs = (C.c_int64 * (2 * len(self.norms)))()
            N.check(N.lib().pa_mlp_norm_offsets(C.byref(self._desc), noffs))
            for li, ln in enumerate(self.norms):
                if ln is None:
                    continue
                tn = self.target_norms[li] if self.target_norms else None
                groups.append(([ln.weight], int(noffs[2 * li]), [tn.weight] if tn is not None else None))
                groups.append(([ln.bias], int(noffs[2 * li + 1]), [tn.bias] if tn is not None else None))
        if self.bnorms:
            boffs = (C.c_int64 * (2 * len(self.bnorms)))()
            N.check(N.lib().pa_mlp_bn_offsets(C.byref(self._desc), boffs))
            for li, bn in enumerate(self.bnorms):
                if bn is None:
                    continue
                tb = self.target_bnorms[li] if self.target_bnorms else None
                groups.append(([bn.weight], int(boffs[2 * li]), [tb.weight] if tb is not None else None))
                groups.append(([bn.bias], int(boffs[2 * li + 1]), [tb.bias] if tb is not None else None))
        frozen = self._frozen()
        with torch.no_grad():
            for plist, o, tl in groups:
                if True:
                    for pi, p in enumerate(plist):
                        n = p.numel()
                        sl = slice(o, o + n)
                        flat["p"][sl].copy_(p.data.reshape(-1).to(dev, torch.float32))
                        if any(p is q for q in frozen):
                            p.data = flat["p"][sl].view(p.shape)
                            o += n
                            bank_i += 1
                            continue
                        st = (self.optimizer.state.get(p) or {}) if self.optimizer is not None else {}
                        for k in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
                            if k in st and k in flat:
                                flat[k][sl].copy_(st[k].reshape(-1).to(dev, torch.float32))
                        p.data = flat["p"][sl].view(p.shape)
                        p.grad = flat["grad"][sl].view(p.shape)
                        if self.optimizer is not None:
                            self.optimizer.state[p] = {
                                "step": self._step_bank[bank_i],
                                "exp_avg": flat["exp_avg"][sl].view(p.shape),
                                "exp_avg_sq": flat["exp_avg_sq"][sl].view(p.shape),
                            }
                            if g.get("amsgrad", False):     # (torch keeps it only then)
                                self.optimizer.state[p]["max_exp_avg_sq"] = \
                                    flat["max_exp_avg_sq"][sl].view(p.shape)
                        if tl is not None:
                            pt = tl[pi]
                            flat["p_target"][sl].copy_(pt.data.reshape(-1).to(dev, torch.float32))
                            pt.data = flat["p_target"][sl].view(pt.shape)
                        o += n
                        bank_i += 1
        bufs = N.MlpBuffers(p=flat["p"].data_ptr(), p_target=N.ptr(flat.get("p_target")),
                            grad=flat["grad"].data_ptr(), exp_avg=N.ptr(flat.get("exp_avg")),
                            exp_avg_sq=N.ptr(flat.get("exp_avg_sq")),
                            max_exp_avg_sq=N.ptr(flat.get("max_exp_avg_sq")))
        N.check(N.lib().pa_mlp_bind(self.handle, C.byref(bufs)))
        # the BatchNorm1d modules' running statistics: updated in place by every forward
        for which, group in ((0, self.bnorms), (1, self.target_bnorms)):
            for li, bn in enumerate(group or []):
                if bn is None:
                    continue
                for b in (bn.running_mean, bn.running_var, bn.num_batches_tracked):
                    assert b.device == dev, "BatchNorm1d buffers must live on the network's device"
                assert bn.running_mean.dtype == torch.float32 and bn.num_batches_tracked.dtype == torch.int64
                N.check(N.lib().pa_mlp_bind_batch_norm(
                    self.handle, which, li, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
                    bn.num_batches_tracked.data_ptr()))
        self.flat = flat
        self._sig = self._signature()
        self._steps = steps
        self._versions = self._flat_versions()
        self._pending_x = None
        return self

    def invalidate(self) -> None:
        """The parameters were (or may have been) written from outside the library: its derived
        copies (MFMA fragment-major weights) are rebuilt by the next launch that needs them."""
        if self.handle is not None:
            N.check(N.lib().pa_mlp_invalidate(self.handle))

    def _flat_versions(self) -> Tuple:
        """Version counters of the PARAMETER tensors (online and target).  They alias the flat
        buffers' storage but keep their own counters (``param.data = view`` does not share them), and
        they are what ``load_state_dict`` / torch optimizers write through.  Writes through a
        ``.data`` alias are not versioned by torch at all: nothing can see those."""
        vs = [p._version for p in self._params()]
        vs.extend(p._version for p in self._target_params())
        return tuple(vs)

    def ready(self, batch: int = 0) -> "FlatMlp":
        """The per-launch check: a bound handle large enough for `batch`.  Whether the torch
        parameters / optimizer state still alias the flat buffers is re-validated by `ensure`,
        which learners call once at the top of every learn_batch."""
        if self.handle is None or not self._sig or batch > self.max_batch:
            return self.ensure(batch)
        return self

    # ------------------------------------------------------------------ ops (all enqueue on torch's stream)
    @property
    def device(self) -> torch.device:
        return self._params()[0].device

    def forward(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, use_target: bool = False,
                keep: bool = False) -> torch.Tensor:
        assert x.dtype == torch.float32 and x.is_cuda and x.stride(-1) == 1 and x.ndim == 2
        B = int(x.shape[0])
        self.ready(B)
        if out is None:
            out = torch.empty(B, self.dims[-1], dtype=torch.float32, device=x.device)
        if self.drop_mask:
            self._set_dropout_masks(B, bool(use_target), bool(keep), x.device)
        N.check(N.lib().pa_mlp_forward(self.handle, int(use_target), x.data_ptr(), x.stride(0), B,
                                       out.data_ptr(), out.stride(0), int(keep),
                                       N.stream_ptr(x.device)))
        return out

    def _set_dropout_masks(self, B: int, use_target: bool, keep: bool, dev: torch.device) -> None:
        """nn.Dropout in training mode (utils.py:114-116; the reference never switches its networks
        to eval, so every forward — online, target, act-time — draws): one keep mask per dropout layer
        for this forward, scaled by 1 / (1 - p), handed to the engine; the kept forward's masks stay
        alive until its backward."""
        for li, p in enumerate(self.dropout):
            if p <= 0.0:
                continue
            mods = self.dropout_modules[1 if use_target else 0] if self.dropout_modules else None
            if mods is not None and mods[li] is not None and not mods[li].training:
                N.check(N.lib().pa_mlp_set_dropout(self.handle, li, None, 0))     # eval(): identity
                continue
            d = self.dims[li + 1]
            if self.dropout_source is not None:
                keep01 = self.dropout_source(li, use_target, B, d, dev).to(dev, torch.float32)
            else:
                keep01 = torch.empty(B, d, dtype=torch.float32, device=dev).bernoulli_(1.0 - p)
            mask = (keep01 * (1.0 / (1.0 - p))).contiguous()
            assert tuple(mask.shape) == (B, d)
            self._drop_live[(li, keep and not use_target)] = mask      # (alive past the launch)
            N.check(N.lib().pa_mlp_set_dropout(self.handle, li, mask.data_ptr(), mask.stride(0)))

    def _dw_mode(self, want_dw: bool, defer: bool, x: torch.Tensor, d_out: torch.Tensor) -> int:
        """0: no weight gradients; 1: now; 2: deferred to adam(), where ONE launch per three layers
        does dW, AdamW and the refresh of the row-pass kernels' packed weights (single process
        only: a data-parallel step all-reduces the gradient between the two)."""
        if not want_dw:
            return 0
        if defer:
            self._pending_x = (x, d_out)     # both operands must outlive the deferred launch
            return 2
        return 1

    def backward(self, x: torch.Tensor, d_out: torch.Tensor, want_dw: bool = True,
                 want_dx: bool = False, defer: bool = False) -> Optional[torch.Tensor]:
        B = int(x.shape[0])
        d_x = torch.empty(B, self.dims[0], dtype=torch.float32, device=x.device) if want_dx else None
        N.check(N.lib().pa_mlp_backward(self.handle, x.data_ptr(), x.stride(0), B, d_out.data_ptr(),
                                        d_out.stride(0) if d_out.ndim == 2 else 1,
                                        self._dw_mode(want_dw, defer, x, d_out),
                                        N.ptr(d_x), self.dims[0], N.stream_ptr(x.device)))
        return d_x

    def supports_q_all(self, n_actions: int) -> bool:
        """pa_mlp_q_all's shapes: a [S + AD, H1 <= 256, H2 <= 256, 1] ReLU critic, <= 64 actions."""
        return (len(self.dims) == 4 and self.dims[3] == 1 and max(self.dims[1:3]) <= 256
                and self.identity_layers == 0 and self.plain and len(self.layers[-1][1]) > 0
                and n_actions <= 64)

    def q_all(self, state: torch.Tensor, rep: torch.Tensor, use_target: bool = False) -> torch.Tensor:
        """Q(s_b, a_i) for every action of every state's action set: (B * A,), row b * A + i —
        TwinCritic.get_q_values on an action set (twin_critic.py:75-91) through the fused
        all-actions kernel (no (B A, S + AD) input, no hidden activations in HBM)."""
        assert state.dtype == torch.float32 and state.is_cuda and state.stride(-1) == 1
        assert rep.dtype == torch.float32 and rep.is_contiguous() and rep.ndim in (2, 3)
        B = int(state.shape[0])
        A, AD = int(rep.shape[-2]), int(rep.shape[-1])
        self.ready(B)
        out = torch.empty(B * A, dtype=torch.float32, device=state.device)
        N.check(N.lib().pa_mlp_q_all(self.handle, int(use_target), state.data_ptr(), state.stride(0),
                                     rep.data_ptr(), A * AD if rep.ndim == 3 else 0, B, A, AD,
                                     out.data_ptr(), N.stream_ptr(state.device)))
        return out

    @staticmethod
    def q_all_pair(m1: "FlatMlp", m2: "FlatMlp", state: torch.Tensor, rep: torch.Tensor,
                   use_target: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """``q_all`` of both critics of a twin in one call (pa_mlp_q_all2: the repack and the
        first-layer GEMM of the two networks share a launch each)."""
        assert state.dtype == torch.float32 and state.is_cuda and state.stride(-1) == 1
        assert rep.dtype == torch.float32 and rep.is_contiguous() and rep.ndim in (2, 3)
        B = int(state.shape[0])
        A, AD = int(rep.shape[-2]), int(rep.shape[-1])
        m1.ready(B)
        m2.ready(B)
        out = torch.empty(2, B * A, dtype=torch.float32, device=state.device)
        N.check(N.lib().pa_mlp_q_all2(m1.handle, m2.handle, int(use_target), state.data_ptr(),
                                      state.stride(0), rep.data_ptr(), A * AD if rep.ndim == 3 else 0,
                                      B, A, AD, out[0].data_ptr(), out[1].data_ptr(),
                                      N.stream_ptr(state.device)))
        return out[0], out[1]

    @staticmethod
    def forward_pair(m1: "FlatMlp", m2: "FlatMlp", x: torch.Tensor, use_target: bool = False,
                     keep: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
        """Both networks of a twin (same shape) on the same input, one launch per layer."""
        assert x.dtype == torch.float32 and x.is_cuda and x.stride(-1) == 1 and x.ndim == 2
        B = int(x.shape[0])
        m1.ready(B)
        m2.ready(B)
        o1 = torch.empty(B, m1.dims[-1], dtype=torch.float32, device=x.device)
        o2 = torch.empty(B, m2.dims[-1], dtype=torch.float32, device=x.device)
        N.check(N.lib().pa_mlp_forward2(m1.handle, m2.handle, int(use_target), x.data_ptr(),
                                        x.stride(0), B, o1.data_ptr(), o1.stride(0), o2.data_ptr(),
                                        o2.stride(0), int(keep), N.stream_ptr(x.device)))
        return o1, o2

    @staticmethod
    def rowstep_supported(m1: "FlatMlp", m2: "FlatMlp", ppo_actions: int = 0) -> bool:
        """The two networks qualify for the fused forward + head + backward launch
        (pa_ppo_rowstep / pa_mse_rowstep2; PEARL_AMD_ROWSTEP=0: never)."""
        return bool(N.lib().pa_rowstep_supported(m1.handle, m2.handle, int(ppo_actions)))

    @staticmethod
    def ppo_rowstep(actor: "FlatMlp", critic: "FlatMlp", x: torch.Tensor, arep: torch.Tensor,
                    p_old: torch.Tensor, gae: torch.Tensor, epsilon: float, entropy_scale: float,
                    value_target: torch.Tensor, value_grad_scale: float) -> torch.Tensor:
        """forward_pair(keep) -> pa_ppo_heads -> backward_pair(defer) as one launch; returns the
        (2,) losses (actor, critic).  The weight gradients are pending: adam_pair next."""
        B = int(x.shape[0])
        actor.ready(B)
        critic.ready(B)
        dev = x.device
        d_logits = torch.empty(B, actor.dims[-1], dtype=torch.float32, device=dev)
        dv = torch.empty(B, dtype=torch.float32, device=dev)
        losses = torch.empty(2, dtype=torch.float32, device=dev)
        N.check(N.lib().pa_ppo_rowstep(
            actor.handle, critic.handle, x.data_ptr(), x.stride(0), B, arep.data_ptr(),
            arep.stride(0), p_old.data_ptr(), gae.data_ptr(), float(epsilon), float(entropy_scale),
            value_target.data_ptr(), float(value_grad_scale), None, 0, None, 0, d_logits.data_ptr(),
            d_logits.stride(0), dv.data_ptr(), losses.data_ptr(), N.stream_ptr(dev)))
        actor._pending_x = (x, d_logits)     # operands of the deferred weight-gradient launch
        critic._pending_x = (x, dv)
        return losses

    @staticmethod
    def mse_rowstep_pair(c1: "FlatMlp", c2: "FlatMlp", x: torch.Tensor, target: torch.Tensor,
                         grad_scale: float, loss_scale: float,
                         loss_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """forward_pair(keep) -> two MSE heads -> backward_pair(defer) of twin critics as one
        launch; returns the (1,) loss ``loss_scale (mse_1 + mse_2)``."""
        B = int(x.shape[0])
        c1.ready(B)
        c2.ready(B)
        dev = x.device
        dq = torch.empty(2, B, dtype=torch.float32, device=dev)
        loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float32, device=dev)
        N.check(N.lib().pa_mse_rowstep2(c1.handle, c2.handle, x.data_ptr(), x.stride(0), B,
                                        target.data_ptr(), float(grad_scale), float(loss_scale), None,
                                        None, dq[0].data_ptr(), dq[1].data_ptr(), loss.data_ptr(),
                                        N.stream_ptr(dev)))
        c1._pending_x = (x, dq)
        c2._pending_x = (x, dq)
        return loss

    @staticmethod
    def backward_pair(m1: "FlatMlp", m2: "FlatMlp", x: torch.Tensor, d1: torch.Tensor,
                      d2: torch.Tensor, want_dw: bool = True, want_dx: bool = False,
                      defer: bool = False):
        B = int(x.shape[0])
        mode = m1._dw_mode(want_dw, defer, x, d1)
        m2._dw_mode(want_dw, defer, x, d2)
        dx1 = dx2 = None
        if want_dx:
            dx1 = torch.empty(B, m1.dims[0], dtype=torch.float32, device=x.device)
            dx2 = torch.empty(B, m2.dims[0], dtype=torch.float32, device=x.device)
        N.check(N.lib().pa_mlp_backward2(
            m1.handle, m2.handle, x.data_ptr(), x.stride(0), B, d1.data_ptr(),
            d1.stride(0) if d1.ndim == 2 else 1, d2.data_ptr(), d2.stride(0) if d2.ndim == 2 else 1,
            mode, N.ptr(dx1), N.ptr(dx2), m1.dims[0], N.stream_ptr(x.device)))
        return dx1, dx2

    def adam(self, reduce: str = "mean") -> None:
        """AdamW(amsgrad) step on the flat gradient buffer.

        Data parallelism (not in the reference): when ``torch.distributed`` is
        initialised every rank holds the gradient of ITS minibatch and parameters stay replicated,
        so the flat gradient buffer is all-reduced (RCCL over xGMI on GPUs — ONE message per
        network: PPO 135 696 + 131 841 floats, SAC actor 86 544 + twin critic 169 474) before the
        optimizer step.  ``reduce`` says how the loss aggregates over the global batch: "mean"
        (MSE / SAC losses: average of the rank gradients) or "sum" (PPO's summed clipped surrogate,
        ppo.py:152-183: plain sum).  With one rank this is exactly the single-GPU step."""
        stream = N.stream_ptr(self.flat["p"].device)
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            # the all-reduce needs the local gradient first: deferred weight gradients run now,
            # without the fused optimizer
            N.check(N.lib().pa_mlp_flush_grads(self.handle, stream))
            reduce_gradient_(self.flat["grad"], reduce)
        step = self.next_adam_step()
        N.check(N.lib().pa_mlp_adam(self.handle, step, stream))
        self.stepped(step)

    def next_adam_step(self) -> int:
        return self._steps + 1

    def stepped(self, step: int) -> None:
        """Bookkeeping after a native launch applied optimizer step `step` to this network."""
        self._pending_x = None
        self._set_adam_steps(step)
        self._steps = step
        self._versions = self._flat_versions()     # our own kernels wrote the buffers: not "external"
        self._frozen_stale = self.frozen_last

    # ------------------------------------------------------------------ data-parallel pairs
    @staticmethod
    def join_grads(m1: "FlatMlp", m2: "FlatMlp") -> torch.Tensor:
        """Both networks' flat gradient buffers as two halves of ONE allocation, so that a
        data-parallel step exchanges them as one message (PPO's actor 135 696 +
        critic 131 841 floats).  Idempotent; redone when a network re-flattened itself."""
        g1, g2 = m1.flat["grad"], m2.flat["grad"]
        joint = getattr(m1, "_joint_grad", None)
        if joint is not None and joint is getattr(m2, "_joint_grad", None) \
                and g1.data_ptr() == joint.data_ptr() \
                and g2.data_ptr() == joint.data_ptr() + 4 * g1.numel():
            return joint
        n1, n2 = g1.numel(), g2.numel()
        assert n1 % 4 == 0, "flat buffers are padded to 16 bytes"
        joint = torch.zeros(n1 + n2, dtype=torch.float32, device=g1.device)
        for m, lo, n in ((m1, 0, n1), (m2, n1, n2)):
            old = m.flat["grad"]
            new = joint[lo:lo + n]
            new.copy_(old)
            for q in m._params():
                if q.grad is not None:
                    off = (q.grad.data_ptr() - old.data_ptr()) // 4
                    q.grad = new[off:off + q.numel()].view(q.shape)
            m.flat["grad"] = new
            f = m.flat
            bufs = N.MlpBuffers(p=f["p"].data_ptr(), p_target=N.ptr(f.get("p_target")),
                                grad=new.data_ptr(), exp_avg=N.ptr(f.get("exp_avg")),
                                exp_avg_sq=N.ptr(f.get("exp_avg_sq")),
                                max_exp_avg_sq=N.ptr(f.get("max_exp_avg_sq")))
            torch.cuda.current_stream(new.device).synchronize()
            N.check(N.lib().pa_mlp_bind(m.handle, C.byref(bufs)))
            m._joint_grad = joint
        return joint

    @staticmethod
    def adam_pair_data_parallel(m1: "FlatMlp", m2: "FlatMlp", force: bool = False) -> None:
        """The data-parallel step of two networks that share a batch (PPO's actor + critic):
        both networks' deferred weight gradients in ONE launch (pa_mlp_flush_grads2), ONE all-reduce
        of the joined gradient buffers (RCCL through the native hooks, _comm.allreduce_sum_), ONE
        AdamW launch for both (pa_mlp_adamw2).  The caller has already scaled the heads so that the
        SUM over ranks is the global-batch gradient (PPO: the surrogate is a sum, the critic's MSE
        head is divided by B * world)."""
        from ... import _comm
        dev = m1.flat["p"].device
        stream = N.stream_ptr(dev)
        joint = FlatMlp.join_grads(m1, m2)
        rc = N.lib().pa_mlp_flush_grads2(m1.handle, m2.handle, stream)
        if rc == N.PA_ERR_UNSUPPORTED:
            N.check(N.lib().pa_mlp_flush_grads(m1.handle, stream))
            N.check(N.lib().pa_mlp_flush_grads(m2.handle, stream))
        else:
            N.check(rc)
        _comm.allreduce_sum_(joint, force=force)
        s1, s2 = m1._steps + 1, m2._steps + 1
        N.check(N.lib().pa_mlp_adamw2(m1.handle, m2.handle, s1, s2, stream))
        for m, st in ((m1, s1), (m2, s2)):
            m._pending_x = None
            m._set_adam_steps(st)
            m._steps = st
            m._versions = m._flat_versions()

    @staticmethod
    def adam_pair(m1: "FlatMlp", m2: "FlatMlp", soft_tau: Optional[float] = None) -> bool:
        """AdamW step of twin networks (twin critics: one optimizer, same shape) whose weight
        gradients were deferred by ``backward_pair(defer=True)``: ONE launch forms both networks'
        gradients, applies AdamW and — with ``soft_tau`` — the soft update of both targets
        (pa_mlp_adam2).  Returns True when the soft update was part of it; otherwise the caller
        performs it as usual.  Falls back to two single steps whenever the pair does not qualify
        (data-parallel runs all-reduce between backward and AdamW)."""
        single = not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1)
        if single and m1._pending_x is not None and m2._pending_x is not None \
                and m1._steps == m2._steps and (soft_tau is None or (
                    m1.target_layers is not None and m2.target_layers is not None)):
            step = m1._steps + 1
            rc = N.lib().pa_mlp_adam2(m1.handle, m2.handle, step,
                                      -1.0 if soft_tau is None else float(soft_tau),
                                      N.stream_ptr(m1.flat["p"].device))
            if rc == 0:
                for m in (m1, m2):
                    m.stepped_natively()
                return soft_tau is not None
            if rc != N.PA_ERR_UNSUPPORTED:
                N.check(rc)
        m1.adam()
        m2.adam()
        return False

    def stepped_natively(self, n: int = 1) -> None:
        """Bookkeeping after ``n`` AdamW steps the library sequenced itself (pa_sac_step,
        pa_sac_learn)."""
        if n <= 0:
            return
        step = self._steps + n
        self._pending_x = None
        self._set_adam_steps(step)
        self._steps = step
        self._versions = self._flat_versions()

    def soft_update(self, tau: float) -> None:
        self.ready()
        N.check(N.lib().pa_mlp_soft_update(self.handle, float(tau), N.stream_ptr(self.device)))
