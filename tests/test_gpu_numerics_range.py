"""GPU: the bf16x3 loops (three bf16 terms per fp32 operand, six products, fp32 accumulation) outside
unit-scale data — VERDICT r5 weak-3.  Every loop that runs on the bf16 matrix pipe is driven with

  * rows whose magnitude spans 1e-20 .. 1e+20 (per-row scale factors; weights stay at unit scale),
  * rows down in the subnormal range,
  * rows that hold +inf, -inf and NaN,

next to the fp32-MFMA build of the same kernel and float64 on the host.  What is asserted:

  finite data    the error of every output, relative to that output's own sum of |terms| (the bound
                 an fp32 dot product obeys), is at most 2x the fp32-MFMA kernel's (floor 3e-7):
                 the split loses nothing over the dynamic range, row by row;
  subnormals     results within 1e-5 relative + K * 2^-126 absolute: split terms below the normal
                 range may be flushed to zero (the matrix pipe does not keep bf16 subnormals) —
                 never garbage, never non-finite;
  non-finite     (a) row isolation: every row WITHOUT a non-finite input is bit-identical to the
                 same launch on clean data (forward passes; a weight gradient sums over rows and is
                 poisoned exactly where torch's is);
                 (b) a row that is non-finite under torch's fp32 arithmetic is non-finite here —
                 never a finite number;  NaN stays NaN;
                 (c) documented difference: +-inf may surface as NaN (inf = inf + NaN + NaN after
                 the three-way split: hi = bf16(inf), mid = bf16(inf - inf)); torch's own fp32 result
                 is NaN as well as soon as two infinities of opposite sign meet in a dot product,
                 which a random weight row makes the common case."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def row_scales(B, lo=-20, hi=20):
    k = torch.arange(B) % (hi - lo + 1) + lo
    return (10.0 ** k.double()).float()


def abs_bound(x64, layers):
    """sum of |terms| an output accumulates through the network: the scale of its rounding error."""
    h = x64.abs()
    for w, b in layers:
        h = h @ w.abs().t() + b.abs()
    return h


def exact_mlp(x64, layers):
    h = x64
    for i, (w, b) in enumerate(layers):
        h = h @ w.t() + b
        if i + 1 < len(layers):
            h = torch.relu(h)
    return h


# --------------------------------------------------------------------------- target_split_kernel
def _dqn(monkeypatch, split, seed=0, zero_bias=False):
    from pearl_amd import DeepQLearning, DiscreteActionSpace, OneHotActionTensorRepresentationModule
    monkeypatch.setenv("PEARL_AMD_TARGET_SPLIT", split)
    torch.manual_seed(seed)
    A = 16
    pl = DeepQLearning(state_dim=128, action_space=DiscreteActionSpace([torch.tensor([k]) for k in range(A)]),
                       hidden_dims=[256, 256], training_rounds=1, batch_size=256,
                       action_representation_module=OneHotActionTensorRepresentationModule(A))
    if zero_bias:
        with torch.no_grad():
            for n, p in list(pl._Q.named_parameters()) + list(pl._Q_target.named_parameters()):
                if n.endswith("bias"):
                    p.zero_()
                elif n.startswith("_model.0.") and p.dim() == 2:
                    p[:, 128:].zero_()     # the one-hot action columns: nothing at unit scale is left
    return pl.to(DEV)


def _dqn_batch(next_state, term=None):
    from pearl_amd import TransitionBatch
    B, A = next_state.shape[0], 16
    g = torch.Generator().manual_seed(5)
    eye = torch.eye(A)
    return TransitionBatch(
        state=torch.randn(B, 128, generator=g).to(DEV), action=eye[torch.arange(B) % A].to(DEV),
        reward=torch.randn(B, generator=g).to(DEV),
        terminated=(torch.zeros(B, dtype=torch.bool) if term is None else term).to(DEV),
        truncated=torch.zeros(B, dtype=torch.bool, device=DEV), next_state=next_state.to(DEV),
        curr_available_actions=eye.expand(B, A, A).contiguous().to(DEV),
        curr_unavailable_actions_mask=torch.zeros(B, A, dtype=torch.bool, device=DEV),
        next_available_actions=eye.expand(B, A, A).contiguous().to(DEV),
        next_unavailable_actions_mask=torch.zeros(B, A, dtype=torch.bool, device=DEV))


def _target_layers(pl, dtype=torch.float64):
    sd = {k: v.detach().cpu().to(dtype) for k, v in pl._Q_target.state_dict().items()}
    ws = [k for k in sd if k.endswith("weight")]
    return [(sd[w], sd[w[:-6] + "bias"]) for w in ws]


def _next_values_host(pl, next_state, dtype):
    """max over the 16 actions of Q_target([s' || onehot(a)]) with torch on the host."""
    layers = _target_layers(pl, dtype)
    B, A = next_state.shape[0], 16
    x = torch.cat([next_state.to(dtype).unsqueeze(1).expand(B, A, -1), torch.eye(A, dtype=dtype).expand(B, A, A)], -1)
    q = exact_mlp(x.reshape(B * A, -1), layers).view(B, A)
    bound = abs_bound(x.reshape(B * A, -1), layers).view(B, A).max(dim=1).values
    return q.max(dim=1).values, bound


def test_target_split_kernel_over_forty_decades(monkeypatch):
    """get_next_state_values (deep_q_learning.py:130-167) through target_split_kernel with next
    states scaled row by row from 1e-20 to 1e+20, against the fp32-MFMA kernels
    (PEARL_AMD_TARGET_SPLIT=0) and float64."""
    B = 256
    g = torch.Generator().manual_seed(1)
    ns = torch.randn(B, 128, generator=g) * row_scales(B).view(B, 1)
    out = {}
    for split in ("1", "0"):
        pl = _dqn(monkeypatch, split)
        out[split] = pl.q_values_and_targets(_dqn_batch(ns))["next_v"].double().cpu()
    v64, bound = _next_values_host(pl, ns, torch.float64)
    e_split = ((out["1"] - v64).abs() / bound)
    e_fp32 = ((out["0"] - v64).abs() / bound)
    print(f"\nnext_v error / sum|terms| over 40 decades: bf16x3 {float(e_split.max()):.2e} "
          f"(row scale 1e{int(torch.log10(row_scales(B)[e_split.argmax()]))}), "
          f"fp32 MFMA {float(e_fp32.max()):.2e}")
    assert not torch.equal(out["1"], out["0"]), "the split kernel did not run"
    assert torch.isfinite(out["1"]).all()
    assert float(e_split.max()) <= max(2.0 * float(e_fp32.max()), 3e-7)
    # and decade by decade (a loss confined to the smallest or largest rows must not hide in a max)
    sc = torch.log10(row_scales(B).double()).round().long()
    for k in (-20, -10, 0, 10, 20):
        rows = sc == k
        assert float(e_split[rows].max()) <= max(2.0 * float(e_fp32[rows].max()), 3e-7), k


def test_target_split_kernel_subnormal_rows(monkeypatch):
    """Next states at 1e-28 .. 1e-40 through a network without biases or action columns: the first
    layer's activations, their bf16 split terms and the products with W2 reach and cross the
    subnormal boundary (2^-126 = 1.2e-38)."""
    B = 256
    g = torch.Generator().manual_seed(2)
    ns = torch.randn(B, 128, generator=g) * row_scales(B, -40, -28).view(B, 1)
    pl = _dqn(monkeypatch, "1", zero_bias=True)
    got = pl.q_values_and_targets(_dqn_batch(ns))["next_v"].double().cpu()
    v64, bound = _next_values_host(pl, ns, torch.float64)
    assert torch.isfinite(got).all()
    err = (got - v64).abs()
    tiny = 2.0 ** -126
    assert float(v64.abs().max()) > 1e4 * tiny           # the largest rows are well inside the normal range
    assert bool((err <= 1e-5 * v64.abs() + 3e-7 * bound + 64 * tiny).all()), float(err.max())


def test_target_split_kernel_non_finite_rows(monkeypatch):
    """Rows with +inf / -inf / NaN in the next state: the other rows are bit-identical to the clean
    launch, the poisoned rows are non-finite wherever torch's fp32 result is (NaN where it is NaN),
    and the Bellman target multiplies through — terminated rows included, as the reference's
    next_v * gamma * (1 - terminated) does (deep_td_learning.py:313-317: inf * 0 = NaN)."""
    B = 256
    g = torch.Generator().manual_seed(3)
    clean = torch.randn(B, 128, generator=g)
    ns = clean.clone()
    bad = {7: float("inf"), 40: float("-inf"), 41: float("nan"), 130: float("inf"), 255: float("nan")}
    for r, v in bad.items():
        ns[r, (r * 5) % 128] = v
    ns[130, 3] = float("-inf")
    term = torch.zeros(B, dtype=torch.bool)
    term[7] = term[41] = term[100] = True
    for split in ("1", "0"):
        pl = _dqn(monkeypatch, split)
        ref = pl.q_values_and_targets(_dqn_batch(clean, term))
        got = pl.q_values_and_targets(_dqn_batch(ns, term))
        rows = torch.ones(B, dtype=torch.bool)
        rows[list(bad)] = False
        for k in ("next_v", "target"):
            assert torch.equal(got[k].cpu()[rows], ref[k].cpu()[rows]), (split, k)
        v32, _ = _next_values_host(pl, ns, torch.float32)           # torch's fp32 arithmetic
        y32 = v32 * 0.99 * (1 - term.float()) + _dqn_batch(ns, term).reward.cpu()
        for k, want in (("next_v", v32), ("target", y32)):
            g_ = got[k].cpu()
            for r in bad:
                if not torch.isfinite(want[r]):
                    assert not torch.isfinite(g_[r]), (split, k, r, float(g_[r]), float(want[r]))
                if torch.isnan(want[r]):
                    assert torch.isnan(g_[r]), (split, k, r)
        assert not torch.isfinite(got["target"].cpu()[7])            # terminated + inf: inf * 0


# ------------------------------------------------------------------- online_rowpass_h2_kernel
# The online row pass on the fp16 matrix pipe (pearl_amd/csrc/online_f16_kernel.hpp): operands scaled by
# powers of two into fp16's range, two fp16 terms each, three products.  Same three properties, against
# the fp32-MFMA row pass (PEARL_AMD_ROWPASS_H2=0) and float64; plus weight ROWS spread over forty
# decades (the per-row scales come from maxima the optimizer epilogue / repack maintain) and the
# gradients of one learn_batch (the backward product carries W2's row scales on the other operand).
def _dqn_h2(monkeypatch, h2, seed=0, unit_decades=None):
    monkeypatch.setenv("PEARL_AMD_ROWPASS_H2", h2)
    pl = _dqn(monkeypatch, "1", seed=seed)
    if unit_decades is not None:
        # rows of W1 / W2 scaled by 10^k (k cycling through the range), the NEXT layer's columns by
        # 10^-k: the same function in exact arithmetic, every unit's pre-activation at another scale
        lo, hi = unit_decades
        with torch.no_grad():
            for net in (pl._Q, pl._Q_target):
                lin = [m for m in net.modules() if isinstance(m, torch.nn.Linear)]
                for i in (0, 1):
                    n = lin[i].weight.shape[0]
                    f = (10.0 ** (torch.arange(n) % (hi - lo + 1) + lo).double()).float().to(DEV)
                    lin[i].weight.mul_(f.view(-1, 1))
                    lin[i].bias.mul_(f)
                    lin[i + 1].weight.div_(f.view(1, -1))
    return pl


def _online_layers(pl, dtype=torch.float64):
    sd = {k: v.detach().cpu().to(dtype) for k, v in pl._Q.state_dict().items()}
    ws = [k for k in sd if k.endswith("weight")]
    return [(sd[w], sd[w[:-6] + "bias"]) for w in ws]


def _q_host(pl, batch, dtype):
    layers = _online_layers(pl, dtype)
    x = torch.cat([batch.state.cpu().to(dtype), batch.action.cpu().to(dtype)], -1)
    return exact_mlp(x, layers).view(-1), abs_bound(x, layers).view(-1)


def _batch_with_states(state):
    b = _dqn_batch(torch.randn(state.shape[0], 128, generator=torch.Generator().manual_seed(9)))
    b.state = state.to(DEV)
    return b


@pytest.mark.parametrize("unit_decades", [None, (-2, 2)])
def test_rowpass_h2_over_forty_decades(monkeypatch, unit_decades):
    """Q(s, a) (q_value_networks.py:152-174) through online_rowpass_h2_kernel with states scaled row by
    row from 1e-20 to 1e+20 (and, second case, hidden units whose weight rows span four decades: a term keeps its 22 bits while it is within 2^-17 of its row's maximum, its precision is 2^-40 of that maximum below),
    against the fp32-MFMA row pass and float64."""
    B = 256
    g = torch.Generator().manual_seed(11)
    st = torch.randn(B, 128, generator=g) * row_scales(B).view(B, 1)
    out = {}
    for h2 in ("1", "0"):
        pl = _dqn_h2(monkeypatch, h2, unit_decades=unit_decades)
        batch = _batch_with_states(st)
        out[h2] = pl.q_values_and_targets(batch)["q"].double().cpu()
    q64, bound = _q_host(pl, batch, torch.float64)
    e_h2 = (out["1"] - q64).abs() / bound
    e_f32 = (out["0"] - q64).abs() / bound
    print(f"\nQ(s, a) error / sum|terms| over 40 decades (unit decades {unit_decades}): fp16x2 "
          f"{float(e_h2.max()):.2e}, fp32 MFMA {float(e_f32.max()):.2e}")
    assert not torch.equal(out["1"], out["0"]), "the fp16 row pass did not run"
    assert torch.isfinite(out["1"]).all()
    assert float(e_h2.max()) <= max(2.0 * float(e_f32.max()), 3e-7)
    sc = torch.log10(row_scales(B).double()).round().long()
    for k in (-20, -10, 0, 10, 20):
        rows = sc == k
        assert float(e_h2[rows].max()) <= max(2.0 * float(e_f32[rows].max()), 3e-7), k


def test_rowpass_h2_subnormal_rows(monkeypatch):
    """States at 1e-28 .. 1e-40 through a network without biases or action columns: row maxima below
    the scaling's clamp (2^-112) keep their fp32 exponent and lose fp16 terms instead — never garbage."""
    B = 256
    g = torch.Generator().manual_seed(12)
    st = torch.randn(B, 128, generator=g) * row_scales(B, -40, -28).view(B, 1)
    monkeypatch.setenv("PEARL_AMD_ROWPASS_H2", "1")
    pl = _dqn(monkeypatch, "1", zero_bias=True)
    batch = _batch_with_states(st)
    # (an activation row's precision is relative to the row's maximum, and x = state || rep(action):
    #  a one-hot 1.0 next to states of 1e-30 leaves them below 2^-40 of the maximum — as it leaves
    #  them below the rounding of any fp32 sum that contains the action's term.  The clamp path is
    #  what this test is about: no action representation in the row)
    batch.action = torch.zeros_like(batch.action)
    got = pl.q_values_and_targets(batch)["q"].double().cpu()
    q64, bound = _q_host(pl, batch, torch.float64)
    assert torch.isfinite(got).all()
    err = (got - q64).abs()
    tiny = 2.0 ** -126
    assert float(q64.abs().max()) > 1e4 * tiny
    # rows whose maximum is below 2^-112 are scaled by the clamp's 2^126: their terms sit up to 2^14
    # below fp16's normal range, i.e. an absolute error of ~2^-24 of 2^-112 per term
    assert bool((err <= 1e-5 * q64.abs() + 3e-7 * bound + 512 * 2.0 ** -136).all()), float(err.max())


def test_rowpass_h2_non_finite_rows(monkeypatch):
    """Rows with +inf / -inf / NaN in the state: the other rows are bit-identical to the clean launch
    (the activation scales are per row), the poisoned rows are non-finite wherever torch's fp32
    result is, NaN where it is NaN."""
    B = 256
    g = torch.Generator().manual_seed(13)
    clean = torch.randn(B, 128, generator=g)
    st = clean.clone()
    bad = {7: float("inf"), 40: float("-inf"), 41: float("nan"), 130: float("inf"), 255: float("nan")}
    for r, v in bad.items():
        st[r, (r * 5) % 128] = v
    st[130, 3] = float("-inf")
    for h2 in ("1", "0"):
        pl = _dqn_h2(monkeypatch, h2)
        ref = pl.q_values_and_targets(_batch_with_states(clean))["q"].cpu()
        batch = _batch_with_states(st)
        got = pl.q_values_and_targets(batch)["q"].cpu()
        rows = torch.ones(B, dtype=torch.bool)
        rows[list(bad)] = False
        assert torch.equal(got[rows], ref[rows]), h2
        q32, _ = _q_host(pl, batch, torch.float32)
        for r in bad:
            if not torch.isfinite(q32[r]):
                assert not torch.isfinite(got[r]), (h2, r, float(got[r]), float(q32[r]))
            if torch.isnan(q32[r]):
                assert torch.isnan(got[r]), (h2, r)


@pytest.mark.parametrize("unit_decades", [None, (-2, 2)])
def test_rowpass_h2_gradients_match_float64_as_the_fp32_pass_does(monkeypatch, unit_decades):
    """One learn_batch on states spread over eight decades: every gradient tensor is as close to
    float64 autograd as the fp32-MFMA row pass puts it (the backward product G = s2 W2 runs on fp16
    terms with W2's row scales moved onto s2)."""
    B = 256
    g = torch.Generator().manual_seed(14)
    st = torch.randn(B, 128, generator=g) * row_scales(B, -4, 4).view(B, 1)
    grads = {}
    for h2 in ("1", "0"):
        pl = _dqn_h2(monkeypatch, h2, unit_decades=unit_decades)
        batch = _batch_with_states(st)
        y = pl.q_values_and_targets(batch)["target"].double().cpu()
        if h2 == "1":
            layers = [(w.clone().requires_grad_(True), b.clone().requires_grad_(True)) for w, b in _online_layers(pl)]
            x = torch.cat([batch.state.cpu().double(), batch.action.cpu().double()], -1)
            loss = ((exact_mlp(x, layers).view(-1) - y) ** 2).mean()
            loss.backward()
            want = [t.grad for wb in layers for t in wb]
        pl.learn_batch(batch)
        grads[h2] = [p.grad.double().cpu() for p in pl._Q.parameters()]
    assert len(want) == len(grads["1"])
    for i, (w64, a, b) in enumerate(zip(want, grads["1"], grads["0"])):
        assert torch.isfinite(a).all(), i
        scale = float(w64.abs().max())
        e_h2, e_f32 = float((a - w64).abs().max()) / scale, float((b - w64).abs().max()) / scale
        print(f"\ngradient tensor {i} (unit decades {unit_decades}): max error / max|g| fp16x2 {e_h2:.2e}, fp32 {e_f32:.2e}")
        assert e_h2 <= max(3.0 * e_f32, 2e-6), (i, e_h2, e_f32)


# ------------------------------------------------------------------- weight-gradient split loops
def _dw(dz, x, mode):
    from pearl_amd import _native as N
    lib = N.lib()
    B, M = dz.shape
    N_ = x.shape[1]
    dev = torch.device(DEV)
    dzd, xd = dz.to(dev), x.to(dev)
    N.check(lib.pa_debug_set_dw_split(mode))
    try:
        dw = torch.full((M, N_), 123.0, device=dev)
        db = torch.full((M,), 123.0, device=dev)
        N.check(lib.pa_debug_weight_grad(dzd.data_ptr(), M, xd.data_ptr(), N_, dw.data_ptr(), N_,
                                         db.data_ptr(), M, N_, B, N.stream_ptr(dev)))
        torch.cuda.synchronize()
        return dw.cpu(), db.cpu()
    finally:
        N.check(lib.pa_debug_set_dw_split(-1))


@pytest.mark.parametrize("M,N_,B", [(256, 256, 4096), (256, 144, 1024), (64, 512, 4096)])
def test_weight_grad_split_loops_over_forty_decades(M, N_, B):
    """dW = dz^T x with the rows of dz scaled 1e-20 .. 1e+20 and the matching rows of x scaled the
    other way (every product is O(1), every OPERAND is not): both tile shapes of the bf16x3 loop."""
    g = torch.Generator().manual_seed(11)
    s = row_scales(B).view(B, 1)
    dz = torch.randn(B, M, generator=g) * s
    x = torch.randn(B, N_, generator=g) / s
    exact = dz.double().t() @ x.double()
    scale = dz.double().abs().t() @ x.double().abs()
    dw32, db32 = _dw(dz, x, 0)
    dws, dbs = _dw(dz, x, 2)
    assert torch.isfinite(dws).all() and not torch.equal(dws, dw32)
    e32 = float(((dw32.double() - exact).abs() / scale).max())
    es = float(((dws.double() - exact).abs() / scale).max())
    print(f"\ndW error / sum|terms|, operands over 40 decades: fp32 MFMA {e32:.2e}, bf16x3 {es:.2e}")
    assert es <= max(2.0 * e32, 3e-7)
    # the bias gradient sums dz itself: its terms span 40 decades, the bound is the sum of |dz|
    sb = dz.double().abs().sum(0)
    eb32 = float(((db32.double() - dz.double().sum(0)).abs() / sb).max())
    ebs = float(((dbs.double() - dz.double().sum(0)).abs() / sb).max())
    assert ebs <= max(2.0 * eb32, 3e-7)


def test_weight_grad_split_loop_non_finite_rows():
    """A non-finite entry of dz or x poisons exactly the dW entries it poisons under torch's fp32
    arithmetic (column m of dz -> row m of dW; column n of x -> column n of dW): same NaN pattern,
    non-finite wherever torch is, every other entry bit-identical to the clean launch."""
    M, N_, B = 256, 256, 4096
    g = torch.Generator().manual_seed(12)
    dz, x = torch.randn(B, M, generator=g), torch.randn(B, N_, generator=g)
    clean = _dw(dz, x, 2)
    dz2, x2 = dz.clone(), x.clone()
    dz2[17, 3] = float("nan")
    dz2[2000, 100] = float("inf")
    x2[999, 7] = float("-inf")
    x2[4095, 255] = float("nan")
    got_w, got_b = _dw(dz2, x2, 2)
    want = dz2.t() @ x2                                   # torch, fp32, host
    touched = torch.zeros(M, N_, dtype=torch.bool)
    touched[3, :] = touched[100, :] = True
    touched[:, 7] = touched[:, 255] = True
    assert torch.equal(got_w[~touched], clean[0][~touched])
    assert bool((~torch.isfinite(got_w[touched])).all()) and bool((~torch.isfinite(want[touched])).all())
    assert bool(torch.isnan(got_w[torch.isnan(want)]).all())
    bad_b = torch.zeros(M, dtype=torch.bool)
    bad_b[3] = bad_b[100] = True
    assert torch.equal(got_b[~bad_b], clean[1][~bad_b]) and bool((~torch.isfinite(got_b[bad_b])).all())


# ------------------------------------------------------------- the fused row step (PPO, 4096 rows)
def _rowstep(x, mode, seed=32, S=256, A=16, hidden=(256, 256)):
    from torch import nn, optim
    from pearl_amd import _native as N
    from pearl_amd.policy_learners.sequential_decision_making.flat_mlp import FlatMlp, layers_of
    B = x.shape[0]
    da, dc = [S] + list(hidden) + [A], [S] + list(hidden) + [1]
    g = torch.Generator().manual_seed(7)
    arep = torch.nn.functional.one_hot(torch.randint(0, A, (B,), generator=g), A).float().to(DEV)
    p_old = (torch.rand(B, generator=g) * 0.5 + 0.05).to(DEV)
    gae, lam = torch.randn(B, generator=g).to(DEV), torch.randn(B, generator=g).to(DEV)
    N.check(N.lib().pa_debug_set_rowstep_split(mode))
    try:
        torch.manual_seed(seed)
        an = [nn.Linear(da[i], da[i + 1]).to(DEV) for i in range(len(da) - 1)]
        cn = [nn.Linear(dc[i], dc[i + 1]).to(DEV) for i in range(len(dc) - 1)]
        ao = optim.AdamW([p for l in an for p in l.parameters()], lr=1e-3, amsgrad=True)
        co = optim.AdamW([p for l in cn for p in l.parameters()], lr=1e-3, amsgrad=True)
        actor = FlatMlp(layers_of(an), ao, max_batch=B).ensure(B)
        critic = FlatMlp(layers_of(cn), co, max_batch=B).ensure(B)
        xd = x.to(DEV)
        logits, value = torch.empty(B, A, device=DEV), torch.empty(B, 1, device=DEV)
        d_logits, dv = torch.empty(B, A, device=DEV), torch.empty(B, device=DEV)
        losses = torch.empty(2, device=DEV)
        N.check(N.lib().pa_ppo_rowstep(
            actor.handle, critic.handle, xd.data_ptr(), xd.stride(0), B, arep.data_ptr(), arep.stride(0),
            p_old.data_ptr(), gae.data_ptr(), 0.1, 0.01, lam.data_ptr(), 2.0 / B, logits.data_ptr(),
            logits.stride(0), value.data_ptr(), value.stride(0), d_logits.data_ptr(), d_logits.stride(0),
            dv.data_ptr(), losses.data_ptr(), N.stream_ptr(xd.device)))
        used = int(N.lib().pa_rowstep_last_split())
        torch.cuda.synchronize()
        layers = lambda net: [(l.weight.detach().double().cpu(), l.bias.detach().double().cpu()) for l in net]
        return dict(used=used, logits=logits.cpu(), value=value.view(-1).cpu(), d_logits=d_logits.cpu(),
                    dv=dv.cpu(), actor=layers(an), critic=layers(cn))
    finally:
        N.check(N.lib().pa_debug_set_rowstep_split(-1))


def test_rowstep_bf16x3_forward_over_forty_decades():
    B = 4096
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, 256, generator=g) * row_scales(B).view(B, 1)
    r32, rs = _rowstep(x, 0), _rowstep(x, -1)
    assert r32["used"] == 0 and rs["used"] == 2
    x64 = x.double()
    for name, net, key in (("logits", "actor", "logits"), ("value", "critic", "value")):
        want = exact_mlp(x64, rs[net])
        bound = abs_bound(x64, rs[net])
        if key == "value":
            want, bound = want.view(-1), bound.view(-1)
        e32 = float(((r32[key].double() - want).abs() / bound).max())
        es = float(((rs[key].double() - want).abs() / bound).max())
        print(f"\nrow step {name} error / sum|terms| over 40 decades: fp32 MFMA {e32:.2e}, bf16x3 {es:.2e}")
        assert torch.isfinite(rs[key]).all()
        assert es <= max(2.0 * e32, 3e-7), name


def test_rowstep_bf16x3_non_finite_rows_stay_in_their_rows():
    B = 4096
    g = torch.Generator().manual_seed(22)
    clean = torch.randn(B, 256, generator=g)
    x = clean.clone()
    bad = {5: float("inf"), 33: float("nan"), 2049: float("-inf"), 4095: float("nan")}
    for r, v in bad.items():
        x[r, (r * 3) % 256] = v
    ref, got = _rowstep(clean, -1), _rowstep(x, -1)
    assert got["used"] == 2
    rows = torch.ones(B, dtype=torch.bool)
    rows[list(bad)] = False
    for key in ("logits", "value", "d_logits", "dv"):
        assert torch.equal(got[key][rows], ref[key][rows]), key
    for net, key in (("actor", "logits"), ("critic", "value")):
        want = exact_mlp(x.float(), [(w.float(), b.float()) for w, b in got[net]])   # torch fp32
        want = want.view(B, -1)
        g_ = got[key].view(B, -1)
        for r in bad:
            nf = ~torch.isfinite(want[r])
            assert bool((~torch.isfinite(g_[r][nf])).all()), (key, r)
            assert bool(torch.isnan(g_[r][torch.isnan(want[r])]).all()), (key, r)
