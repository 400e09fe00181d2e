"""GPU: single-kernel checks through the C ABI (pa_debug_*), against fp64 torch on the same
inputs.  Asymmetric operands so a transposed fragment map cannot pass."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from pearl_amd import _native as N
    return N, N.lib()


def _rand(*shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32)


@pytest.mark.parametrize("M,N_,K", [(1024, 256, 144), (1024, 256, 256), (1024, 256, 128),
                                   (128, 64, 6), (33, 70, 37), (7, 5, 3), (100, 24, 21)])
@pytest.mark.parametrize("epi", [0, 1])
def test_linear_xwT(M, N_, K, epi):
    N, lib = _lib()
    dev = torch.device("cuda:0")
    x, w, b = _rand(M, K, seed=1), _rand(N_, K, seed=2), _rand(N_, seed=3)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    out = torch.full((M, N_), float("nan"), device=dev)
    N.check(lib.pa_debug_linear(xd.data_ptr(), K, wd.data_ptr(), K, out.data_ptr(), N_,
                                bd.data_ptr(), None, 0, M, N_, K, 0, epi, N.stream_ptr(dev)))
    ref = x.double() @ w.double().t() + b.double()
    if epi == 1:
        ref = ref.clamp_min(0)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-5 * K ** 0.5)


@pytest.mark.parametrize("M,N_,K", [(1024, 256, 256), (64, 24, 16), (45, 70, 33), (9, 3, 5)])
def test_linear_dx_masked(M, N_, K):
    """dX = (dY W) * [H > 0] with W stored [K, N]."""
    N, lib = _lib()
    dev = torch.device("cuda:0")
    dy, w, hm = _rand(M, K, seed=4), _rand(K, N_, seed=5), _rand(M, N_, seed=6)
    dyd, wd, hd = dy.to(dev), w.to(dev), hm.to(dev)
    out = torch.full((M, N_), float("nan"), device=dev)
    N.check(lib.pa_debug_linear(dyd.data_ptr(), K, wd.data_ptr(), N_, out.data_ptr(), N_, None,
                                hd.data_ptr(), N_, M, N_, K, 1, 2, N.stream_ptr(dev)))
    ref = (dy.double() @ w.double()) * (hm > 0)
    torch.testing.assert_close(out.cpu().double(), ref, rtol=1e-5, atol=1e-5 * K ** 0.5)


@pytest.mark.parametrize("M,N_,B", [(256, 256, 1024), (256, 144, 1024), (64, 6, 128), (24, 21, 100),
                                   (40, 33, 7), (5, 3, 1),
                                   # split-K path (batch cut across workgroups, ticketed combine)
                                   (256, 256, 4096), (65, 66, 4099), (1, 256, 2500), (16, 257, 9000)])
def test_weight_grad(M, N_, B):
    N, lib = _lib()
    dev = torch.device("cuda:0")
    dz, x = _rand(B, M, seed=7), _rand(B, N_, seed=8)
    dzd, xd = dz.to(dev), x.to(dev)
    dw = torch.full((M, N_), float("nan"), device=dev)
    db = torch.full((M,), float("nan"), device=dev)
    N.check(lib.pa_debug_weight_grad(dzd.data_ptr(), M, xd.data_ptr(), N_, dw.data_ptr(), N_,
                                     db.data_ptr(), M, N_, B, N.stream_ptr(dev)))
    torch.testing.assert_close(dw.cpu().double(), dz.double().t() @ x.double(), rtol=1e-5,
                               atol=1e-5 * B ** 0.5)
    torch.testing.assert_close(db.cpu().double(), dz.double().sum(0), rtol=1e-5,
                               atol=1e-5 * B ** 0.5)


@pytest.mark.parametrize("M,N_,B", [(256, 256, 4096), (16, 256, 4096), (64, 512, 4096), (256, 256, 2500),
                                   (128, 96, 300), (256, 144, 8192)])
def test_weight_grad_bf16x3_split_loop_has_fp32_accuracy(M, N_, B):
    """The weight-gradient main loop on the bf16 matrix pipe (weight_grad_split_kernel: every operand
    split exactly into three bf16 terms, six products, fp32 accumulation) against float64, next to
    the fp32-MFMA kernel on the same operands: its error relative to sum |dz x| must be that of an
    fp32 dot product (no worse than 2x the fp32 kernel's own, floor 2e-7), the bias gradient too,
    ragged batches (rows past B read as zeros), and two launches agree bitwise."""
    N, lib = _lib()
    dev = torch.device("cuda:0")
    dz, x = _rand(B, M, seed=11), _rand(B, N_, seed=12)
    dz[::3] *= 37.0            # a wide dynamic range across rows
    x[:, ::5] *= 1e-3
    dzd, xd = dz.to(dev), x.to(dev)
    exact = dz.double().t() @ x.double()
    scale = dz.double().abs().t() @ x.double().abs()
    exact_b, scale_b = dz.double().sum(0), dz.double().abs().sum(0)

    def run(mode):
        N.check(lib.pa_debug_set_dw_split(mode))
        try:
            dw = torch.full((M, N_), float("nan"), device=dev)
            db = torch.full((M,), float("nan"), device=dev)
            N.check(lib.pa_debug_weight_grad(dzd.data_ptr(), M, xd.data_ptr(), N_, dw.data_ptr(), N_,
                                             db.data_ptr(), M, N_, B, N.stream_ptr(dev)))
            torch.cuda.synchronize()
            return dw.cpu(), db.cpu()
        finally:
            N.check(lib.pa_debug_set_dw_split(-1))

    dw32, db32 = run(0)
    dws, dbs = run(2)
    dws2, dbs2 = run(2)
    assert torch.equal(dws, dws2) and torch.equal(dbs, dbs2)
    assert not torch.equal(dws, dw32), "the split loop did not run (same bits as the fp32 kernel)"
    e32 = float(((dw32.double() - exact).abs() / scale).max())
    es = float(((dws.double() - exact).abs() / scale).max())
    eb32 = float(((db32.double() - exact_b).abs() / scale_b).max())
    ebs = float(((dbs.double() - exact_b).abs() / scale_b).max())
    print(f"\ndW error / sum|terms|: fp32 MFMA {e32:.2e}, bf16x3 {es:.2e}; db: {eb32:.2e}, {ebs:.2e}")
    assert es <= max(2.0 * e32, 2e-7), (es, e32)
    assert ebs <= max(2.0 * eb32, 2e-7), (ebs, eb32)


def test_weight_grad_split_k_is_deterministic():
    """Two launches of the split-K path give bitwise-equal results (slice-ordered combine) and
    leave the tickets ready for the next launch."""
    N, lib = _lib()
    dev = torch.device("cuda:0")
    M, N_, B = 128, 96, 4096
    dz, x = _rand(B, M, seed=3).to(dev), _rand(B, N_, seed=4).to(dev)
    outs = []
    for _ in range(3):
        dw = torch.full((M, N_), float("nan"), device=dev)
        db = torch.full((M,), float("nan"), device=dev)
        N.check(lib.pa_debug_weight_grad(dz.data_ptr(), M, x.data_ptr(), N_, dw.data_ptr(), N_,
                                         db.data_ptr(), M, N_, B, N.stream_ptr(dev)))
        outs.append((dw.clone(), db.clone()))
    for dw, db in outs[1:]:
        assert torch.equal(dw, outs[0][0]) and torch.equal(db, outs[0][1])


def test_linear_is_deterministic():
    N, lib = _lib()
    dev = torch.device("cuda:0")
    x, w, b = _rand(1024, 256, seed=1).to(dev), _rand(256, 256, seed=2).to(dev), _rand(256, seed=3).to(dev)
    outs = []
    for _ in range(3):
        out = torch.empty(1024, 256, device=dev)
        N.check(lib.pa_debug_linear(x.data_ptr(), 256, w.data_ptr(), 256, out.data_ptr(), 256,
                                    b.data_ptr(), None, 0, 1024, 256, 256, 0, 1, N.stream_ptr(dev)))
        outs.append(out.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])


@pytest.mark.parametrize("dtype", [torch.int64, torch.int32, torch.float32])
def test_one_hot_kernel(dtype):
    from pearl_amd import OneHotActionTensorRepresentationModule
    dev = torch.device("cuda:0")
    rep = OneHotActionTensorRepresentationModule(7)
    idx = torch.tensor([[0], [6], [3], [3]], dtype=dtype)
    got = rep(idx.to(dev)).cpu()
    want = torch.nn.functional.one_hot(idx.long(), 7).squeeze(-2).float()
    assert torch.equal(got, want)
    tab = torch.tensor([[[0.], [2.], [4.], [0.], [0.]]])  # test_dynamic_action_space.py:28-157
    got = OneHotActionTensorRepresentationModule(5)(tab.to(dev)).cpu()
    assert torch.equal(got, torch.nn.functional.one_hot(tab.long(), 5).squeeze(-2).float())
    assert got.shape == (1, 5, 5)


@pytest.mark.parametrize("kw,dims,B", [
    (dict(use_batch_norm=True), [9, 20, 12, 5], 37),
    (dict(dropout_ratio=0.3), [9, 20, 12, 5], 37),
    (dict(use_skip_connections=True), [16, 16, 16, 16], 50),            # every layer wrapped, the last too
    (dict(use_batch_norm=True, use_layer_norm=True, dropout_ratio=0.2, use_skip_connections=True,
          hidden_activation="leaky_relu"), [24, 24, 70, 70, 8], 129),
    (dict(use_batch_norm=True, use_skip_connections=True, hidden_activation="tanh"), [64, 64, 64, 3], 1000),
])
def test_mlp_block_batch_norm_dropout_skip_match_torch_autograd(kw, dims, B):
    """mlp_block's remaining options (common/utils.py:113-131, :142-150) in the generic engine's
    layer-by-layer path against torch's own modules on the same device: the kept forward (BatchNorm1d
    in training mode, the same dropout keep masks, residual adds), the input gradient, every
    parameter gradient (Linear, LayerNorm, BatchNorm1d) and the running statistics a forward leaves
    behind — then a target-network forward (its own running statistics) and an eval-mode forward."""
    import copy
    import torch.nn as nn
    from pearl_amd.neural_networks.common.utils import mlp_block
    from pearl_amd.policy_learners.sequential_decision_making.generic_q import flat_mlp_of
    dev = torch.device("cuda:0")
    torch.manual_seed(5)
    model = mlp_block(dims[0], dims[1:-1], dims[-1], **kw).to(dev)
    for m in model.modules():           # non-trivial affine parameters and statistics
        if isinstance(m, (nn.LayerNorm, nn.BatchNorm1d)):
            with torch.no_grad():
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    ref, target = copy.deepcopy(model), copy.deepcopy(model)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3, amsgrad=True)
    net = flat_mlp_of(model, target, opt, B, "test network")
    assert not net.plain
    g = torch.Generator(device=dev).manual_seed(6)
    x = torch.randn(B, dims[0], device=dev, generator=g)
    d_out = torch.randn(B, dims[-1], device=dev, generator=g)
    p = kw.get("dropout_ratio", 0.0)
    masks = [torch.empty(B, d, device=dev).bernoulli_(1 - p, generator=g) for d in dims[1:-1]] if p else []
    net.dropout_source = lambda li, tgt, B_, d, dv: masks[li]
    calls = []
    real = nn.Dropout.forward
    nn.Dropout.forward = lambda self, t: (calls.append(1), t * masks[len(calls) - 1].div(1 - self.p))[1] \
        if self.training else t
    try:
        xr = x.clone().requires_grad_(True)
        want = ref(xr)
        want.backward(d_out)
    finally:
        nn.Dropout.forward = real
    got = net.forward(x, keep=True)
    torch.testing.assert_close(got, want.detach(), rtol=2e-5, atol=2e-5)
    dx = net.backward(x, d_out, want_dw=True, want_dx=True)
    torch.cuda.synchronize()
    torch.testing.assert_close(dx, xr.grad, rtol=2e-4, atol=2e-5)
    for (k, pm), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(pm.grad, pr.grad, rtol=2e-4, atol=2e-5 * max(1.0, float(pr.grad.abs().max())), msg=k)
    for (k, bm), (_, br) in zip(model.named_buffers(), ref.named_buffers()):
        torch.testing.assert_close(bm.float(), br.float(), rtol=1e-5, atol=1e-6, msg=k)   # running_mean / _var / count
    # the target copy: its own parameters, its own running statistics
    tref = copy.deepcopy(target)
    calls.clear()
    nn.Dropout.forward = lambda self, t: (calls.append(1), t * masks[len(calls) - 1].div(1 - self.p))[1] \
        if self.training else t
    try:
        with torch.no_grad():
            twant = tref(x)
    finally:
        nn.Dropout.forward = real
    torch.testing.assert_close(net.forward(x, use_target=True), twant, rtol=2e-5, atol=2e-5)
    for (k, bm), (_, br) in zip(target.named_buffers(), tref.named_buffers()):
        torch.testing.assert_close(bm.float(), br.float(), rtol=1e-5, atol=1e-6, msg=f"target {k}")
    # one optimizer step through the engine = torch.optim.AdamW on the reference's gradients
    ropt = torch.optim.AdamW(ref.parameters(), lr=1e-3, amsgrad=True)
    ropt.step()
    net.adam()
    torch.cuda.synchronize()
    for (k, pm), (_, pr) in zip(model.named_parameters(), ref.named_parameters()):
        torch.testing.assert_close(pm.detach(), pr.detach(), rtol=1e-4, atol=2e-6, msg=f"after AdamW {k}")
