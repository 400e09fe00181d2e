#!/usr/bin/env python3
"""Trajectory of cfg2_shape_small_batch with the paired / single row pass against the fixture."""
import os, random, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_dqn import make_learner, fill_arena_buffer
from oracle import pearl_oracle as O
fx = torch.load("tests/golden/dqn_cfg2_shape_small_batch.pt", map_location="cpu", weights_only=False)
res = {}
for mode in ("1", "0"):
    os.environ["PEARL_AMD_ROWPASS_PAIR"] = mode
    for ov in ("1", "0"):
        os.environ["PEARL_AMD_OVERLAP"] = ov
        pl = make_learner(fx)
        rb = fill_arena_buffer(fx, "python")
        random.seed(fx["learn_seed"])
        rep = pl.learn(rb)
        sd = {k: v.cpu() for k, v in pl._Q.state_dict().items()}
        res[(mode, ov)] = sd
        print(f"pair={mode} overlap={ov}: loss diff {float((torch.tensor(rep['loss']) - fx['learn_losses']).abs().max()):.3g}")
        for k in O.PARAM_KEYS:
            d = (sd[k] - fx["params_after"][k]).abs()
            tol = 2e-5 + 1e-3 * fx["params_after"][k].abs()
            bad = (d > tol)
            print(f"   {k:24s} max abs diff {float(d.max()):.3g}  violations {int(bad.sum())} of {d.numel()}"
                  + (f"  worst at {tuple(int(i) for i in torch.nonzero(bad)[0])}" if bad.any() else ""))
a, b = res[("1", "1")], res[("0", "1")]
for k in O.PARAM_KEYS:
    d = (a[k] - b[k]).abs()
    print(f"pair vs single {k:24s} max abs diff {float(d.max()):.3g} at {tuple(int(i) for i in torch.nonzero(d == d.max())[0])}")

# ---- one batch: gradients against float64 autograd
from test_gpu_dqn import batch_from
print("one learn_batch: max |grad - float64| / max |float64 grad| per tensor")
bp = fx["batch_pre"]
x = torch.cat([bp["state"], bp["action"]], dim=-1).double()
w = {k: v.double().requires_grad_(True) for k, v in fx["params0"].items()}
h = torch.relu(x @ w["_model.0.0.weight"].t() + w["_model.0.0.bias"])
h = torch.relu(h @ w["_model.1.0.weight"].t() + w["_model.1.0.bias"])
q = (h @ w["_model.2.0.weight"].t() + w["_model.2.0.bias"]).squeeze(-1)
loss = torch.nn.functional.mse_loss(q, fx["target"].double())
loss.backward()
g64 = {k: v.grad for k, v in w.items()}
def report(tag, grads):
    out = []
    for k in O.PARAM_KEYS:
        e = (grads[k].double() - g64[k]).abs()
        rel_small = (e / (g64[k].abs() + 1e-30))[g64[k].abs() > 0]
        out.append(f"{float(e.max() / g64[k].abs().max()):.2e}")
    print(f"   {tag:18s} " + "  ".join(out))
report("reference (fixture)", fx["grads"])
for mode in ("1", "0"):
    os.environ["PEARL_AMD_ROWPASS_PAIR"] = mode
    pl = make_learner(fx)
    pl.learn_batch(batch_from(fx, "batch_pre"))
    report(f"hip pair={mode}", {k: p.grad.cpu() for k, p in pl._Q.named_parameters()})

# ---- where do the two row passes part?  n rounds of learn() each
print("pair vs single after n rounds: max |W1 diff|, |W2 diff|, elements of W1 beyond 1e-6")
for n in (1, 2, 3, 4, 6):
    out = {}
    for mode in ("1", "0"):
        os.environ["PEARL_AMD_ROWPASS_PAIR"] = mode
        os.environ["PEARL_AMD_OVERLAP"] = "0"
        pl = make_learner(fx)
        pl._training_rounds = n
        rb = fill_arena_buffer(fx, "python")
        random.seed(fx["learn_seed"])
        pl.learn(rb)
        out[mode] = {k: v.cpu() for k, v in pl._Q.state_dict().items()}
    d1 = (out["1"]["_model.0.0.weight"] - out["0"]["_model.0.0.weight"]).abs()
    d2 = (out["1"]["_model.1.0.weight"] - out["0"]["_model.1.0.weight"]).abs()
    rows = torch.nonzero((d1 > 1e-6).any(dim=1)).flatten().tolist()
    print(f"   n={n}: W1 {float(d1.max()):.3g}  W2 {float(d2.max()):.3g}  W1 elements > 1e-6: {int((d1 > 1e-6).sum())} in rows {rows[:12]}")

print("gradient of the LAST round after n rounds, pair vs single: max |diff| / max |g|, rows (units) of W1 with |diff| > 1e-5 max|g|")
for n in (1, 2):
    out = {}
    for mode in ("1", "0"):
        os.environ["PEARL_AMD_ROWPASS_PAIR"] = mode
        os.environ["PEARL_AMD_OVERLAP"] = "0"
        pl = make_learner(fx)
        pl._training_rounds = n
        rb = fill_arena_buffer(fx, "python")
        random.seed(fx["learn_seed"])
        pl.learn(rb)
        out[mode] = {k: p.grad.cpu().clone() for k, p in pl._Q.named_parameters()}
    for k in O.PARAM_KEYS:
        a, b = out["1"][k], out["0"][k]
        d = (a - b).abs()
        line = f"   n={n} {k:20s} {float(d.max() / b.abs().max()):.3g}  (max|g| {float(b.abs().max()):.3g}, median|g| {float(b.abs().median()):.3g})"
        if k == "_model.0.0.weight":
            rows = torch.nonzero((d > 1e-5 * b.abs().max()).any(dim=1)).flatten().tolist()
            cols = torch.nonzero((d > 1e-5 * b.abs().max()).any(dim=0)).flatten().tolist()
            line += f" rows {rows[:16]} ({len(rows)}) cols {cols[:16]} ({len(cols)})"
        print(line)

import ctypes as C
from pearl_amd import _native as N
print("workspaces of round n, pair vs single: max |diff| / max |single|")
for n in (1, 2):
    out = {}
    for mode in ("1", "0"):
        os.environ["PEARL_AMD_ROWPASS_PAIR"] = mode
        os.environ["PEARL_AMD_OVERLAP"] = "0"
        pl = make_learner(fx)
        pl._training_rounds = n
        rb = fill_arena_buffer(fx, "python")
        random.seed(fx["learn_seed"])
        pl.learn(rb)
        B, H = fx["config"]["B"], 256
        ws = {}
        for name, cnt in (("H1a", B * H), ("H2a", B * H), ("dZ2", B * H), ("dZ1", 2 * B * H), ("dq", B), ("q", B)):
            buf = torch.zeros(cnt, device="cuda:0")
            paired = C.c_int32(0)
            N.check(N.lib().pa_debug_workspace(pl._native.handle, name.encode(), buf.data_ptr(), cnt if name != "dZ1" else (2 * B * H if mode == "1" else B * H), C.byref(paired)))
            t = buf.cpu()
            if name == "dZ1":
                t = (t.view(B, H // 2, 2, 2)[:, :, 0, :] + t.view(B, H // 2, 2, 2)[:, :, 1, :]).reshape(B, H) if paired.value else t[:B * H].view(B, H)
            elif cnt == B * H:
                t = t.view(B, H)
            ws[name] = t
        out[mode] = ws
    for name in out["0"]:
        a, b = out["1"][name], out["0"][name]
        d = (a - b).abs()
        msg = f"   n={n} {name:4s} {float(d.max() / b.abs().max()):.3g}"
        if d.dim() == 2 and float(d.max()) > 1e-5 * float(b.abs().max()):
            bad = d > 1e-5 * b.abs().max()
            msg += f"  bad rows {torch.nonzero(bad.any(1)).flatten().tolist()[:12]} ({int(bad.any(1).sum())}) bad cols {torch.nonzero(bad.any(0)).flatten().tolist()[:12]} ({int(bad.any(0).sum())})"
        print(msg)
