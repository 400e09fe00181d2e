#!/usr/bin/env python3
"""Phase timeline of the fused row step (mlp_rowstep_kernel) at PPO's config-4 shapes from
in-kernel wall-clock stamps (pa_debug_rowstep_prof).

    python tools/prof_rowstep.py                 # B = 4096, S = 256, A = 16, [256, 256]
    PEARL_AMD_ROWSTEP_RT=1 python tools/prof_rowstep.py
"""
import os
import sys

import numpy as np
import torch
from torch import nn, optim

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearl_amd import _native as N  # noqa: E402
from pearl_amd.policy_learners.sequential_decision_making.flat_mlp import FlatMlp, layers_of  # noqa: E402

NAMES = {0: "start", 1: "L1 staged", 2: "L1 gemm", 13: "L1 epilogue", 3: "L2 staged", 4: "L2 gemm",
         14: "L2 epilogue", 5: "L3 staged", 6: "L3 gemm", 9: "forward done", 10: "head done",
         11: "backward done", 12: "ticket"}


def main():
    dev = torch.device("cuda", 0)
    B, S, A = int(os.environ.get("PROF_B", "4096")), 256, 16
    torch.manual_seed(0)
    da, dc = [S, 256, 256, A], [S, 256, 256, 1]
    an = [nn.Linear(da[i], da[i + 1]).to(dev) for i in range(3)]
    cn = [nn.Linear(dc[i], dc[i + 1]).to(dev) for i in range(3)]
    actor = FlatMlp(layers_of(an), optim.AdamW([p for l in an for p in l.parameters()], amsgrad=True), B).ensure(B)
    critic = FlatMlp(layers_of(cn), optim.AdamW([p for l in cn for p in l.parameters()], amsgrad=True), B).ensure(B)
    x = torch.randn(B, S, device=dev)
    arep = torch.nn.functional.one_hot(torch.randint(0, A, (B,), device=dev), A).float()
    p_old, gae, lam = torch.rand(B, device=dev) * 0.5 + 0.05, torch.randn(B, device=dev), torch.randn(B, device=dev)

    def step():
        FlatMlp.ppo_rowstep(actor, critic, x, arep, p_old, gae, 0.1, 0.01, lam, 2.0 / B)
        FlatMlp.adam_pair(actor, critic, None)

    for _ in range(20):
        step()
    torch.cuda.synchronize()
    stamps = torch.zeros(int(os.environ.get("PROF_WGS", "1024")), 8, 16, dtype=torch.int64, device=dev)
    dw = torch.zeros(1024, 8, 16, dtype=torch.int64, device=dev)
    if not os.environ.get("PROF_OFF"):
        N.check(N.lib().pa_debug_rowstep_prof(stamps.data_ptr()))
        N.check(N.lib().pa_debug_mlp_dw_prof(dw.data_ptr()))
    print("warm-up done", flush=True)
    step()
    torch.cuda.synchronize()
    N.check(N.lib().pa_debug_rowstep_prof(None))
    N.check(N.lib().pa_debug_mlp_dw_prof(None))
    st = stamps.cpu().numpy().astype(np.int64)
    print("highest workgroup slot written:", int(np.nonzero((st != 0).any(axis=(1, 2)))[0].max()))
    live = st[:, :, 0] > 0
    nwg = int(live.any(axis=1).sum())
    t0 = st[:, :, 0][live].min()
    print(f"== mlp_rowstep_kernel, B = {B}: {nwg} workgroups; us since the first wave started")
    print(f"{'phase':16s} {'min':>8s} {'median':>8s} {'max':>8s}")
    for i, name in NAMES.items():
        v = (st[:, :, i] - t0) / 100.0
        v = v[(st[:, :, i] > 0) & live]
        if v.size:
            print(f"{name:16s} {v.min():8.2f} {np.median(v):8.2f} {v.max():8.2f}")
    show_dw(dw)


def show_dw(dw):
    names = ["start", "setup", "main loop", "partials out", "stage-1 sum", "bar", "end"]
    st = dw.cpu().numpy().astype(np.int64)
    live = st[:, :, 0] > 0
    if not live.any():
        return
    t0 = st[:, :, 0][live].min()
    print(f"== weight_grad_kernel of the same step: {int(live.any(axis=1).sum())} workgroups; us since its first wave")
    print(f"{'phase':16s} {'min':>8s} {'median':>8s} {'max':>8s}")
    for i, name in enumerate(names):
        v = (st[:, :, i] - t0) / 100.0
        v = v[(st[:, :, i] > 0) & live]
        if v.size:
            print(f"{name:16s} {v.min():8.2f} {np.median(v):8.2f} {v.max():8.2f}")
    # the main loop per workgroup (slowest wave), by tile (blockIdx / ksplit, ksplit = 2 at B = 4096:
    # tiles 0-15 W1 actor, 16-31 W2 actor, 32-35 W3 actor, 36-.. critic), slice and XCD (blockIdx mod 8)
    ml = (st[:, :, 2].max(axis=1) - np.where(st[:, :, 1] > 0, st[:, :, 1], 1 << 62).min(axis=1)) / 100.0
    wgs = [i for i in range(st.shape[0]) if live[i].any()]
    ks = int(os.environ.get("PROF_KS", "2"))
    print("main loop us per workgroup (tile: slice0 slice1 ...):")
    for t in range((len(wgs) + ks - 1) // ks):
        row = [ml[t * ks + k] for k in range(ks) if t * ks + k < len(wgs)]
        print(f"  tile {t:3d}: " + " ".join(f"{v:6.2f}" for v in row))
    by_xcd = [[ml[i] for i in wgs if i % 8 == x] for x in range(8)]
    print("main loop us by XCD (mean / max):", ", ".join(f"{np.mean(v):.1f}/{np.max(v):.1f}" for v in by_xcd if v))
    # which workgroups are the late ones, and when did they START (a late start = it waited for a CU)
    end = (st[:, :, 6].max(axis=1) - t0) / 100.0
    start = (np.where(st[:, :, 0] > 0, st[:, :, 0], 1 << 62).min(axis=1) - t0) / 100.0
    late = [i for i in range(st.shape[0]) if live[i].any() and end[i] > np.median(end[live.any(axis=1)]) * 1.3]
    print("late workgroups (index: start -> end us):",
          ", ".join(f"{i}: {start[i]:.1f} -> {end[i]:.1f}" for i in late[:40]))


if __name__ == "__main__":
    main()
