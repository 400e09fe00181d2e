#!/usr/bin/env python3
"""Phase timeline of the two online-chain kernels (online_rowpass_kernel, weight_grad_kernel)
from in-kernel wall-clock stamps (pa_debug_set_prof).  Runs the cfg2 learner for a few learn()
calls, the last launch's stamps are what is printed.

    python tools/prof_chain.py            # overlapped loop (default)
    PEARL_AMD_OVERLAP=0 python tools/prof_chain.py
"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pearl_amd import (BasicReplayBuffer, DeepQLearning, OneHotActionTensorRepresentationModule,  # noqa: E402
                       PearlAgent, _native as N)

ROW_NAMES = ["start", "loads issued", "L1 done", "h1 stored", "barA", "L2 done", "s2/head part",
             "barB", "G=s2 W2 done", "y consumed", "end", "peer q consumed"]
DW_NAMES = ["start", "setup", "main loop", "partials out", "stage-1 sum", "bar", "end"]


def show(title, stamps, names, nwg):
    st = stamps[:nwg].cpu().numpy().astype(np.int64)     # [wg][wave][16]
    t0 = st[:, :, 0][st[:, :, 0] > 0].min()
    print(f"== {title}: {nwg} workgroups; us since the first wave started (10 ns ticks)")
    print(f"{'phase':16s} {'min':>8s} {'median':>8s} {'max':>8s}")
    for i, n in enumerate(names):
        v = (st[:, :, i] - t0) / 100.0
        v = v[st[:, :, i] > 0]
        if v.size:
            print(f"{n:16s} {v.min():8.2f} {np.median(v):8.2f} {v.max():8.2f}")
    # effective shader clock: s_memtime ticks (slots 14, 15) over the wall time between the same
    # two points (first and last stamp of the wave)
    last = min(len(names) - 1, 10)
    ok = (st[:, :, 15] > 0) & (st[:, :, 14] > 0) & (st[:, :, last] > st[:, :, 0])
    if ok.any():
        ghz = (st[:, :, 15] - st[:, :, 14])[ok] / ((st[:, :, last] - st[:, :, 0])[ok] * 10.0)
        print(f"effective shader clock: median {np.median(ghz):.3f} GHz (min {ghz.min():.3f}, max {ghz.max():.3f})")
    if os.environ.get("PROF_PER_WG"):
        last = max(i for i in range(len(names)))
        for w in range(nwg):
            row = [(st[w, :, i].max() - t0) / 100.0 for i in range(len(names))]
            print(f"wg {w:3d}: " + " ".join(f"{x:6.2f}" for x in row))


def main():
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    random.seed(0)
    S, A, B = bench.S, bench.A, bench.B
    pl = DeepQLearning(state_dim=S, action_space=bench.space(A), hidden_dims=bench.HIDDEN,
                       training_rounds=40, batch_size=B,
                       action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = BasicReplayBuffer(200_000, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    bench.N_REPLAY = 200_000
    bench.fill_arena(rb, dev, seed=0)
    agent.learn()
    nat = pl._ensure_bound(B, A)
    row = torch.zeros(128, 8, 16, dtype=torch.int64, device=dev)
    dw = torch.zeros(128, 8, 16, dtype=torch.int64, device=dev)
    # training_steps = 40 -> windows start at rounds 0, 8, 18, ...: round 13 is mid-window, its
    # chain runs beside the persistent target pass
    rnd = int(os.environ.get("PROF_ROUND", "13"))
    N.check(N.lib().pa_debug_set_prof(nat.handle, row.data_ptr(), dw.data_ptr(), rnd))
    pl._training_rounds = 40
    agent.learn()
    torch.cuda.synchronize()
    N.check(N.lib().pa_debug_set_prof(nat.handle, None, None, -1))
    ndw = int(os.environ.get("PROF_DW_WGS", "112"))
    nrow = 128 if os.environ.get("PEARL_AMD_ROWPASS_PAIR", "0") == "1" else 64
    r = row[:nrow].cpu().numpy().astype(np.int64)
    d = dw[:ndw].cpu().numpy().astype(np.int64)
    row_end = r[:, :, 10].max()
    dw_start = d[:, :, 0][d[:, :, 0] > 0]
    print(f"kernel boundary inside the round (same 100 MHz clock): last row-pass wave ended -> first "
          f"weight-gradient wave started {(dw_start.min() - row_end) / 100.0:.2f} us, -> median wave "
          f"{(np.median(dw_start) - row_end) / 100.0:.2f} us, -> last {(dw_start.max() - row_end) / 100.0:.2f} us")
    show("online_rowpass_kernel" + ("" if nrow == 64 else " (paired: 2 workgroups per tile)"), row, ROW_NAMES, nrow)
    show("weight_grad_kernel", dw, DW_NAMES, int(os.environ.get("PROF_DW_WGS", "112")))


if __name__ == "__main__":
    main()
