#!/usr/bin/env python3
"""Phase timeline of the bandit's row step (mlp_rowstep_kernel at config 5: 4096 contexts of 512
features, trunk [256, 64], one output) from in-kernel wall-clock stamps (pa_debug_rowstep_prof),
plus the step rate of learn_batch.

    python tools/prof_bandit_rowstep.py
    PEARL_AMD_ROWSTEP_RT=2 python tools/prof_bandit_rowstep.py      # 32 rows per workgroup
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pearl_amd import NeuralLinearBandit, TransitionBatch  # noqa: E402
from pearl_amd import _native as N  # noqa: E402

NAMES = {0: "start", 1: "L1 staged", 2: "L1 gemm", 13: "L1 epilogue", 3: "L2 staged", 4: "L2 gemm",
         14: "L2 epilogue", 5: "L3 staged", 6: "L3 gemm", 9: "forward done", 10: "head done",
         11: "backward done", 12: "ticket"}


def main():
    dev = torch.device("cuda", 0)
    F, B = 512, int(os.environ.get("PROF_B", "4096"))
    torch.manual_seed(0)
    pl = NeuralLinearBandit(feature_dim=F, hidden_dims=[256, 64], batch_size=B, learning_rate=1e-3)
    pl.to(dev)
    g = torch.Generator(device=dev).manual_seed(0)
    x = torch.randn(B, F, device=dev, generator=g)
    y = torch.rand(B, device=dev, generator=g)
    tb = TransitionBatch(state=x, action=torch.zeros(B, 1, device=dev), reward=y, weight=None)
    for _ in range(30):
        pl.learn_batch(tb)
    torch.cuda.synchronize()
    steps = 300
    t0 = time.perf_counter()
    for _ in range(steps):
        pl.learn_batch(tb)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"learn_batch: {1e6 * dt / steps:.1f} us per step, {B * steps / dt / 1e6:.2f} M contexts/s "
          f"(PEARL_AMD_ROWSTEP_RT={os.environ.get('PEARL_AMD_ROWSTEP_RT', 'default')})", flush=True)
    stamps = torch.zeros(1024, 8, 16, dtype=torch.int64, device=dev)
    N.check(N.lib().pa_debug_rowstep_prof(stamps.data_ptr()))
    pl.learn_batch(tb)
    torch.cuda.synchronize()
    N.check(N.lib().pa_debug_rowstep_prof(None))
    st = stamps.cpu().numpy().astype(np.int64)
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


if __name__ == "__main__":
    main()
