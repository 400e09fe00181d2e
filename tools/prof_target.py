#!/usr/bin/env python3
"""Phase timeline of target_fused_kernel from in-kernel wall-clock stamps.

    PEARL_AMD_OVERLAP=0 python tools/prof_target.py [rounds_per_launch]
      1 round  = 256 tiles = one workgroup per CU (a tile's latency when it has the CU to itself)
      2 rounds = 512 tiles = two per CU, ...
"""
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PEARL_AMD_OVERLAP", "0")
import bench  # noqa: E402
from pearl_amd import (BasicReplayBuffer, DeepQLearning, OneHotActionTensorRepresentationModule,  # noqa: E402
                       PearlAgent, _native as N)

NAMES = ["start", "loads issued", "L1 mfma done", "h1 in LDS", "bar", "L2 done", "l3 partial",
         "bar", "end"]
NAMES = ["start", "loads issued", "L1+h1 in LDS", "bar1", "L2 done", "l3 partial", "bar2", "end"]


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    random.seed(0)
    S, A, B = bench.S, bench.A, bench.B
    pl = DeepQLearning(state_dim=S, action_space=bench.space(A), hidden_dims=bench.HIDDEN,
                       training_rounds=rounds, batch_size=B, target_update_freq=1000,
                       action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = BasicReplayBuffer(200_000, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    bench.N_REPLAY = 200_000
    bench.fill_arena(rb, dev, seed=0)
    agent.learn()
    nat = pl._ensure_bound(B, A)
    ntiles = rounds * B // 4
    st = torch.zeros(ntiles, 8, 16, dtype=torch.int64, device=dev)
    N.check(N.lib().pa_debug_set_prof_target(nat.handle, st.data_ptr(), ntiles))
    agent.learn()
    torch.cuda.synchronize()
    N.check(N.lib().pa_debug_set_prof_target(nat.handle, None, 0))
    s = st.cpu().numpy().astype(np.int64)
    t0 = s[:, :, 0][s[:, :, 0] > 0].min()
    print(f"== target_fused_kernel, {ntiles} tiles in one launch; us (10 ns ticks)")
    print(f"{'phase':16s} {'min':>8s} {'median':>8s} {'max':>8s}   per-tile duration since its start (median)")
    for i, n in enumerate(NAMES):
        v = (s[:, :, i] - t0) / 100.0
        rel = (s[:, :, i] - s[:, :, 0].min(axis=1, keepdims=True)) / 100.0
        print(f"{n:16s} {v.min():8.2f} {np.median(v):8.2f} {v.max():8.2f}   {np.median(rel):8.2f}")
    # residency: per CU, the time-average number of tiles in flight
    key = s[:, 0, 8]
    start = (s[:, :, 0].min(axis=1) - t0) / 100.0
    end = (s[:, :, 7].max(axis=1) - t0) / 100.0
    span = end.max()
    occ = []
    for k in np.unique(key):
        m = key == k
        occ.append(((end[m] - start[m]).sum() / span, m.sum()))
    occ = np.array(occ)
    print(f"launch span {span:.1f} us; CUs seen {len(occ)}; tiles in flight per CU: mean {occ[:, 0].mean():.2f} "
          f"min {occ[:, 0].min():.2f} max {occ[:, 0].max():.2f}; tiles per CU: min {occ[:, 1].min():.0f} "
          f"max {occ[:, 1].max():.0f}; mean tile latency {np.mean(end - start):.1f} us")


if __name__ == "__main__":
    main()
