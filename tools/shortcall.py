#!/usr/bin/env python3
"""Fixed cost of one DeepQLearning.learn() call (BASELINE config 2, one MI355X).

Real Pearl calls learn() with small `training_rounds` (often 1-10, policy_learner.py:162-195), so
what a call costs BESIDES its rounds matters as much as the steady-state round time.  For
rounds in {1, 10, 20, 100} this prints, averaged over `--calls` back-to-back calls:

  wall_us      perf_counter around learn() + torch.cuda.synchronize()
  host_us      time until learn() returned (it returns after its own single readback)
  enqueue_us   time inside pa_dqn_learn (host side of every launch of the call)
  per_round_us wall_us / rounds;   fixed_us = wall_us - rounds * steady (steady from the 100-round row)

    python tools/shortcall.py [--calls 30] [--rounds 1,10,20,100] [--trace]

--trace: three 20-round calls only (run it under rocprofv3 --kernel-trace, then
tools/rocpd_timeline.py --last-call).
"""
import argparse
import ctypes as C
import gc
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=30)
    ap.add_argument("--rounds", default="1,10,20,100")
    ap.add_argument("--replay", type=int, default=1_000_000)
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--profile", action="store_true",
                    help="cProfile of 300 one-round learn() calls (host side of the fixed cost)")
    args = ap.parse_args()
    from pearl_amd import (BasicReplayBuffer, DeepQLearning, OneHotActionTensorRepresentationModule,
                           PearlAgent, _native as N)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    random.seed(0)
    bench.N_REPLAY = args.replay
    pl = DeepQLearning(state_dim=bench.S, action_space=bench.space(bench.A), hidden_dims=bench.HIDDEN,
                       training_rounds=5, batch_size=bench.B,
                       action_representation_module=OneHotActionTensorRepresentationModule(bench.A))
    rb = BasicReplayBuffer(args.replay, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    bench.fill_arena(rb, dev, seed=0)
    agent.learn()
    torch.cuda.synchronize()

    lib = N.lib()
    enq = {"t": 0.0}
    real = lib.pa_dqn_learn

    def timed_learn(*a):
        t0 = time.perf_counter()
        rc = real(*a)
        enq["t"] += time.perf_counter() - t0
        return rc

    class _Lib:
        def __getattr__(self, k):
            return timed_learn if k == "pa_dqn_learn" else getattr(lib, k)

    N_lib = N.lib
    N.lib = lambda: _Lib()
    try:
        if args.profile:
            import cProfile
            import pstats
            pl._training_rounds = 1
            for _ in range(20):
                agent.learn()
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(300):
                agent.learn()
            pr.disable()
            torch.cuda.synchronize()
            pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
            return
        if args.trace:
            pl._training_rounds = 20
            for _ in range(3):
                agent.learn()
                torch.cuda.synchronize()
            return
        rows = []
        for r in [int(x) for x in args.rounds.split(",")]:
            pl._training_rounds = r
            for _ in range(3):
                agent.learn()
            torch.cuda.synchronize()
            enq["t"] = 0.0
            wall = host = 0.0
            gc.collect()
            gc.disable()     # like timeit: a generation-2 pass (~80 ms in a torch process) is not the call
            for _ in range(args.calls):
                t0 = time.perf_counter()
                agent.learn()
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                host += t1 - t0
                wall += t2 - t0
            gc.enable()
            n = args.calls
            rows.append({"rounds": r, "wall_us": 1e6 * wall / n, "host_us": 1e6 * host / n,
                         "enqueue_us": 1e6 * enq["t"] / n, "per_round_us": 1e6 * wall / n / r,
                         "transitions_per_s": bench.B * r * n / wall})
        steady = rows[-1]["per_round_us"]
        for row in rows:
            row["fixed_us_vs_longest_row"] = row["wall_us"] - row["rounds"] * steady
            print(json.dumps(row), flush=True)
    finally:
        N.lib = N_lib


if __name__ == "__main__":
    main()
