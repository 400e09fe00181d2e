#!/usr/bin/env python3
"""Where the host time of a short DeepQLearning.learn() call goes (one MI355X): per-helper cost of
the Python side on the BOUND learner (CUDA tensors), and the split of a 1-round / 20-round call into
python-before, enqueue (inside pa_dqn_learn), wait-for-device and python-after."""
import gc
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    from pearl_amd import (BasicReplayBuffer, DeepQLearning, OneHotActionTensorRepresentationModule,
                           PearlAgent, _native as N)
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    random.seed(0)
    bench.N_REPLAY = 200_000
    pl = DeepQLearning(state_dim=bench.S, action_space=bench.space(bench.A), hidden_dims=bench.HIDDEN,
                       training_rounds=5, batch_size=bench.B,
                       action_representation_module=OneHotActionTensorRepresentationModule(bench.A))
    rb = BasicReplayBuffer(bench.N_REPLAY, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    bench.fill_arena(rb, dev, seed=0)
    agent.learn()
    torch.cuda.synchronize()

    def t(name, fn, n=3000):
        fn()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        print(f"{name:34s} {(time.perf_counter() - t0) / n * 1e6:8.2f} us", flush=True)

    t("_signature", pl._signature)
    t("_param_pairs", pl._param_pairs)
    t("_adam_steps", pl._adam_steps)
    t("_set_adam_steps", lambda: pl._set_adam_steps(pl._adam_steps()))
    t("_dims", pl._dims)
    t("_ensure_bound", lambda: pl._ensure_bound(bench.B, bench.A))
    t("_arena_path_ok", lambda: pl._arena_path_ok(rb))
    t("len(rb)", lambda: len(rb))
    t("versions", lambda: tuple(v for pq, pt in pl._param_pairs() for v in (pq._version, pt._version)))
    t("N.stream_ptr", lambda: N.stream_ptr(dev))
    t("current_stream().synchronize (idle)", lambda: torch.cuda.current_stream(dev).synchronize())
    nat = pl._native
    t("loss_host[:20].tolist()", lambda: nat.loss_host[:20].tolist())
    t("pa_dqn_check", lambda: N.lib().pa_dqn_check(nat.handle))
    t("random.getrandbits", lambda: random.getrandbits(64))
    # the call itself
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
    for r in (1, 20):
        pl._training_rounds = r
        for _ in range(5):
            agent.learn()
        torch.cuda.synchronize()
        gc.collect()
        gc.disable()
        enq["t"] = 0.0
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            agent.learn()
        dt = (time.perf_counter() - t0) / n * 1e6
        gc.enable()
        print(f"rounds {r:3d}: call {dt:8.1f} us, inside pa_dqn_learn {enq['t'] / n * 1e6:7.1f} us", flush=True)
    N.lib = N_lib


if __name__ == "__main__":
    main()
