#!/usr/bin/env python3
"""Environment-knob sweep of the DQN learn loop on ONE arena (BASELINE config 2): for every knob
setting a fresh learner (the knobs are read when its native handle is created), then the driver's
short call (warm-up 5, 20 rounds timed) and a long call (warm-up 200, 2000 rounds).

    python tools/sweep.py "PEARL_AMD_SPLIT_FIRST=3" "PEARL_AMD_SPLIT_FIRST=1" "A=1,B=2" ...
"""
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402


def timed(agent, pl, rounds):
    pl._training_rounds = rounds
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rep = agent.learn()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert len(rep["loss"]) == rounds and all(x == x for x in rep["loss"])
    return bench.B * rounds / dt, 1e6 * dt / rounds


def main():
    from pearl_amd import (BasicReplayBuffer, DeepQLearning, OneHotActionTensorRepresentationModule,
                           PearlAgent)
    dev = torch.device("cuda", 0)
    rb = BasicReplayBuffer(bench.N_REPLAY, sampler="device")
    rb.device_for_batches = dev
    rb._is_action_continuous = False
    bench.fill_arena(rb, dev, seed=0)
    base_env = dict(os.environ)
    for spec in sys.argv[1:] or [""]:
        os.environ.clear()
        os.environ.update(base_env)
        for kv in filter(None, ("" if spec in ("default", '""') else spec).split(",")):
            k, v = kv.split("=")
            os.environ[k] = v
        torch.manual_seed(0)
        random.seed(1000)
        pl = DeepQLearning(state_dim=bench.S, action_space=bench.space(bench.A),
                           hidden_dims=bench.HIDDEN, training_rounds=5, batch_size=bench.B,
                           action_representation_module=OneHotActionTensorRepresentationModule(bench.A))
        agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
        agent.learn()                                    # warm-up 5 (as the driver's command)
        short = [timed(agent, pl, 20) for _ in range(3)]
        pl._training_rounds = 200
        agent.learn()
        long_ = timed(agent, pl, 2000)
        print(json.dumps({"env": spec, "short_first_Mtps": round(short[0][0] / 1e6, 2),
                          "short_best_Mtps": round(max(s[0] for s in short) / 1e6, 2),
                          "short_us_per_round": [round(s[1], 1) for s in short],
                          "long_Mtps": round(long_[0] / 1e6, 2),
                          "long_us_per_round": round(long_[1], 2)}), flush=True)
        pl._native.close()


if __name__ == "__main__":
    main()
