#!/usr/bin/env python3
"""Where the host time of PPO's preprocess_replay_buffer goes (65 536-transition rollout, cfg4):
wall time of the call with and without a trailing sync, then a cProfile of 20 calls.
    python tools/prof_ppo_preprocess.py"""
import cProfile
import os
import pstats
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_algos as BA  # noqa: E402
from pearl_amd import (OneHotActionTensorRepresentationModule, PearlAgent, PPOReplayBuffer,  # noqa: E402
                       ProximalPolicyOptimization)

DEV = BA.DEV
S, A, B, N = 256, 16, 4096, 65_536
torch.manual_seed(0)
random.seed(1000)
pl = ProximalPolicyOptimization(action_space=BA.dspace(A), state_dim=S, actor_hidden_dims=[256, 256],
                                critic_hidden_dims=[256, 256], training_rounds=10, batch_size=B, epsilon=0.1,
                                action_representation_module=OneHotActionTensorRepresentationModule(A))
rb = PPOReplayBuffer(N, sampler="device")
PearlAgent(pl, replay_buffer=rb, device_id=DEV.index)
st = torch.randn(N + 1, S, device=DEV)
ids = torch.arange(N, device=DEV)
rb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 7).float(),
             terminated=(ids % 500 == 499), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
             next_state=st[1:], curr_available_actions=BA.dspace(A), next_available_actions=BA.dspace(A),
             max_number_actions=A)
for _ in range(3):
    pl.preprocess_replay_buffer(rb)
BA.sync()
import gc
gc.collect(); gc.disable()
for _ in range(3):
    t0 = time.perf_counter()
    pl.preprocess_replay_buffer(rb)
    t1 = time.perf_counter()
    BA.sync()
    t2 = time.perf_counter()
    print(f"host (enqueue) {1e6 * (t1 - t0):8.1f} us   + drain {1e6 * (t2 - t1):8.1f} us   = wall {1e6 * (t2 - t0):8.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    pl.preprocess_replay_buffer(rb)
    BA.sync()
pr.disable()
st_ = pstats.Stats(pr).sort_stats("cumulative")
st_.print_stats(28)
