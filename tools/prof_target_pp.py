#!/usr/bin/env python3
"""Phase timeline of target_pp_kernel (ping-pong teams): stamps of the first 8 phases per wave.
   PEARL_AMD_OVERLAP=0 python tools/prof_target_pp.py [rounds]"""
import os, random, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("PEARL_AMD_OVERLAP", "0")
import bench  # noqa: E402
from pearl_amd import (BasicReplayBuffer, DeepQLearning, OneHotActionTensorRepresentationModule,  # noqa: E402
                       PearlAgent, _native as N)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    torch.manual_seed(0); random.seed(0)
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
    nwg = 512
    st = torch.zeros(nwg, 16, 32, dtype=torch.int64, device=dev)
    N.check(N.lib().pa_debug_set_prof_target(nat.handle, st.data_ptr(), 1 << 30))
    agent.learn()
    torch.cuda.synchronize()
    N.check(N.lib().pa_debug_set_prof_target(nat.handle, None, 0))
    s = st.cpu().numpy().astype(np.int64)
    live = s[:, 0, 0] > 0
    print("workgroups that ran:", int(live.sum()))
    s = s[live]
    t0 = s[s > 0].min()
    names = ["start", "after B1", "E done/after loop", "before B0"]
    for ph in range(8):
        for k in range(4):
            for team in (0, 1):
                v = s[:, team * 8:(team + 1) * 8, ph * 4 + k]
                v = v[v > 0]
                if v.size:
                    v = (v - t0) / 100.0
                    role = "EP" if (ph & 1) == team else "M "
                    print(f"phase {ph} team {team} {role} {names[k]:18s} min {v.min():8.2f} med {np.median(v):8.2f} max {v.max():8.2f}")


if __name__ == "__main__":
    main()
