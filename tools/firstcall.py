"""Is the driver's bench (5 warm-up rounds, then ONE timed 20-round learn()) slower than the same
call in steady state?  Times consecutive 20-round calls after a 5-round warm-up, like bench.py."""
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
    warm = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    pl = DeepQLearning(state_dim=bench.S, action_space=bench.space(bench.A), hidden_dims=bench.HIDDEN,
                       training_rounds=warm, batch_size=bench.B,
                       action_representation_module=OneHotActionTensorRepresentationModule(bench.A))
    rb = BasicReplayBuffer(bench.N_REPLAY, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    bench.fill_arena(rb, dev, seed=0)
    agent.learn()
    pl._training_rounds = 20
    nat = pl._ensure_bound(bench.B, bench.A)
    N.check(N.lib().pa_dqn_enable_timing(nat.handle, 1))
    out = []
    for i in range(6):
        torch.cuda.synchronize(dev)
        if i == 3:
            time.sleep(0.5)          # an idle gap, like the one before a driver's timed region
            torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        agent.learn()
        torch.cuda.synchronize(dev)
        out.append(round(1e6 * (time.perf_counter() - t0), 1))
    print(f"warmup {warm} rounds, then 20-round calls (us; call 3 follows a 0.5 s idle gap):", out)


if __name__ == "__main__":
    main()
