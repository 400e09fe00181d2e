"""Is ContinuousSoftActorCritic.learn() bound by the host or by the device?  Times `steps` rounds of
learn() twice: wall time until the last launch is enqueued (host), and until the device is idle."""
import os, sys, time, random
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
DEV = "cuda:0"


def main(steps=300):
    from pearl_amd import BasicReplayBuffer, BoxActionSpace, ContinuousSoftActorCritic, PearlAgent
    S, A, B, N = 64, 8, 1024, 200_000
    torch.manual_seed(0); random.seed(0)
    pl = ContinuousSoftActorCritic(action_space=BoxActionSpace(-torch.ones(A), torch.ones(A)), state_dim=S,
                                   actor_hidden_dims=[256, 256], critic_hidden_dims=[256, 256],
                                   batch_size=B, training_rounds=steps)
    rb = BasicReplayBuffer(N, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    st = torch.randn(N + 1, S, device=DEV)
    ids = torch.arange(N, device=DEV)
    rb.push_many(state=st[:-1], action=torch.rand(N, A, device=DEV) * 2 - 1, reward=(ids % 7).float(),
                 terminated=(ids % 50 == 0), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
                 next_state=st[1:])
    agent.learn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    agent.learn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"SAC: host enqueue {1e6*(t1-t0)/steps:.0f} us/step, device-idle {1e6*(t2-t0)/steps:.0f} us/step")
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); agent.learn(); pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
