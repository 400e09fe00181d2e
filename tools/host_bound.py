"""Host- or device-bound?  For one learner, times `steps` rounds of learn() twice: wall time until the
last launch is enqueued (host) and until the device is idle; then a cProfile of the host side.
Usage: python tools/host_bound.py [sac|ppo] [steps]"""
import cProfile
import os
import pstats
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

DEV = "cuda:0"


def make_sac(steps):
    from pearl_amd import BasicReplayBuffer, BoxActionSpace, ContinuousSoftActorCritic, PearlAgent
    S, A, B, N = 64, 8, 1024, 200_000
    pl = ContinuousSoftActorCritic(action_space=BoxActionSpace(-torch.ones(A), torch.ones(A)), state_dim=S,
                                   actor_hidden_dims=[256, 256], critic_hidden_dims=[256, 256],
                                   batch_size=B, training_rounds=steps)
    rb = BasicReplayBuffer(N, sampler="device")
    PearlAgent(pl, replay_buffer=rb, device_id=0)
    st = torch.randn(N + 1, S, device=DEV)
    ids = torch.arange(N, device=DEV)
    rb.push_many(state=st[:-1], action=torch.rand(N, A, device=DEV) * 2 - 1, reward=(ids % 7).float(),
                 terminated=(ids % 50 == 0), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
                 next_state=st[1:])
    return lambda: pl.learn(rb)


def make_td3(steps):
    from pearl_amd import TD3, BasicReplayBuffer, BoxActionSpace, PearlAgent
    S, A, B, N = 64, 8, 1024, 200_000
    pl = TD3(action_space=BoxActionSpace(-torch.ones(A), torch.ones(A)), state_dim=S,
             actor_hidden_dims=[256, 256], critic_hidden_dims=[256, 256], batch_size=B,
             training_rounds=steps)
    rb = BasicReplayBuffer(N, sampler="device")
    PearlAgent(pl, replay_buffer=rb, device_id=0)
    st = torch.randn(N + 1, S, device=DEV)
    ids = torch.arange(N, device=DEV)
    rb.push_many(state=st[:-1], action=torch.rand(N, A, device=DEV) * 2 - 1, reward=(ids % 7).float(),
                 terminated=(ids % 50 == 0), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
                 next_state=st[1:])
    return lambda: pl.learn(rb)


def make_ppo(steps):
    from pearl_amd import (DiscreteActionSpace, OneHotActionTensorRepresentationModule, PearlAgent,
                           PPOReplayBuffer, ProximalPolicyOptimization)
    S, A, B, N = 256, 16, 4096, 65_536
    sp = DiscreteActionSpace([torch.tensor([k]) for k in range(A)])
    pl = ProximalPolicyOptimization(action_space=sp, state_dim=S, actor_hidden_dims=[256, 256],
                                    critic_hidden_dims=[256, 256], training_rounds=steps, batch_size=B,
                                    epsilon=0.1,
                                    action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = PPOReplayBuffer(N, sampler="device")
    PearlAgent(pl, replay_buffer=rb, device_id=0)
    st = torch.randn(N + 1, S, device=DEV)
    ids = torch.arange(N, device=DEV)
    rb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 7).float(),
                 terminated=(ids % 500 == 499), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
                 next_state=st[1:], curr_available_actions=sp, next_available_actions=sp,
                 max_number_actions=A)
    pl.preprocess_replay_buffer(rb)
    if "full" in sys.argv:
        return lambda: pl.learn(rb)                  # with preprocess_replay_buffer, as agent.learn()
    from pearl_amd.policy_learners.sequential_decision_making.actor_critic_base import ActorCriticBase
    return lambda: ActorCriticBase.learn(pl, rb)     # learn() without the one-off rollout pass


def _discrete_buffer(S, A, N):
    from pearl_amd import BasicReplayBuffer, DiscreteActionSpace
    sp = DiscreteActionSpace([torch.tensor([k]) for k in range(A)])
    rb = BasicReplayBuffer(N, sampler="device")
    return sp, rb


def _fill_discrete(rb, sp, S, A, N):
    st = torch.randn(N + 1, S, device=DEV)
    ids = torch.arange(N, device=DEV)
    rb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 7).float(),
                 terminated=(ids % 50 == 0), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
                 next_state=st[1:], curr_available_actions=sp, next_available_actions=sp,
                 max_number_actions=A)


def make_dsac(steps):
    from pearl_amd import OneHotActionTensorRepresentationModule, PearlAgent, SoftActorCritic
    S, A, B, N = 128, 16, 1024, 200_000
    sp, rb = _discrete_buffer(S, A, N)
    pl = SoftActorCritic(action_space=sp, state_dim=S, actor_hidden_dims=[256, 256],
                         critic_hidden_dims=[256, 256], batch_size=B, training_rounds=steps,
                         action_representation_module=OneHotActionTensorRepresentationModule(A))
    PearlAgent(pl, replay_buffer=rb, device_id=0)
    _fill_discrete(rb, sp, S, A, N)
    return lambda: pl.learn(rb)


def make_ddqn(steps):
    from pearl_amd import DoubleDQN, OneHotActionTensorRepresentationModule, PearlAgent
    S, A, B, N = 128, 16, 1024, 200_000
    sp, rb = _discrete_buffer(S, A, N)
    pl = DoubleDQN(state_dim=S, action_space=sp, hidden_dims=[256, 256], training_rounds=steps,
                   batch_size=B, action_representation_module=OneHotActionTensorRepresentationModule(A))
    PearlAgent(pl, replay_buffer=rb, device_id=0)
    _fill_discrete(rb, sp, S, A, N)
    return lambda: pl.learn(rb)


def make_iql(steps):
    from pearl_amd import ImplicitQLearning, OneHotActionTensorRepresentationModule, PearlAgent
    S, A, B, N = 128, 16, 1024, 200_000
    sp, rb = _discrete_buffer(S, A, N)
    pl = ImplicitQLearning(action_space=sp, state_dim=S, actor_hidden_dims=[256, 256],
                           critic_hidden_dims=[256, 256], value_critic_hidden_dims=[256, 256],
                           batch_size=B, training_rounds=steps,
                           action_representation_module=OneHotActionTensorRepresentationModule(A))
    PearlAgent(pl, replay_buffer=rb, device_id=0)
    _fill_discrete(rb, sp, S, A, N)
    return lambda: pl.learn(rb)


def make_bandit(steps):
    from pearl_amd import NeuralLinearBandit, TransitionBatch
    F, B = 512, 4096
    pl = NeuralLinearBandit(feature_dim=F, hidden_dims=[256, 64], batch_size=B, learning_rate=1e-3)
    pl.to(DEV)
    tb = TransitionBatch(state=torch.randn(B, F, device=DEV), action=torch.zeros(B, 1, device=DEV),
                         reward=torch.rand(B, device=DEV), weight=None)

    def run():
        for _ in range(steps):
            pl.learn_batch(tb)
    return run


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "sac"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    if "threads32" in sys.argv:
        torch.set_num_threads(32)
    torch.manual_seed(0)
    random.seed(0)
    makers = {"sac": make_sac, "ppo": make_ppo, "td3": make_td3, "bandit": make_bandit,
              "dsac": make_dsac, "ddqn": make_ddqn, "iql": make_iql}
    learn = makers[which](steps)
    if "warm20" in sys.argv:          # a short warm-up instead of a full learn()
        short = makers[which](20)
        short()
    else:
        learn()
    torch.cuda.synchronize()
    # (a generation-2 pass of the cyclic collector is ~80 ms in a torch process and used to land
    # inside this call: 270 us per step of "host time" that is not the learner's)
    import gc
    gc.collect()
    gc.disable()
    t0 = time.perf_counter()
    learn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"{which}: host enqueue {1e6 * (t1 - t0) / steps:.0f} us/step, "
          f"device idle after {1e6 * (t2 - t0) / steps:.0f} us/step")
    pr = cProfile.Profile()
    pr.enable()
    learn()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(int(os.environ.get('TOPN', '16')))


if __name__ == "__main__":
    main()
