#!/usr/bin/env python3
"""cpu_baseline.kind == "reference": the REAL reference (facebookresearch/Pearl) timed on the host
cores — `PearlAgent.learn()` (pearl/pearl_agent.py:213-220 -> policy_learner.py:162-195:
sample + preprocess_batch + learn_batch per round) of BASELINE.json config 2 on a bounded replay,
and (`--config sac,ppo,bandit`) the learners of configs 3 / 4 / 5 the same way.

TEST / MEASUREMENT INFRASTRUCTURE ONLY.  Executed as a child process by bench.py's cpu_baseline
leg with no GPU visible (HIP_VISIBLE_DEVICES= / CUDA_VISIBLE_DEVICES= empty), because the reference
picks `cuda:{rank}` whenever torch sees a device (pearl/utils/device.py:48-59).  The reference is
imported from /root/reference (build container) or oracle/_ref (staged by oracle/stage_ref.sh; the
copy that travels to the GPU box).  Prints ONE JSON object.
"""
import argparse
import json
import os
import random
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))


def reference_root():
    for cand in (os.environ.get("PEARL_REFERENCE"), "/root/reference", os.path.join(HERE, "_ref")):
        if cand and os.path.isdir(os.path.join(cand, "pearl")):
            return cand
    return None


def _prep(args):
    root = reference_root()
    if root is None:
        print(json.dumps({"error": "reference not found (neither /root/reference nor oracle/_ref)"}))
        return None
    sys.path[:0] = [os.path.join(HERE, "gymstub"), root]
    import torch
    assert not torch.cuda.is_available(), "the reference baseline must run with no GPU visible"
    torch.set_num_threads(max(1, min(args.threads, os.cpu_count() or 1)))
    return root


def _loop(fn, seconds, per_call):
    """>= 1 warm-up call, then calls until `seconds` have passed: (units, dt)."""
    fn()
    units, t0 = 0, time.perf_counter()
    while True:
        units += per_call(fn())
        dt = time.perf_counter() - t0
        if dt >= seconds:
            return units, dt


def dqn(args, root):
    import torch
    from pearl.action_representation_modules.one_hot_action_representation_module import (
        OneHotActionTensorRepresentationModule)
    from pearl.pearl_agent import PearlAgent
    from pearl.policy_learners.sequential_decision_making.deep_q_learning import DeepQLearning
    from pearl.replay_buffers.basic_replay_buffer import BasicReplayBuffer
    from pearl.utils.instantiations.spaces.discrete_action import DiscreteActionSpace

    S, A, B, n = 128, 16, 1024, args.replay
    torch.manual_seed(0)
    random.seed(0)
    space = DiscreteActionSpace([torch.tensor([k]) for k in range(A)])
    pl = DeepQLearning(state_dim=S, action_space=space, hidden_dims=[256, 256],
                       training_rounds=args.rounds, batch_size=B,
                       action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = BasicReplayBuffer(n)
    agent = PearlAgent(policy_learner=pl, replay_buffer=rb, device_id=-1)
    states = torch.randn(n + 1, S)
    t0 = time.perf_counter()
    for i in range(n):
        rb.push(state=states[i], action=torch.tensor([i % A]), reward=float(i % 7),
                terminated=(i % 50 == 0), truncated=False, curr_available_actions=space,
                next_state=states[i + 1], next_available_actions=space, max_number_actions=A)
    fill_s = time.perf_counter() - t0
    agent.learn()                      # warm-up (>= 3 rounds)
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.seconds:
        report = agent.learn()
        steps += len(report["loss"])
    dt = time.perf_counter() - t0
    return {
        "value": B * steps / dt, "unit": "transitions/s", "cores": torch.get_num_threads(),
        "kind": "reference",
        "sample": f"{steps} rounds of the reference's own PearlAgent.learn() (sample + preprocess + "
                  f"learn_batch, B={B}, torch {torch.__version__} CPU) on a {n}-entry "
                  f"BasicReplayBuffer, {dt:.1f}s; fill {n / fill_s:.0f} push/s; "
                  f"os.cpu_count()={os.cpu_count()}; reference from {os.path.basename(root)}",
        "final_loss": float(report["loss"][-1])}


def sac(args, root):
    """BASELINE config 3: the reference's ContinuousSoftActorCritic (soft_actor_critic_continuous.py
    :131-231) through PearlAgent.learn() — sample + preprocess_batch + learn_batch per round."""
    import torch
    from pearl.pearl_agent import PearlAgent
    from pearl.policy_learners.sequential_decision_making.soft_actor_critic_continuous import (
        ContinuousSoftActorCritic)
    from pearl.replay_buffers.basic_replay_buffer import BasicReplayBuffer
    from pearl.utils.instantiations.spaces.box_action import BoxActionSpace

    S, A, B, n = 64, 8, 1024, min(args.replay, 8192)
    torch.manual_seed(0)
    random.seed(0)
    pl = ContinuousSoftActorCritic(action_space=BoxActionSpace(-torch.ones(A), torch.ones(A)),
                                   state_dim=S, actor_hidden_dims=[256, 256],
                                   critic_hidden_dims=[256, 256], batch_size=B, training_rounds=5)
    rb = BasicReplayBuffer(n)
    agent = PearlAgent(policy_learner=pl, replay_buffer=rb, device_id=-1)
    states = torch.randn(n + 1, S)
    acts = torch.rand(n, A) * 2 - 1
    for i in range(n):
        rb.push(state=states[i], action=acts[i], reward=float(i % 7), terminated=(i % 50 == 0),
                truncated=False, next_state=states[i + 1])
    steps, dt = _loop(agent.learn, args.seconds, lambda rep: len(rep["critic_loss"]))
    return {"value": B * steps / dt, "unit": "transitions/s", "cores": torch.get_num_threads(),
            "kind": "reference",
            "sample": f"{steps} rounds of the reference's PearlAgent.learn() with "
                      f"ContinuousSoftActorCritic (S=64, A=8, [256,256], B={B}) on a {n}-entry "
                      f"BasicReplayBuffer, {dt:.1f}s"}


def ppo(args, root):
    """BASELINE config 4: the reference's ProximalPolicyOptimization.learn() (ppo.py:152-293):
    preprocess_replay_buffer (GAE loop over the rollout) + training_rounds minibatch steps of 4096.
    The rollout is bounded (8192 transitions instead of 65 536: the reference pushes at ~3.5 k/s
    and its GAE loop runs at ~12 k transitions/s); both parts are reported."""
    import torch
    from pearl.action_representation_modules.one_hot_action_representation_module import (
        OneHotActionTensorRepresentationModule)
    from pearl.pearl_agent import PearlAgent
    from pearl.policy_learners.sequential_decision_making.ppo import (PPOReplayBuffer,
                                                                      ProximalPolicyOptimization)
    from pearl.utils.instantiations.spaces.discrete_action import DiscreteActionSpace

    S, A, B, n = 256, 16, 4096, 8192
    torch.manual_seed(0)
    random.seed(0)
    space = DiscreteActionSpace([torch.tensor([k]) for k in range(A)])
    rounds = 4
    pl = ProximalPolicyOptimization(action_space=space, state_dim=S, actor_hidden_dims=[256, 256],
                                    critic_hidden_dims=[256, 256], training_rounds=rounds,
                                    batch_size=B, epsilon=0.1,
                                    action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = PPOReplayBuffer(n)
    PearlAgent(policy_learner=pl, replay_buffer=rb, device_id=-1)
    states = torch.randn(n + 1, S)
    for i in range(n):
        rb.push(state=states[i], action=torch.tensor([i % A]), reward=float(i % 7),
                terminated=(i % 500 == 499), truncated=False, curr_available_actions=space,
                next_state=states[i + 1], next_available_actions=space, max_number_actions=A)
    t0 = time.perf_counter()
    pl.preprocess_replay_buffer(rb)
    pre = time.perf_counter() - t0
    # the minibatch steps alone (policy_learner.py:162-195 without the per-call preprocessing):
    # sample + preprocess_batch + learn_batch, as PolicyLearner.learn runs them
    from pearl.policy_learners.policy_learner import PolicyLearner
    steps, dt = _loop(lambda: PolicyLearner.learn(pl, rb), args.seconds,
                      lambda rep: len(rep["critic_loss"]))
    return {"value": B * steps / dt, "unit": "transitions/s", "cores": torch.get_num_threads(),
            "kind": "reference", "preprocess_transitions_per_s": n / pre,
            "sample": f"{steps} minibatch rounds (sample + preprocess_batch + learn_batch, B={B}) of "
                      f"the reference's ProximalPolicyOptimization on a {n}-transition "
                      f"PPOReplayBuffer, {dt:.1f}s; preprocess_replay_buffer of the {n} transitions "
                      f"{pre:.2f}s"}


def bandit(args, root):
    """BASELINE config 5: the reference's NeuralLinearBandit.learn_batch (neural_linear_bandit.py
    :159-225) on 4096 contexts of 512 features, trunk [256, 64]."""
    import torch
    from pearl.policy_learners.contextual_bandits.neural_linear_bandit import NeuralLinearBandit
    from pearl.replay_buffers.transition import TransitionBatch

    F, B = 512, 4096
    torch.manual_seed(0)
    pl = NeuralLinearBandit(feature_dim=F, hidden_dims=[256, 64], batch_size=B, learning_rate=1e-3)
    tb = TransitionBatch(state=torch.randn(B, F), action=torch.zeros(B, 1), reward=torch.rand(B),
                         weight=None)
    steps, dt = _loop(lambda: pl.learn_batch(tb), args.seconds, lambda rep: 1)
    return {"value": B * steps / dt, "unit": "contexts/s", "cores": torch.get_num_threads(),
            "kind": "reference",
            "sample": f"{steps} calls of the reference's NeuralLinearBandit.learn_batch (F=512, "
                      f"[256,64], B={B}), {dt:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--replay", type=int, default=50_000)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--config", default="dqn",
                    help="dqn (BASELINE configs[1]) | sac | ppo | bandit, or a comma list: one JSON "
                         "line per config, each tagged with its name")
    args = ap.parse_args()
    root = _prep(args)
    if root is None:
        return 2
    fns = {"dqn": dqn, "sac": sac, "ppo": ppo, "bandit": bandit}
    for name in args.config.split(","):
        out = fns[name](args, root)
        out["config"] = name
        print(json.dumps(out), flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
