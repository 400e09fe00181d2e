#!/usr/bin/env python3
"""cpu_baseline.kind == "reference": the REAL reference (facebookresearch/Pearl) timed on the host
cores — `PearlAgent.learn()` (pearl/pearl_agent.py:213-220 -> policy_learner.py:162-195:
sample + preprocess_batch + learn_batch per round) of BASELINE.json config 2 on a bounded replay.

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--replay", type=int, default=50_000)
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=10)
    args = ap.parse_args()
    root = reference_root()
    if root is None:
        print(json.dumps({"error": "reference not found (neither /root/reference nor oracle/_ref)"}))
        return 2
    sys.path[:0] = [os.path.join(HERE, "gymstub"), root]
    import torch
    assert not torch.cuda.is_available(), "the reference baseline must run with no GPU visible"
    torch.set_num_threads(max(1, min(args.threads, os.cpu_count() or 1)))
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
    print(json.dumps({
        "value": B * steps / dt, "unit": "transitions/s", "cores": torch.get_num_threads(),
        "kind": "reference",
        "sample": f"{steps} rounds of the reference's own PearlAgent.learn() (sample + preprocess + "
                  f"learn_batch, B={B}, torch {torch.__version__} CPU) on a {n}-entry "
                  f"BasicReplayBuffer, {dt:.1f}s; fill {n / fill_s:.0f} push/s; "
                  f"os.cpu_count()={os.cpu_count()}; reference from {os.path.basename(root)}",
        "final_loss": float(report["loss"][-1])}))
    return 0


if __name__ == "__main__":
    sys.exit(main())
