#!/usr/bin/env python3
"""Throughput of the other hot-path rows of SURVEY.md §8 (a13-a16) on one MI355X, at the shapes of
BASELINE.json configs[2..4], each next to the reference-pinned CPU oracle on the host cores.

    python bench_algos.py [--steps K] [--cpu-seconds T]      # one JSON line per config

  cfg3  ContinuousSoftActorCritic  S=64, 8-dim action, twin-Q, [256,256], batch 1024
        (sample from a 200k HBM arena + preprocess + learn_batch, through PolicyLearner.learn)
  cfg4  PPO + GAE  S=256, 16 actions, rollout 65 536, minibatch 4096
        (preprocess_replay_buffer = action probabilities, values, GAE scan; then learn())
  cfg5  NeuralLinearBandit  512-dim contexts, [256,64] trunk, batch 4096 (learn_batch)

bench.py (the headline DQN metric) is the driver's contract; this script only documents that the
rows built after it run on the device and what they deliver.  The oracle is the checker / baseline
here, never the thing measured.
"""
import argparse
import gc
import json
import os
import random
import sys
import time

import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)
DEV = torch.device("cuda", 0)


def sync():
    torch.cuda.synchronize(DEV)


def dspace(n):
    from pearl_amd import DiscreteActionSpace
    return DiscreteActionSpace([torch.tensor([k]) for k in range(n)])


PEAK_F32_MFMA = 157.3e12   # dense fp32 MFMA, MI355X_MICROARCH.md
PEAK_SPLIT_MFMA = 2500e12 / 6   # fp32 results on the bf16 pipe: six bf16 products per fp32 product


def mlp_macs(dims):
    return sum(a * b for a, b in zip(dims[:-1], dims[1:]))


def step_roofline(flop_per_transition, transitions, seconds, kernel=None):
    """fp32-MFMA roofline of a learner step: algorithmic flops (2 per multiply-add of every layer
    GEMM the reference's forward / backward performs) over wall time."""
    ach = flop_per_transition * transitions / seconds
    out = {"bound": "mfma", "achieved": ach / 1e12, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
           "frac": ach / PEAK_F32_MFMA, "traffic": None, "flop_per_transition": flop_per_transition,
           "scope": "whole learner step (wall time of learn())"}
    if kernel:
        out["kernel"] = kernel
    return out


def engine_kernel_times(run_steps):
    """HIP-event durations of the MLP engine's two launches of a step — the fused row step
    (forward + loss head + backward, mlp_rowstep.hpp) and the weight gradients + AdamW
    (weight_grad_kernel) — over a few steps run for that purpose (pa_mlp_timing: an event record
    costs ~6 us of GPU idle, so these steps are not the ones `value` is taken from)."""
    import ctypes as C
    from pearl_amd import _native as N_
    N_.check(N_.lib().pa_mlp_timing(1))
    run_steps()
    sync()
    out = {}
    for which, name in ((0, "rowstep"), (1, "weight_grad")):
        us, n = C.c_double(), C.c_int64()
        N_.check(N_.lib().pa_mlp_timing_read(which, C.byref(us), C.byref(n)))
        if n.value:
            out[name] = {"avg_us": us.value, "launches": n.value}
    N_.check(N_.lib().pa_mlp_timing(0))
    return out


def engine_kernels(times, dims_list, rows, split_rowstep=False, split_dw=False):
    """Per-kernel roofline entries of an MLP-engine step from engine_kernel_times(): algorithmic
    FLOPs (2 per multiply-add; row step = forward of every layer + dX of layers >= 1, weight
    gradients = dW of every layer) of `rows` rows through the networks `dims_list`, over the
    launch duration, against the fp32-MFMA peak (`frac`, the contract figure) and against the rate
    of the pipe the kernel runs on (`frac_pipe`: the same peak for fp32 MFMA, 2.5 PF / 6 for the
    bf16x3 split form)."""
    f_row = 2 * sum(mlp_macs(d) + sum(a * b for a, b in zip(d[1:-1], d[2:])) for d in dims_list)
    f_dw = 2 * sum(mlp_macs(d) for d in dims_list)
    out = []
    for name, f, split, label in (("rowstep", f_row, split_rowstep, "mlp_rowstep_kernel (forward + head + backward)"),
                                  ("weight_grad", f_dw, split_dw, "weight_grad_kernel (dW + AdamW)")):
        if name not in times:
            continue
        rate = f * rows / (times[name]["avg_us"] * 1e-6)
        pipe = PEAK_SPLIT_MFMA if split else PEAK_F32_MFMA
        out.append({"kernel": label, "avg_launch_us": times[name]["avg_us"],
                    "launches_timed": times[name]["launches"], "flop_per_launch": f * rows,
                    "achieved": rate / 1e12, "unit": "TFLOP/s", "frac": rate / PEAK_F32_MFMA,
                    "frac_pipe": rate / pipe,
                    "pipe": ("bf16x3 split MFMA (2.5 PF / 6)" + (": forward layers; backward on fp32 MFMA"
                                                                   if name == "rowstep" else ""))
                    if split else "fp32 MFMA"})
    return out


def timed(fn, warm=1):
    """Wall time of one fn() call.  Like `timeit`, without the cyclic garbage collector inside the
    timed region: a generation-2 pass over a torch process's heap is ~80 ms — 800 SAC steps — and
    lands wherever the allocation counter happens to trip (measured: always inside the second
    learn() call of the SAC bench, tools/debug_loop.py)."""
    for _ in range(warm):
        fn()
    sync()
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        t0 = time.perf_counter()
        out = fn()
        sync()
        return time.perf_counter() - t0, out
    finally:
        if was:
            gc.enable()


def bench_sac(steps, cpu_seconds):
    from oracle.actor_critic_oracle import SacOracle
    from pearl_amd import BasicReplayBuffer, BoxActionSpace, ContinuousSoftActorCritic, PearlAgent
    S, A, B, N = 64, 8, 1024, 200_000
    torch.manual_seed(0)
    random.seed(0)
    low, high = -torch.ones(A), torch.ones(A)
    pl = ContinuousSoftActorCritic(action_space=BoxActionSpace(low, high), state_dim=S,
                                   actor_hidden_dims=[256, 256], critic_hidden_dims=[256, 256],
                                   batch_size=B, training_rounds=steps)
    rb = BasicReplayBuffer(N, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    g = torch.Generator(device=DEV).manual_seed(0)
    st = torch.randn(N + 1, S, device=DEV, generator=g)
    act = torch.rand(N, A, device=DEV, generator=g) * 2 - 1
    ids = torch.arange(N, device=DEV)
    rb.push_many(state=st[:-1], action=act, reward=(ids % 7).float(), terminated=(ids % 50 == 0),
                 truncated=torch.zeros(N, dtype=torch.bool, device=DEV), next_state=st[1:])
    import ctypes as C
    from pearl_amd import _native as N_
    agent.learn()                      # warm-up
    sync()
    N_.check(N_.lib().pa_sac_timing(1))
    dt, _ = timed(lambda: agent.learn(), warm=0)
    ua, ub, nt = C.c_double(), C.c_double(), C.c_int64()
    N_.check(N_.lib().pa_sac_timing_read(C.byref(ua), C.byref(ub), C.byref(nt)))
    N_.check(N_.lib().pa_sac_timing(0))
    gpu = B * steps / dt
    # algorithmic flops per transition (2 per multiply-add): what the reference's autograd does
    H = 256
    actor, critic = mlp_macs([S, H, H, 2 * A]), mlp_macs([S + A, H, H, 1])
    bwd_dx = lambda dims: sum(a * b for a, b in zip(dims[1:-1], dims[2:]))       # dX of layers >= 1
    f_actor_upd = 2 * (actor + 2 * critic + 2 * (bwd_dx([S + A, H, H, 1]) + H * A)   # critics: fwd + dx
                       + bwd_dx([S, H, H, 2 * A]) + actor)                           # actor: dX + dW
    f_critic_upd = 2 * (actor + 2 * critic                                           # targets
                        + 2 * (critic + bwd_dx([S + A, H, H, 1]) + critic))          # online: fwd, dX, dW
    flop_step = f_actor_upd + f_critic_upd
    # the dominant kernel, sac_rows_a (actor rows || both critics at (s, a_batch)): its own flops
    f_rows_a = 2 * (actor + 2 * (critic + H * H + H * A) + 2 * A * H + H * H       # actor rows
                    + 2 * (critic + H * H))                                         # critic rows
    roof = step_roofline(flop_step, B * steps, dt)
    if nt.value:
        ka = f_rows_a * B / (ua.value * 1e-6)
        roof.update({"kernel": "sac_rows_a_kernel<16,4,5,0,SPLIT> (actor-update rows || helper: critic 2 at the fresh action || critics at (s, a_batch))",
                     "achieved": ka / 1e12, "frac": ka / PEAK_F32_MFMA, "avg_launch_us": ua.value,
                     "flop_per_launch": f_rows_a * B, "launches_timed": nt.value,
                     "scope": "dominant kernel, HIP events on the learner stream inside the timed learn()",
                     "other_kernels_us": {"sac_rows_b_kernel": ub.value},
                     "step": {"achieved": flop_step * B * steps / dt / 1e12,
                              "frac": flop_step * B * steps / dt / PEAK_F32_MFMA,
                              "flop_per_transition": flop_step}})
    # CPU oracle on the same shapes (one fixed batch: the oracle has no replay of its own)
    orc = SacOracle({k: v.cpu() for k, v in pl._actor.state_dict().items()},
                    {k: v.cpu() for k, v in pl._critic.state_dict().items()},
                    {k: v.cpu() for k, v in pl._critic_target.state_dict().items()}, low, high)
    batch = dict(state=torch.randn(B, S), action=torch.rand(B, A) * 2 - 1, reward=torch.rand(B),
                 terminated=torch.zeros(B, dtype=torch.bool), next_state=torch.randn(B, S))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < cpu_seconds:
        orc.learn_batch(batch, torch.randn(B, A), torch.randn(B, A))
        n += 1
    cpu = B * n / (time.perf_counter() - t0)
    return {"config": "cfg3 ContinuousSoftActorCritic S=64 A=8 twin-Q [256,256] B=1024 replay 200k",
            "metric": "learner transitions/s through PolicyLearner.learn (sample+preprocess+learn_batch)",
            "value": gpu, "steps": steps, "ms_per_step": 1e3 * dt / steps, "roofline": roof,
            "cpu_baseline": {"value": cpu, "kind": "port", "cores": torch.get_num_threads(),
                             "sample": f"{n} oracle learn_batch calls on one batch (no sampling cost)"}}


def bench_td3(steps, cpu_seconds):
    """TD3 at config 3's shapes: deterministic tanh actor + target, twin critics +
    targets, actor / target updates every second round, target policy smoothing."""
    from oracle.actor_critic_oracle import DdpgOracle
    from pearl_amd import TD3, BasicReplayBuffer, BoxActionSpace, PearlAgent
    S, A, B, N = 64, 8, 1024, 200_000
    torch.manual_seed(0)
    random.seed(0)
    low, high = -torch.ones(A), torch.ones(A)
    pl = TD3(action_space=BoxActionSpace(low, high), state_dim=S, actor_hidden_dims=[256, 256],
             critic_hidden_dims=[256, 256], batch_size=B, training_rounds=steps)
    rb = BasicReplayBuffer(N, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    g = torch.Generator(device=DEV).manual_seed(0)
    st = torch.randn(N + 1, S, device=DEV, generator=g)
    ids = torch.arange(N, device=DEV)
    rb.push_many(state=st[:-1], action=torch.rand(N, A, device=DEV, generator=g) * 2 - 1,
                 reward=(ids % 7).float(), terminated=(ids % 50 == 0),
                 truncated=torch.zeros(N, dtype=torch.bool, device=DEV), next_state=st[1:])
    dt, _ = timed(lambda: agent.learn())
    gpu = B * steps / dt
    sd = lambda m: {k: v.cpu() for k, v in m.state_dict().items()}
    orc = DdpgOracle(sd(pl._actor), sd(pl._actor_target), sd(pl._critic), sd(pl._critic_target),
                     low, high, td3=True)
    batch = dict(state=torch.randn(B, S), action=torch.rand(B, A) * 2 - 1, reward=torch.rand(B),
                 terminated=torch.zeros(B, dtype=torch.bool), next_state=torch.randn(B, S))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < cpu_seconds:
        orc.training_steps = n
        orc.learn_batch(batch, 0.2 * torch.randn(B, A))
        n += 1
    cpu = B * n / (time.perf_counter() - t0)
    return {"config": "cfg3 shapes, TD3 S=64 A=8 twin-Q [256,256] B=1024 replay 200k",
            "metric": "learner transitions/s through PolicyLearner.learn (sample+preprocess+learn_batch)",
            "value": gpu, "steps": steps, "ms_per_step": 1e3 * dt / steps,
            # critic update every step (target policy + 2 target critics forward; 2 online critics
            # forward, dX, dW), actor update every second step (actor forward / dX / dW, critic 1
            # forward + dX)
            "roofline": step_roofline(
                2 * (mlp_macs([S, 256, 256, A]) + 2 * mlp_macs([S + A, 256, 256, 1])
                     + 2 * (2 * mlp_macs([S + A, 256, 256, 1]) + 256 * 256)
                     + 0.5 * (2 * mlp_macs([S, 256, 256, A]) + 256 * 256 + 256 * A
                              + mlp_macs([S + A, 256, 256, 1]) + 256 * 256 + 256 * (S + A))),
                B * steps, dt, kernel="sac_rows_a/b_kernel<16,4,5,1> + weight_grad_kernel (fused rows, "
                                      "deterministic-policy head)"),
            "cpu_baseline": {"value": cpu, "kind": "port", "cores": torch.get_num_threads(),
                             "sample": f"{n} oracle learn_batch calls on one batch (no sampling cost)"}}


def bench_dsac(steps, cpu_seconds):
    """Discrete SoftActorCritic at config 2's shapes (128-dim states, 16 actions,
    [256,256], batch 1024): the twin critics see 16 384 (state, action) rows per pass."""
    from oracle.actor_critic_oracle import DiscreteSacOracle
    from pearl_amd import (BasicReplayBuffer, OneHotActionTensorRepresentationModule, PearlAgent,
                           SoftActorCritic)
    S, A, B, N = 128, 16, 1024, 200_000
    torch.manual_seed(0)
    random.seed(0)
    pl = SoftActorCritic(action_space=dspace(A), state_dim=S, actor_hidden_dims=[256, 256],
                         critic_hidden_dims=[256, 256], batch_size=B, training_rounds=steps,
                         action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = BasicReplayBuffer(N, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    g = torch.Generator(device=DEV).manual_seed(0)
    st = torch.randn(N + 1, S, device=DEV, generator=g)
    ids = torch.arange(N, device=DEV)
    rb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 7).float(),
                 terminated=(ids % 50 == 0), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
                 next_state=st[1:], curr_available_actions=dspace(A),
                 next_available_actions=dspace(A), max_number_actions=A)
    dt, _ = timed(lambda: agent.learn())
    gpu = B * steps / dt
    sd = lambda m: {k: v.cpu() for k, v in m.state_dict().items()}
    orc = DiscreteSacOracle(sd(pl._actor), sd(pl._critic), sd(pl._critic_target), A)
    idx = torch.randint(0, 1000, (B,))
    eye = torch.eye(A)
    batch = dict(state=torch.randn(B, S), action=eye[idx % A], reward=(idx % 7).float(),
                 terminated=(idx % 50 == 0), next_state=torch.randn(B, S),
                 curr_available_actions=eye.expand(B, A, A), next_available_actions=eye.expand(B, A, A),
                 curr_unavailable_actions_mask=torch.zeros(B, A, dtype=torch.bool),
                 next_unavailable_actions_mask=torch.zeros(B, A, dtype=torch.bool))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < cpu_seconds:
        orc.learn_batch(batch)
        n += 1
    cpu = B * n / (time.perf_counter() - t0)
    return {"config": "cfg2 shapes, discrete SoftActorCritic S=128 A=16 twin-Q [256,256] B=1024",
            "metric": "learner transitions/s through PolicyLearner.learn (sample+preprocess+learn_batch)",
            "value": gpu, "steps": steps, "ms_per_step": 1e3 * dt / steps,
            # actor update (soft_actor_critic.py:153-208): actor forward / dX / dW, both critics on
            # all A actions (no gradient); critic update (:210-287): actor at s', both TARGET critics
            # on all A actions, both online critics at (s, a): forward, dX, dW.  An all-actions pass
            # is counted with the one-hot structure of its first layer, as for DQN:
            # the state part once per row, hidden + output layers once per (row, action).
            "roofline": step_roofline(
                2 * (3 * mlp_macs([S, 256, 256, A]) + 256 * 256 + 256 * A
                     + 4 * (S * 256 + A * (256 * 256 + 256))
                     + 2 * (2 * mlp_macs([S + A, 256, 256, 1]) + 256 * 256 + 256)),
                B * steps, dt, kernel="target_fused_kernel (4 all-actions passes per step) + generic "
                                      "row passes; per-kernel durations: profiles/r02_dsac_kernel_stats.txt"),
            "cpu_baseline": {"value": cpu, "kind": "port", "cores": torch.get_num_threads(),
                             "sample": f"{n} oracle learn_batch calls on one batch (no sampling cost)"}}


def bench_ppo(steps, cpu_seconds):
    """Single process, or — under `torchrun --nproc-per-node N bench_algos.py --only ppo` —
    BASELINE config 4's data-parallel form: every rank owns a rollout shard of 65 536 transitions
    and steps on its own minibatch of 4096; actor + critic gradients travel as ONE RCCL message
    per step (FlatMlp.adam_pair_data_parallel).  value = B * steps * world / max-over-ranks time."""
    import torch.distributed as dist
    from oracle.actor_critic_oracle import PpoOracle
    from pearl_amd import (OneHotActionTensorRepresentationModule, PearlAgent, PPOReplayBuffer,
                           ProximalPolicyOptimization, _comm)
    S, A, B, N = 256, 16, 4096, 65_536
    world = dist.get_world_size() if dist.is_initialized() else 1
    rank = dist.get_rank() if dist.is_initialized() else 0
    torch.manual_seed(0)                 # identical initial parameters on every rank
    random.seed(1000 + rank)             # rank-private minibatch stream
    pl = ProximalPolicyOptimization(action_space=dspace(A), state_dim=S, actor_hidden_dims=[256, 256],
                                    critic_hidden_dims=[256, 256], training_rounds=steps, batch_size=B,
                                    epsilon=0.1,
                                    action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = PPOReplayBuffer(N, sampler="device")
    PearlAgent(pl, replay_buffer=rb, device_id=DEV.index)
    g = torch.Generator(device=DEV).manual_seed(rank)       # rank-private rollout shard
    st = torch.randn(N + 1, S, device=DEV, generator=g)
    ids = torch.arange(N, device=DEV)

    def fill():
        rb.clear()
        rb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 7).float(),
                     terminated=(ids % 500 == 499), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
                     next_state=st[1:], curr_available_actions=dspace(A),
                     next_available_actions=dspace(A), max_number_actions=A)

    fill()
    dt_pre, _ = timed(lambda: pl.preprocess_replay_buffer(rb))
    # one full learn() as warm-up (first-call allocations, scratch growth), then the timed one;
    # both include preprocess_replay_buffer (1.2 ms), as every PPO learn() does
    if world > 1:
        pl.learn(rb)                      # warm-up (incl. the communicator's first collective)
        sync()
        dist.barrier()
        sync()
        gc.collect()
        gc.disable()
        t0 = time.perf_counter()
        pl.learn(rb)
        sync()
        dist.barrier()
        sync()
        dt = time.perf_counter() - t0
        gc.enable()
        t = torch.tensor([dt], device=DEV, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    else:
        dt, _ = timed(lambda: pl.learn(rb), warm=1)
    gpu = B * steps * world / dt
    kernels = []
    if world == 1:
        keep = pl._training_rounds
        pl._training_rounds = 16
        times = engine_kernel_times(lambda: pl.learn(rb))
        from pearl_amd import _native as N_
        kernels = engine_kernels(times, ([S, 256, 256, A], [S, 256, 256, 1]), B,
                                 split_rowstep=bool(N_.lib().pa_rowstep_last_split()),
                                 split_dw=os.environ.get("PEARL_AMD_DW_SPLIT", "1") != "0")
        pl._training_rounds = keep
    comm = _comm.comm_info() if dist.is_initialized() else None
    if comm is not None:
        comm["allreduce_floats_per_step"] = int(sum(m.flat["grad"].numel() for m in pl._flat.values()))
        comm["messages_per_step"] = 1
    if rank != 0:
        return None
    orc = PpoOracle({k: v.cpu() for k, v in pl._actor.state_dict().items()},
                    {k: v.cpu() for k, v in pl._critic.state_dict().items()}, A, epsilon=0.1)
    Ns = 4096      # bounded slice of the rollout for the python GAE loop of the oracle
    scpu = st[:Ns + 1].cpu()
    onehot = torch.eye(A)[(torch.arange(Ns) % A)]
    rew, term = (torch.arange(Ns) % 7).float(), (torch.arange(Ns) % 500 == 499)
    t0 = time.perf_counter()
    gae, lam_ret, p_old = orc.preprocess(scpu[:Ns], onehot, rew, term,
                                         torch.zeros(Ns, dtype=torch.bool), scpu[Ns])
    cpu_pre = Ns / (time.perf_counter() - t0)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < cpu_seconds:
        orc.learn_batch(scpu[:Ns], onehot, p_old, gae, lam_ret)
        n += 1
    cpu = Ns * n / (time.perf_counter() - t0)
    return {"config": "cfg4 PPO+GAE S=256 A=16 [256,256] rollout 65536 minibatch 4096",
            "metric": "learner transitions/s through PolicyLearner.learn (minibatch sample + learn_batch)",
            "value": gpu, "steps": steps, "ms_per_step": 1e3 * dt / steps, "n_gpus": world,
            "scaling": "weak", "comm": comm,
            "roofline": step_roofline(
                # actor [S,256,256,A] + critic [S,256,256,1]: forward, dX of layers >= 1, dW of all
                2 * sum(2 * mlp_macs(d) + sum(a * b for a, b in zip(d[1:-1], d[2:]))
                        for d in ([S, 256, 256, A], [S, 256, 256, 1])),
                B * steps, dt,      # per GPU
                kernel="mlp_rowfwd_kernel 43 us + weight_grad_kernel 43 us + mlp_rowbwd_kernel 25 us per "
                       "step (per-kernel durations: profiles/r02_ppo_kernel_stats.txt)"),
            "kernels": kernels,
            "preprocess_replay_buffer": {"transitions_per_s": N / dt_pre, "ms": 1e3 * dt_pre,
                                         "what": "action probs + values of 65536 states, GAE / lambda-return scan"},
            "cpu_baseline": {"value": cpu, "kind": "port", "cores": torch.get_num_threads(),
                             "preprocess_transitions_per_s": cpu_pre,
                             "sample": f"{n} oracle learn_batch calls on a 4096-transition minibatch; "
                                       f"GAE loop timed on {Ns} transitions"}}


def bench_bandit(steps, cpu_seconds):
    from oracle.actor_critic_oracle import NeuralLinearOracle
    from pearl_amd import NeuralLinearBandit, TransitionBatch
    F, B = 512, 4096
    torch.manual_seed(0)
    pl = NeuralLinearBandit(feature_dim=F, hidden_dims=[256, 64], batch_size=B, learning_rate=1e-3)
    sd0 = {k: v.clone() for k, v in pl.model.state_dict().items()}
    pl.to(DEV)
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(B, F, device=DEV, generator=g)
    y = torch.rand(B, device=DEV, generator=g)
    tb = TransitionBatch(state=x, action=torch.zeros(B, 1, device=DEV), reward=y, weight=None)

    def run():
        for _ in range(steps):
            pl.learn_batch(tb)

    dt, _ = timed(run)
    gpu = B * steps / dt
    times = engine_kernel_times(lambda: [pl.learn_batch(tb) for _ in range(16)])
    from pearl_amd import _native as N_
    kernels = engine_kernels(times, ([F, 256, 64, 1],), B,
                             split_rowstep=bool(N_.lib().pa_rowstep_last_split()),
                             split_dw=os.environ.get("PEARL_AMD_DW_SPLIT", "1") != "0")
    orc = NeuralLinearOracle(sd0, lr=1e-3)
    xc, yc = x.cpu(), y.cpu()
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < cpu_seconds:
        orc.learn_batch(xc, yc, None)
        n += 1
    cpu = B * n / (time.perf_counter() - t0)
    return {"config": "cfg5 NeuralLinearBandit 512-dim contexts trunk [256,64] B=4096",
            "metric": "contexts/s through learn_batch (NN step + LinUCB A/b/inv(A)/coefs update)",
            "value": gpu, "steps": steps, "ms_per_step": 1e3 * dt / steps,
            # trunk forward + dW + dX of its second layer, the 65-wide head, A += x x^T, b += x r.
            # The step is pa_bandit_step: three launches on the learner stream (row step incl. the
            # LinUCB operands, weight gradients + AdamW + the moment update in ONE launch, apply); the
            # 65 x 65 fp64 solve (one workgroup, 45 us; 98 us before the in-place kernel) runs on two alternating side streams
            # beside the next steps and no longer bounds the step
            "roofline": dict(step_roofline(
                2 * (2 * mlp_macs([F, 256, 64]) + 256 * 64 + 3 * 65 + 65 * 65), B * steps, dt,
                kernel="mlp_rowstep_kernel + weight_grad_split_kernel (network dW + AdamW + X^T R) "
                       "+ linreg operands / apply; linreg_solve_spd_inplace_kernel off the learner stream"),
                note="bound by the learner stream's three launches (row step 45 us, weight "
                     "gradients + moment update 27 us, apply 4 us)"),
            "kernels": kernels,
            "cpu_baseline": {"value": cpu, "kind": "port", "cores": torch.get_num_threads(),
                             "sample": f"{n} oracle learn_batch calls"}}


def bench_double_dqn(steps, cpu_seconds):
    """DoubleDQN on BASELINE config 2's shapes: the next action comes from the
    ONLINE network, so every round is sequential (all-actions pass on all CUs, then the online chain
    with the value pass beside the row pass's forward half) — the inputs of a window of rounds
    share a gather, the target work cannot be batched."""
    from oracle.pearl_oracle import DqnOracle
    from pearl_amd import (BasicReplayBuffer, DoubleDQN, OneHotActionTensorRepresentationModule,
                           PearlAgent)
    S, A, B, N = 128, 16, 1024, 1_000_000
    torch.manual_seed(0)
    random.seed(0)
    pl = DoubleDQN(state_dim=S, action_space=dspace(A), hidden_dims=[256, 256], training_rounds=steps,
                   batch_size=B, action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = BasicReplayBuffer(N, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
    g = torch.Generator(device=DEV).manual_seed(0)
    st = torch.randn(N + 1, S, device=DEV, generator=g)
    ids = torch.arange(N, device=DEV)
    rb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 7).float(),
                 terminated=(ids % 50 == 0), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
                 next_state=st[1:], curr_available_actions=dspace(A),
                 next_available_actions=dspace(A), max_number_actions=A)
    dt, _ = timed(lambda: agent.learn())
    gpu = B * steps / dt
    orc = DqnOracle({k: v.cpu() for k, v in pl._Q.state_dict().items()},
                    {k: v.cpu() for k, v in pl._Q_target.state_dict().items()}, double_q=True)
    idx = torch.randint(0, 1000, (B,))
    batch = dict(state=torch.randn(B, S), action=torch.eye(A)[idx % A], reward=(idx % 7).float(),
                 terminated=(idx % 50 == 0), next_state=torch.randn(B, S),
                 next_available_actions=torch.eye(A).expand(B, A, A),
                 next_unavailable_actions_mask=torch.zeros(B, A, dtype=torch.bool))
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < cpu_seconds:
        orc.training_steps += 1
        orc.learn_batch(batch)
        n += 1
    cpu = B * n / (time.perf_counter() - t0)
    return {"config": "cfg2 shapes, DoubleDQN S=128 A=16 [256,256] B=1024 replay 1M",
            "metric": "learner transitions/s through PolicyLearner.learn (sample+preprocess+learn_batch)",
            "value": gpu, "steps": steps, "ms_per_step": 1e3 * dt / steps,
            # DQN's count with the all-actions pass on the ONLINE network and a
            # single-action pass on the target network (double_dqn.py:29-57): online forward, dW +
            # dX, online all-actions (argmax), target value of the chosen action
            "roofline": step_roofline(
                2 * (2 * mlp_macs([S + A, 256, 256, 1]) + 256 * 256 + 256
                     + S * 256 + A * (256 * 256 + 256) + S * 256 + 256 * 256 + 256),
                B * steps, dt, kernel="target_fused_kernel (online all-actions pass, then the value "
                                      "pass) + online_rowpass_kernel + weight_grad_kernel, one stream"),
            "cpu_baseline": {"value": cpu, "kind": "port", "cores": torch.get_num_threads(),
                             "sample": f"{n} oracle learn_batch calls on one batch (no sampling cost)"}}


def bench_dqn_generic(steps, cpu_seconds):
    """What the GENERIC TD engine costs beyond the fused shape: DeepQLearning on
    BASELINE config 2's data with Q networks the fused `pa_dqn_*` path does not take — three hidden
    layers, a multi-head network, a dueling network — through `generic_q.py` (pa_mlp forward /
    backward / fused dW + AdamW, head kernels; the per-round Python loop), next to the fused path on
    the same buffer."""
    from pearl_amd import (BasicReplayBuffer, DeepQLearning, OneHotActionTensorRepresentationModule,
                           PearlAgent)
    from pearl_amd.neural_networks.sequential_decision_making import q_value_networks as Q
    S, A, B, N = 128, 16, 1024, 200_000
    rb = BasicReplayBuffer(N, sampler="device")
    rb.device_for_batches = DEV
    g = torch.Generator(device=DEV).manual_seed(0)
    st = torch.randn(N + 1, S, device=DEV, generator=g)
    ids = torch.arange(N, device=DEV)
    rb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 7).float(),
                 terminated=(ids % 50 == 0), truncated=torch.zeros(N, dtype=torch.bool, device=DEV),
                 next_state=st[1:], curr_available_actions=dspace(A),
                 next_available_actions=dspace(A), max_number_actions=A)
    rows = {}
    for name, kw in (("fused [256, 256]", dict(hidden_dims=[256, 256])),
                     ("generic: Vanilla [256, 256, 256]", dict(hidden_dims=[256, 256, 256])),
                     ("generic: multi-head [256, 256]", dict(hidden_dims=[256, 256],
                                                              network_type=Q.VanillaQValueMultiHeadNetwork)),
                     ("generic: dueling [256, 256]", dict(hidden_dims=[256, 256],
                                                           network_type=Q.DuelingQValueNetwork))):
        torch.manual_seed(0)
        random.seed(0)
        n = steps if name.startswith("fused") else max(20, min(steps, 100))
        pl = DeepQLearning(state_dim=S, action_space=dspace(A), training_rounds=n, batch_size=B,
                           action_representation_module=OneHotActionTensorRepresentationModule(A), **kw)
        agent = PearlAgent(pl, replay_buffer=rb, device_id=0)
        dt, _ = timed(lambda: agent.learn())
        rows[name] = {"transitions_per_s": B * n / dt, "us_per_round": 1e6 * dt / n, "rounds": n,
                      "fused": bool(pl._fused)}
    main_key = "generic: Vanilla [256, 256, 256]"
    return {"config": "cfg2 data, DeepQLearning beyond the fused shape (generic TD engine) S=128 A=16 B=1024",
            "metric": "learner transitions/s through PolicyLearner.learn (sample+preprocess+learn_batch)",
            "value": rows[main_key]["transitions_per_s"], "steps": rows[main_key]["rounds"],
            "ms_per_step": rows[main_key]["us_per_round"] * 1e-3, "networks": rows,
            # three hidden layers: online fwd + dW + dX on B rows, target all-actions pass on B A rows
            "roofline": step_roofline(
                2 * (3 * mlp_macs([S + A, 256, 256, 256, 1]) - (S + A) * 256
                     + A * mlp_macs([S + A, 256, 256, 256, 1])),
                B * rows[main_key]["rounds"], rows[main_key]["us_per_round"] * 1e-6 * rows[main_key]["rounds"],
                kernel="pa_mlp forward / backward / weight_grad kernels, per-round Python loop")}


def bench_push(steps, cpu_seconds):
    """Ingest: per-transition push() through the pinned staging ring, and
    push_many() from host tensors (one H2D copy + one scatter kernel) — the PCIe-inclusive side of
    the boundary; the reference's push costs 287 us per transition on the CPU."""
    from oracle.pearl_oracle import ReplayOracle
    from pearl_amd import BasicReplayBuffer
    S, A, N1, N2 = 128, 16, 20_000, 1_000_000
    torch.manual_seed(0)
    states = torch.randn(N1 + 1, S)
    sp = dspace(A)
    rb = BasicReplayBuffer(N2, sampler="device")
    rb.device_for_batches = DEV
    acts = [torch.tensor([i % A]) for i in range(A)]
    t0 = time.perf_counter()
    for i in range(N1):
        rb.push(state=states[i], action=acts[i % A], reward=float(i % 7), terminated=(i % 50 == 0),
                truncated=False, curr_available_actions=sp, next_state=states[i + 1],
                next_available_actions=sp, max_number_actions=A)
    _ = len(rb)
    rb.sample(256)          # forces the flush of the staging ring
    sync()
    push_rate = N1 / (time.perf_counter() - t0)
    big = torch.randn(N2 + 1, S).pin_memory()
    ids = torch.arange(N2)
    args = dict(action=(ids % A).view(-1, 1), reward=(ids % 7).float(), terminated=(ids % 50 == 0),
                truncated=torch.zeros(N2, dtype=torch.bool))
    rb2 = BasicReplayBuffer(N2, sampler="device")
    rb2.device_for_batches = DEV
    sync()
    t0 = time.perf_counter()
    rb2.push_many(state=big[:-1], next_state=big[1:], curr_available_actions=sp,
                  next_available_actions=sp, max_number_actions=A, **args)
    rb2.sample(256)
    sync()
    dt = time.perf_counter() - t0
    row_bytes = 2 * S * 4 + 8 + 4 + 2
    orc = ReplayOracle(N1)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < cpu_seconds and n < N1:
        orc.push(states[n], acts[n % A], float(n % 7), n % 50 == 0, False, A, states[n + 1], A, A)
        n += 1
    cpu = n / (time.perf_counter() - t0)
    return {"config": "ingest: cfg2 transitions (128-dim states, 16 actions) into the HBM arena",
            "metric": "transitions/s through ReplayBuffer.push (host python call per transition)",
            "value": push_rate, "steps": N1, "ms_per_step": 1e3 / push_rate,
            "push_many_from_host": {"transitions_per_s": N2 / dt, "GB_per_s": N2 * row_bytes / dt / 1e9,
                                    "what": "1M transitions from pinned host tensors: H2D copies + scatter kernel, PCIe inclusive"},
            # ingest is a copy: the batched path moves row_bytes per transition over PCIe and writes
            # them once to HBM; its bound is the host link (PCIe gen5 x16, ~64 GB/s), not HBM
            "roofline": {"bound": "hbm", "achieved": N2 * row_bytes / dt / 1e9, "peak": 8000.0,
                         "unit": "GB/s", "frac": N2 * row_bytes / dt / 8e12, "traffic": None,
                         "bytes_per_transition": row_bytes,
                         "scope": "push_many from pinned host tensors, PCIe inclusive (the per-transition "
                                  "push() line above is interpreter-bound: one Python call per transition)",
                         "pcie_frac": N2 * row_bytes / dt / 64e9},
            "cpu_baseline": {"value": cpu, "kind": "port", "cores": 1,
                             "sample": f"{n} oracle (deque of per-transition tensors) pushes"}}


def bench_gather(steps, cpu_seconds):
    """The sample / gather kernel at an HBM-bound size: inside the DQN loop a
    window gather moves 22 MB per launch and is launch-bound; here ONE launch gathers 262 144 random
    rows of a 1 M-row cfg2 arena — every stored column of a transition, as
    `_create_transition_batch` returns them (tensor_based_replay_buffer.py:290-400) — 628 MB of
    algorithmic traffic (each byte read once, written once), and the learn loop's own form
    (next_state + reward + flags + x = state || one-hot(action): SURVEY.md §8d's 2 132 B) at the
    same row count.  HIP events on the launch stream, best and median of `steps` launches."""
    from pearl_amd import BasicReplayBuffer, _native as N
    S, A, NR, ROWS = 128, 16, 1_000_000, 262_144
    sp = dspace(A)
    rb = BasicReplayBuffer(NR, sampler="device")
    rb.device_for_batches = DEV
    g = torch.Generator(device=DEV).manual_seed(0)
    chunk = 250_000
    for c in range(0, NR, chunk):
        st = torch.randn(chunk + 1, S, device=DEV, generator=g)
        ids = torch.arange(c, c + chunk, device=DEV)
        rb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 7).float(),
                     terminated=(ids % 50 == 0), truncated=torch.zeros(chunk, dtype=torch.bool, device=DEV),
                     next_state=st[1:], curr_available_actions=sp, next_available_actions=sp,
                     max_number_actions=A)
    idx = torch.randperm(NR, device=DEV, generator=g)[:ROWS].contiguous()
    z = rb._layout
    row = 4 * z.state_dim * 2 + 8 * z.action_elems + 4 + 2 + 2 * z.max_actions * (4 * z.avail_dim + 1)
    n = max(10, min(int(steps), 50))

    def run(fn):
        fn()
        sync()
        ts = []
        for _ in range(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e-3)
        ts.sort()
        return ts[0], ts[len(ts) // 2]

    best, med = run(lambda: rb._gather_batch(idx))
    # the learn loop's form: pa_arena_gather_device with x = state || one-hot(action) and no tables
    x = torch.empty(ROWS, S + A, device=DEV)
    nxt = torch.empty(ROWS, S, device=DEV)
    rew = torch.empty(ROWS, device=DEV)
    term = torch.empty(ROWS, dtype=torch.uint8, device=DEV)
    out = N.BatchOut()
    out.x, out.next_state, out.reward_f32, out.terminated = x.data_ptr(), nxt.data_ptr(), rew.data_ptr(), term.data_ptr()
    out.rep_dim, out.rep_onehot = A, 1
    loop_row = (4 * S + 8 + 4 * S + 4 + 1) + (4 * (S + A) + 4 * S + 4 + 1)       # read + written = 2 130 B
    best2, med2 = run(lambda: rb.arena.gather_device(idx, out))
    ach = 2 * row * ROWS / med / 1e9
    return {"config": f"gather: {ROWS} random rows of a 1M-row cfg2 arena in ONE launch",
            "metric": "transitions/s through one gather launch (every stored column, tables and masks included)",
            "value": ROWS / med, "steps": n, "ms_per_step": 1e3 * med,
            "roofline": {"bound": "hbm", "kernel": "gather_kernel", "achieved": ach, "peak": 8000.0,
                         "unit": "GB/s", "frac": ach / 8000.0, "frac_of_achievable_6300": ach / 6300.0,
                         "bytes_per_transition": 2 * row, "launch_bytes": 2 * row * ROWS,
                         "best_GBps": 2 * row * ROWS / best / 1e9, "traffic": None,
                         "scope": "median of the timed launches; rows are 512-byte contiguous runs at random "
                                  "offsets of a 1.2 GB arena, written to contiguous outputs"},
            "learn_loop_form": {"what": "next_state + reward + terminated + x = state || one-hot(action) "
                                        "(the window gather of pa_dqn_learn), same rows",
                                "bytes_per_transition": loop_row, "ms": 1e3 * med2,
                                "GBps": loop_row * ROWS / med2 / 1e9, "best_GBps": loop_row * ROWS / best2 / 1e9,
                                "frac": loop_row * ROWS / med2 / 8e12}}


def bench_feeder(steps, cpu_seconds):
    """Batched observe (second half; pearl_agent.py:169-211): E = 4096 cfg2-shaped
    environments that live on the device, one batched epsilon-greedy act + one push_many per vector
    step (pearl_amd/vector_env.py), against E = 1 `agent.act` + `agent.observe` per transition."""
    from pearl_amd import (BasicReplayBuffer, BatchedActionResult, BatchedEnvironment, DeepQLearning,
                           OneHotActionTensorRepresentationModule, PearlAgent, VectorEnvFeeder)
    from pearl_amd.pearl_agent import ActionResult
    S, A, E = 128, 16, 4096
    sp = dspace(A)

    class Sim(BatchedEnvironment):
        def __init__(self, n):
            self.n = n
            self.t = 0

        def reset(self, seed=None):
            torch.manual_seed(0)
            self.x = torch.randn(self.n, S, device=DEV)
            return self.x, sp

        def step(self, actions):
            self.t += 1
            a = actions.reshape(self.n)
            self.x = torch.roll(self.x, 1, dims=1) * 0.99 + torch.nn.functional.one_hot(a, S).float()
            reward = self.x.gather(1, a.reshape(-1, 1)).reshape(-1)
            done = torch.zeros(self.n, dtype=torch.bool, device=DEV)
            return BatchedActionResult(observation=self.x, reward=reward, terminated=done, truncated=done)

    def make(n_replay):
        torch.manual_seed(0)
        pl = DeepQLearning(state_dim=S, action_space=sp, hidden_dims=[256, 256], training_rounds=1,
                           batch_size=1024,
                           action_representation_module=OneHotActionTensorRepresentationModule(A))
        return PearlAgent(pl, replay_buffer=BasicReplayBuffer(n_replay, sampler="device"), device_id=DEV.index)

    out = {}
    for exploit in (False, True):
        agent = make(1_000_000)
        f = VectorEnvFeeder(agent, Sim(E))
        f.reset()
        f.run(3, exploit=exploit)
        sync()
        n = max(20, steps // 4)
        t0 = time.perf_counter()
        f.run(n, exploit=exploit)
        sync()
        out["exploit" if exploit else "explore"] = E * n / (time.perf_counter() - t0)
    # the reference's shape of the same loop: one environment, one act + one observe per transition
    agent = make(100_000)
    sim = Sim(1)
    obs, _ = sim.reset()
    agent.reset(obs[0], sp)
    t0 = time.perf_counter()
    n1 = 0
    while time.perf_counter() - t0 < min(cpu_seconds, 4.0):
        a = agent.act(exploit=False)
        r = sim.step(torch.as_tensor(a).reshape(1))
        agent.observe(ActionResult(observation=r.observation[0], reward=float(r.reward[0]),
                                   terminated=False, truncated=False))
        n1 += 1
    sync()
    single = n1 / (time.perf_counter() - t0)
    row_bytes = 2 * S * 4 + 8 + 4 + 2
    return {"config": f"batched observe: {E} device-resident cfg2 environments, DQN act (eps 0.05) + push_many per vector step",
            "metric": "transitions/s through act_many + env.step + push_many", "value": out["explore"],
            "steps": max(20, steps // 4), "ms_per_step": 1e3 * E / out["explore"],
            "exploit_only": out["exploit"],
            "one_env_act_observe_per_transition": single,
            "roofline": {"bound": "hbm", "achieved": out["explore"] * row_bytes / 1e9, "peak": 8000.0,
                         "unit": "GB/s", "frac": out["explore"] * row_bytes / 8e12, "traffic": None,
                         "bytes_per_transition": row_bytes,
                         "scope": "interpreter-bound: ~a dozen launches and one Python loop over E "
                                  "exploration draws per vector step; the arena write itself is one scatter launch"},
            "cpu_baseline": {"value": single, "kind": "port", "cores": 1,
                             "sample": f"{n1} act + observe calls of one pearl_amd agent (the reference's loop shape)"}}


def reference_baselines(configs, seconds, threads=32):
    """cpu_baseline.kind == "reference" for configs 3 / 4 / 5: the reference's own learners timed by
    oracle/ref_cpu_baseline.py in ONE child process that sees no GPU; {} when the reference is not
    staged (oracle/_ref) or the child fails."""
    import subprocess
    script = os.path.join(REPO, "oracle", "ref_cpu_baseline.py")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, script, "--config", ",".join(configs), "--seconds",
                              str(seconds), "--threads", str(min(threads, os.cpu_count() or 1))],
                             env=env, capture_output=True, text=True, timeout=300)
        rows = [json.loads(ln) for ln in out.stdout.splitlines() if ln.startswith("{")]
        return {r["config"]: r for r in rows if "value" in r}
    except Exception:
        return {}


def _max_rel(got, ref, floor=0.01):
    """bench.py's metric: max |got - ref| / |ref| over the elements with |ref| >= floor * max |ref|,
    and max |got - ref| / max |ref| over all of them."""
    got, ref = got.double().cpu().reshape(-1), ref.double().cpu().reshape(-1)
    d = (got - ref).abs()
    big = ref.abs() >= floor * ref.abs().max()
    return float((d[big] / ref.abs()[big]).max()), float(d.max() / ref.abs().max())


def parity_probe(name):
    """Observed error of the HIP path against the REFERENCE's own outputs on the config's full-size
    fixture (tests/golden/{sac_cfg3,ppo_cfg4,bandit_cfg5}_fullbatch.pt — minted by
    oracle/make_golden_ac.py from the real reference): the one-batch quantities north_star's 1e-5 bar
    is about (Q-values / action probabilities / predictions) and the first step's reported losses.
    VERDICT r4 weak-2: the driver line carried this figure for the DQN only."""
    from pearl_amd import TransitionBatch
    gold = os.path.join(REPO, "tests", "golden")
    fixture = {"sac": "sac_cfg3_fullbatch.pt", "ppo": "ppo_cfg4_fullbatch.pt",
               "bandit": "bandit_cfg5_fullbatch.pt"}[name]
    path = os.path.join(gold, fixture)
    if not os.path.exists(path):
        return None
    fx = torch.load(path, map_location="cpu", weights_only=False)
    cfg = fx["config"]
    out = {"fixture": f"tests/golden/{fixture} (outputs of the reference, B={cfg['B']})",
           "rel_floor": "|ref| >= 0.01 max|ref|"}
    if name == "sac":
        from pearl_amd import BasicReplayBuffer, BoxActionSpace, ContinuousSoftActorCritic, PearlAgent
        pl = ContinuousSoftActorCritic(action_space=BoxActionSpace(fx["low"], fx["high"]),
                                       state_dim=cfg["S"], actor_hidden_dims=cfg["hidden"],
                                       critic_hidden_dims=cfg["hidden"], batch_size=cfg["B"])
        pl._actor.load_state_dict(fx["actor0"])
        pl._critic.load_state_dict(fx["critic0"])
        pl._critic_target.load_state_dict(fx["critic_target0"])
        PearlAgent(pl, replay_buffer=BasicReplayBuffer(10), device_id=DEV.index or 0)
        actor, c1, c2 = pl._nets(cfg["B"])
        b = TransitionBatch(**{k: v.to(DEV) for k, v in fx["batch"].items()})
        S, A = cfg["S"], cfg["A"]
        pl.noise_source = lambda B, A_, dev: fx["probe"]["noise"]
        xa = torch.empty(b.state.shape[0], S + A, device=DEV)
        xa[:, :S].copy_(b.state)
        _, _, logp = pl._sample(actor, b.state.contiguous(), xa, keep=False)
        out["max_rel_action"], out["max_abs_over_scale_action"] = _max_rel(xa[:, S:], fx["probe"]["action"])
        out["max_rel_q1"], out["max_abs_over_scale_q1"] = _max_rel(c1.forward(xa).view(-1), fx["probe"]["q1"])
        out["max_rel_q2"], out["max_abs_over_scale_q2"] = _max_rel(c2.forward(xa).view(-1), fx["probe"]["q2"])
        out["max_rel_q_values"] = max(out["max_rel_q1"], out["max_rel_q2"])
        (na, nc), want = fx["noises"][0], fx["reports"][0]
        seq = iter([na, nc])
        pl.noise_source = lambda B, A_, dev: next(seq)
        got = pl.learn_batch(pl.preprocess_batch(b))
        out["first_step_loss_rel"] = {k: abs(float(got[k]) - want[k]) / max(1.0, abs(want[k])) for k in want}
    elif name == "ppo":
        from pearl_amd import (OneHotActionTensorRepresentationModule, PearlAgent, PPOReplayBuffer,
                               ProximalPolicyOptimization)
        A, N = cfg["A"], cfg["N"]
        pl = ProximalPolicyOptimization(
            action_space=dspace(A), state_dim=cfg["S"], actor_hidden_dims=cfg["hidden"],
            critic_hidden_dims=cfg["hidden"], training_rounds=cfg["rounds"], batch_size=cfg["B"],
            epsilon=cfg["epsilon"], action_representation_module=OneHotActionTensorRepresentationModule(A))
        pl._actor.load_state_dict(fx["actor0"])
        pl._critic.load_state_dict(fx["critic0"])
        rb = PPOReplayBuffer(N + 5, sampler="python")
        PearlAgent(pl, replay_buffer=rb, device_id=DEV.index or 0)
        sp = dspace(A)
        for i in range(N):
            rb.push(state=fx["states"][i], action=torch.tensor([int(fx["actions"][i])]),
                    reward=float(fx["rewards"][i]), terminated=bool(fx["terminated"][i]),
                    truncated=bool(fx["truncated"][i]), curr_available_actions=sp,
                    next_state=fx["states"][i + 1], next_available_actions=sp, max_number_actions=A)
        pl.preprocess_replay_buffer(rb)
        out["max_rel_action_probs"], out["max_abs_over_scale_action_probs"] = _max_rel(
            rb.extra["action_probs"], fx["action_probs"].view(-1))
        out["max_rel_gae"], out["max_abs_over_scale_gae"] = _max_rel(rb.extra["gae"], fx["gae"])
        out["max_rel_lam_return"], out["max_abs_over_scale_lam_return"] = _max_rel(rb.extra["lam_return"],
                                                                                   fx["lam_return"])
        import random as _r
        _r.seed(fx["learn_seed"])
        rep = pl.learn(rb)
        out["first_step_loss_rel"] = {
            "actor_loss": abs(rep["actor_loss"][0] - float(fx["actor_losses"][0])) / max(1.0, abs(float(fx["actor_losses"][0]))),
            "critic_loss": abs(rep["critic_loss"][0] - float(fx["critic_losses"][0])) / max(1.0, abs(float(fx["critic_losses"][0])))}
    else:
        sys.path.insert(0, os.path.join(REPO, "tests"))
        from test_oracle_ac_golden import bandit_batches
        from pearl_amd import NeuralLinearBandit
        pl = NeuralLinearBandit(feature_dim=cfg["F"], hidden_dims=cfg["hidden"], batch_size=cfg["B"],
                                learning_rate=1e-3, loss_type=cfg.get("loss", "mse"),
                                output_activation_name=cfg.get("out", "linear"))
        pl.model.load_state_dict(fx["model0"])
        pl.to(DEV)
        (x, r, w), want = next(iter(zip(bandit_batches(fx), fx["reports"])))
        tb = TransitionBatch(state=x.to(DEV), action=torch.zeros(cfg["B"], 1, device=DEV),
                             reward=r.to(DEV), weight=None if w is None else w.to(DEV))
        rep = pl.learn_batch(tb)
        out["max_rel_prediction"], out["max_abs_over_scale_prediction"] = _max_rel(rep["prediction"], want["prediction"])
        out["first_step_loss_rel"] = {"loss": abs(float(rep["loss"]) - want["loss"]) / max(1.0, abs(want["loss"]))}
    return {k: v for k, v in out.items() if v is not None}


def driver_block(cpu_seconds=4.0):
    """bench.py's `other_configs`: BASELINE.json configs[2..4] (SAC / PPO / bandit on one MI355X)
    measured in the driver's own bench run — value through learn(), whole-step roofline fraction,
    the dominant kernel timed live with HIP events, and the REFERENCE's CPU path on the same host
    (VERDICT r3 N3).  Bounded: a few hundred steps per config, one child process for the three
    reference legs."""
    rows = []
    for name, fn, steps in (("sac", bench_sac, 300), ("ppo", bench_ppo, 100), ("bandit", bench_bandit, 100)):
        try:
            r = fn(steps, 0.0)
        except Exception as e:      # a failure here must not take the headline line with it
            rows.append({"config": name, "error": f"{type(e).__name__}: {e}"[:300]})
            continue
        roof = r["roofline"]
        step = roof.get("step", roof)
        row = {"config": name, "workload": r["config"], "metric": r["metric"], "value": r["value"],
               "unit": "contexts/s" if name == "bandit" else "transitions/s",
               "steps": steps, "ms_per_step": r["ms_per_step"], "step_frac": step["frac"],
               "flop_per_transition": step["flop_per_transition"]}
        ks = r.get("kernels") or []
        if name == "sac" and "avg_launch_us" in roof:
            ks = [{"kernel": roof["kernel"], "avg_launch_us": roof["avg_launch_us"],
                   "launches_timed": roof["launches_timed"], "achieved": roof["achieved"],
                   "unit": "TFLOP/s", "frac": roof["frac"], "frac_pipe": roof["frac"], "pipe": "fp32 MFMA"}]
        if ks:
            dom = max(ks, key=lambda k: k["avg_launch_us"])
            row.update({"kernel": dom["kernel"], "kernel_us": dom["avg_launch_us"],
                        "kernel_frac": dom["frac"], "kernel_frac_pipe": dom["frac_pipe"],
                        "kernel_pipe": dom["pipe"], "kernels": ks})
        if "preprocess_replay_buffer" in r:
            row["preprocess_replay_buffer"] = r["preprocess_replay_buffer"]
        try:
            par = parity_probe(name)
            if par is not None:
                row["parity"] = par
        except Exception as e:
            row["parity"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        rows.append(row)
    ref = reference_baselines([r["config"] for r in rows], cpu_seconds) if cpu_seconds > 0 else {}
    for r in rows:
        base = ref.get(r["config"])
        if base is not None:
            base.pop("config", None)
            r["cpu_baseline"] = base
            if "value" in r:
                r["vs_cpu_reference"] = r["value"] / base["value"]
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--cpu-seconds", type=float, default=6.0)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    global DEV
    # torchrun --nproc-per-node N bench_algos.py --only ppo: one rank per GPU over RCCL (config 4's
    # data-parallel form); PEARL_AMD_FORCE_DP=1 drives the same path through a 1-rank communicator
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 or os.environ.get("PEARL_AMD_FORCE_DP") == "1":
        import torch.distributed as dist
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29534")
        DEV = torch.device("cuda", local_rank)
        torch.cuda.set_device(DEV)
        dist.init_process_group("nccl", rank=int(os.environ.get("RANK", "0")), world_size=world,
                                device_id=DEV)
        if not args.only:
            args.only = "ppo"
        assert args.only == "ppo", "the multi-process form is built for --only ppo"
    torch.cuda.set_device(DEV)
    torch.set_num_threads(min(32, os.cpu_count() or 1))   # the CPU oracle's best pool size (bench.py)
    for name, fn in (("sac", bench_sac), ("td3", bench_td3), ("dsac", bench_dsac), ("ppo", bench_ppo), ("bandit", bench_bandit),
                     ("double_dqn", bench_double_dqn), ("push", bench_push), ("feeder", bench_feeder),
                     ("gather", bench_gather), ("dqn_generic", bench_dqn_generic)):
        if args.only and name not in args.only.split(","):
            continue
        out = fn(args.steps, args.cpu_seconds)
        if out is None:       # ranks > 0 of a multi-process run
            continue
        out.setdefault("n_gpus", 1)
        out.update({"unit": "transitions/s", "dtype": "f32", "data": "synthetic"})
        print(json.dumps(out), flush=True)
    if world > 1 or os.environ.get("PEARL_AMD_FORCE_DP") == "1":
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
