#!/usr/bin/env python3
"""Headline benchmark: learner transitions/sec through PolicyLearner.learn() for the DQN of
BASELINE.json config 2 (128-dim obs, 16 discrete actions, hidden [256,256], replay 1M, B=1024).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one round of learn(): device-side sample + gather(+one-hot) + learn_batch on one
batch of 1024 transitions that are already resident in HBM.  Rank 0 prints ONE JSON line.
"""
import argparse
import gc
import json
import os
import random
import sys
import time

import torch
import torch.distributed as dist

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

S, A, HIDDEN, N_REPLAY, B = 128, 16, [256, 256], 1_000_000, 1024
# Algorithmic work:
FLOP_PER_TRANSITION_STEP = 2.713e6          # whole learn step, one-hot-structured layer 1
FLOP_TARGET_KERNEL_PER_TRANSITION = (2 * A * 256 * 256 + 2 * A * 256)   # layer 2 + layer 3 of the target net
PEAK_F32_MFMA = 157.3e12                    # MI355X_MICROARCH.md, dense fp32 matrix peak
PEAK_BF16_MFMA = 2500e12                    # MI355X_MICROARCH.md, dense bf16 matrix peak
PEAK_HBM_GBS = 8000.0                       # MI355X_MICROARCH.md, HBM3E spec peak (6.3 TB/s achievable)
# the online chain of one round (SURVEY.md §8d's derivation, 1024 rows, layers 144 -> 256 -> 256 -> 1):
FLOP_ROWPASS_PER_TRANSITION = 2 * (144 * 256 + 256 * 256 + 256) + 2 * (256 * 256 + 256)   # forward + dX of layers 2-3
FLOP_DW_PER_TRANSITION = 2 * (144 * 256 + 256 * 256 + 256)                                   # dW of the three layers


def pmc_chain():
    """Matrix-pipe busy fraction of the chain's kernels from the newest committed PMC pass
    (tools/pmc_summary.py --chain-json; separate rocprofv3 --pmc runs, never inside this run)."""
    for name in ("r06_final_pmc_chain.json", "r06_zzz_pmc_chain.json", "r06_z_pmc_chain.json", "r06_pmc_chain.json",
                 "r05_pmc_chain.json"):
        path = os.path.join(REPO, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            out = {}
            for k, v in d.get("kernels", {}).items():
                key = ("rowpass" if ("online_rowpass_h2_kernel<0>" in k or
                                     ("online_rowpass_kernel" in k and ", 0>" in k)) else
                       "weight_grad" if "weight_grad" in k else None)
                if key:
                    out[key] = v["mfma_busy_frac"]
            return out, f"profiles/{name}"
    return {}, None


def gather_bytes(with_x: bool, tables: bool) -> int:
    """Algorithmic HBM bytes per transition of one window gather of the DQN learn loop (cfg2):
    read  next_state 512 + reward 4 + terminated 1  (+ state 512 + action 8 when the launch also
          writes the chain's x)  (+ the row's padded next-action table 64 + mask 16 when rows do
          not share one table);
    write next_state 512 + reward 4 + terminated 1  (+ x = state || one-hot(action) 576)
          (+ one-hot (A, A) table 1024 + mask 16)."""
    rd = 4 * S + 4 + 1 + ((4 * S + 8) if with_x else 0) + ((4 * A + A) if tables else 0)
    wr = 4 * S + 4 + 1 + (4 * (S + A) if with_x else 0) + ((4 * A * A + A) if tables else 0)
    return rd + wr


def space(n):
    from pearl_amd import DiscreteActionSpace
    return DiscreteActionSpace([torch.tensor([k]) for k in range(n)])


def fill_arena(rb, dev, seed):
    """SURVEY.md §8(d) cfg2 inputs: transition i = (S[i], i % 16, float(i % 7), i % 50 == 0,
    False, S[i+1]), generated on the device and ingested with push_many."""
    g = torch.Generator(device=dev).manual_seed(seed)
    chunk = 250_000
    prev_last = torch.randn(1, S, device=dev, generator=g)
    for c in range(0, N_REPLAY, chunk):
        n = min(chunk, N_REPLAY - c)
        st = torch.cat([prev_last, torch.randn(n, S, device=dev, generator=g)])
        ids = torch.arange(c, c + n, device=dev)
        rb.push_many(state=st[:-1], action=(ids % A).view(-1, 1), reward=(ids % 7).float(),
                     terminated=(ids % 50 == 0),
                     truncated=torch.zeros(n, dtype=torch.bool, device=dev), next_state=st[1:],
                     curr_available_actions=space(A), next_available_actions=space(A),
                     max_number_actions=A)
        prev_last = st[-1:].clone()


def cpu_baseline(budget_s: float = 20.0):
    """The oracle (reference-pinned CPU restatement, same deque/cat/MKL cost structure as the
    reference) on the host cores, bounded sample of the same workload."""
    from oracle.pearl_oracle import DqnOracle, ReplayOracle, PARAM_KEYS
    from pearl_amd import DeepQLearning, OneHotActionTensorRepresentationModule
    # The baseline deserves its best configuration: on the 256-thread hosts of the GPU boxes the
    # default intra-op pool (128 threads) is 4.7x SLOWER than 32 threads for these small ops
    # (measured: 8 -> 25.2k, 16 -> 26.9k, 32 -> 29.3k, 64 -> 14.8k, 128 -> 6.2k transitions/s).
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    torch.manual_seed(0)
    random.seed(0)
    n = 50_000
    states = torch.randn(n + 1, S)
    rb = ReplayOracle(n)
    t0 = time.perf_counter()
    for i in range(n):
        rb.push(states[i], torch.tensor([i % A]), float(i % 7), i % 50 == 0, False, A, states[i + 1],
                A, A)
    fill_s = time.perf_counter() - t0
    pl = DeepQLearning(state_dim=S, action_space=space(A), hidden_dims=HIDDEN, batch_size=B,
                       action_representation_module=OneHotActionTensorRepresentationModule(A))
    orc = DqnOracle({k: v for k, v in pl._Q.state_dict().items()},
                    {k: v for k, v in pl._Q_target.state_dict().items()})
    orc.learn(rb, 3, B, A)  # warm-up
    steps = 0
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < budget_s:
        orc.learn(rb, 10, B, A)
        steps += 10
    dt = time.perf_counter() - t0
    return {"value": B * steps / dt, "unit": "transitions/s", "cores": torch.get_num_threads(),
            "kind": "port",
            "sample": f"{steps} learn steps (sample+preprocess+learn_batch, B={B}) on a {n}-entry "
                      f"deque replay, {dt:.1f}s; fill {n / fill_s:.0f} push/s; "
                      f"os.cpu_count()={os.cpu_count()}"}


def reference_cpu_baseline(budget_s: float = 15.0):
    """cpu_baseline.kind == "reference": the reference's own PearlAgent.learn() in a child process
    that sees no GPU (pearl/utils/device.py:48-59 would otherwise put it on cuda:0), from
    oracle/_ref (staged by oracle/stage_ref.sh) — oracle/ref_cpu_baseline.py.  None when the
    reference is not staged or the child fails; the caller then times the port."""
    import subprocess
    script = os.path.join(REPO, "oracle", "ref_cpu_baseline.py")
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    try:
        out = subprocess.run([sys.executable, script, "--seconds", str(budget_s), "--threads",
                              str(min(32, os.cpu_count() or 1))], env=env, capture_output=True,
                             text=True, timeout=240)
        last = [ln for ln in out.stdout.strip().splitlines() if ln.startswith("{")]
        d = json.loads(last[-1]) if last else {}
        return d if "value" in d else None
    except Exception:
        return None


def target_dtype():
    """The arithmetic the path computes in: fp32 results everywhere; the two GEMM-heavy kernels run
    their products on the 16-bit matrix pipe on exactly split operands."""
    if os.environ.get("PEARL_AMD_TARGET_SPLIT", "1") == "0":
        return "f32"
    h2t = os.environ.get("PEARL_AMD_TARGET_H2", "1") != "0" and os.environ.get("PEARL_AMD_FUSE_U", "0") == "0"
    h2r = os.environ.get("PEARL_AMD_ROWPASS_H2", "1") != "0"
    return ("f32 (target layer 2: " + ("fp16x2" if h2t else "bf16x3") + "-split MFMA; online row pass: " +
            ("fp16x2-split MFMA" if h2r else "fp32 MFMA") + "; fp32 accumulate)")


def pmc_traffic(transitions_per_launch, split_on):
    """HBM bytes of one target-kernel launch from committed rocprofv3 PMC passes (separate --pmc runs
    of the single-stream loop; tools/pmc_traffic.py: FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950, + WRITE_SIZE, per transition).  Counters cannot be collected inside this
    run, so the line names the file the figure comes from and the kernel that pass measured; None
    when no pass exists for the kernel that ran."""
    names = (["r06_final_pmc_target.json", "r06_zzz_pmc_target.json", "r06_z_pmc_target.json", "r06_pmc_target.json", "r05_pmc_target.json", "r04_pmc_target.json",
              "r03_pmc_target.json"] if split_on else ["r02_pmc_target.json"])
    for name in names:
        path = os.path.join(REPO, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                d = json.load(f)
            kernel = d.get("kernel", "target_split_kernel" if split_on else "target_fused_kernel<32>")
            return (d["hbm_bytes_per_transition"] * transitions_per_launch, f"profiles/{name}",
                    kernel + " (single-stream loop, separate rocprofv3 --pmc passes)")
    return None, None, None


def parity_probe(dev):
    """Observed error of the HIP path against the reference ITSELF on BASELINE config 2's own batch
    (tests/golden/dqn_cfg2_fullbatch.pt: outputs of the real reference, minted by
    oracle/make_golden.py): Q(s, a), max_a' Q_target(s', a') and the Bellman target of 1024
    transitions.  north_star's bar is 1e-5 relative; fresh networks have |Q| ~ 0.1 with values
    crossing zero, where an elementwise relative error has no meaning (a dot product's rounding is
    relative to sum |terms|), so `max_rel` is taken over the elements with |ref| >= 1 % of max |ref|
    and `max_abs_over_scale` = max |diff| / max |ref| over all of them."""
    from pearl_amd import (DeepQLearning, OneHotActionTensorRepresentationModule, TransitionBatch)
    path = os.path.join(REPO, "tests", "golden", "dqn_cfg2_fullbatch.pt")
    if not os.path.exists(path):
        return None
    fx = torch.load(path, map_location="cpu", weights_only=False)
    cfg, want = fx["config"], fx["learners"]["dqn"]
    pl = DeepQLearning(state_dim=cfg["S"], action_space=space(cfg["A"]), hidden_dims=cfg["hidden"],
                       training_rounds=1, batch_size=cfg["B"],
                       action_representation_module=OneHotActionTensorRepresentationModule(cfg["A"]))
    pl._Q.load_state_dict(fx["params0"])
    pl._Q_target.load_state_dict(fx["target0"])
    pl = pl.to(dev)
    batch = pl.preprocess_batch(TransitionBatch(
        **{k: (None if v is None else v.to(dev)) for k, v in fx["batch_raw"].items()}))
    out = pl.q_values_and_targets(batch)
    res = {"fixture": "tests/golden/dqn_cfg2_fullbatch.pt (outputs of the reference, B=1024)",
           "rel_floor": "|ref| >= 0.01 max|ref|"}
    for k, name in (("q", "q"), ("next_v", "next_v"), ("target", "target")):
        got, ref = out[k].double().cpu(), want[k].double()
        d = (got - ref).abs()
        big = ref.abs() >= 0.01 * ref.abs().max()
        res[f"max_rel_{name}"] = float((d[big] / ref.abs()[big]).max())
        res[f"max_abs_over_scale_{name}"] = float(d.max() / ref.abs().max())
        res["allclose_pass"] = bool(res.get("allclose_pass", True) and
                                    bool((d <= 1e-6 + 1e-5 * ref.abs()).all()))
    res["max_rel_q_values"] = max(res["max_rel_q"], res["max_rel_next_v"])
    # The yardstick for those figures: the same quantities in float64 on the host (exact to 1e-16)
    # against (a) the reference's own fp32 outputs and (b) the HIP path's — how far fp32 arithmetic
    # itself (MKL's blocked sums there, MFMA k-ordered chains / the bf16x3 split here) is from the
    # exact value, in the same max-relative metric.
    p64 = {k: v.double() for k, v in fx["params0"].items()}
    t64 = {k: v.double() for k, v in fx["target0"].items()}

    def mlp64(w, x):
        h = torch.relu(x @ w["_model.0.0.weight"].t() + w["_model.0.0.bias"])
        h = torch.relu(h @ w["_model.1.0.weight"].t() + w["_model.1.0.bias"])
        return (h @ w["_model.2.0.weight"].t() + w["_model.2.0.bias"]).squeeze(-1)

    bc = {k: (None if v is None else v.cpu()) for k, v in
          (("state", batch.state), ("action", batch.action), ("next_state", batch.next_state),
           ("next_available_actions", batch.next_available_actions),
           ("next_unavailable_actions_mask", batch.next_unavailable_actions_mask))}
    Bn, An = bc["next_available_actions"].shape[:2]
    q64 = mlp64(p64, torch.cat([bc["state"].double(), bc["action"].double()], dim=-1))
    xs = torch.cat([bc["next_state"].double().unsqueeze(1).expand(Bn, An, -1),
                    bc["next_available_actions"].double()], dim=-1)
    nq = mlp64(t64, xs)
    nq[bc["next_unavailable_actions_mask"]] = -float("inf")
    nv64 = nq.max(1)[0]
    for name, exact, k in (("q", q64, "q"), ("next_v", nv64, "next_v")):
        big = exact.abs() >= 0.01 * exact.abs().max()
        for who, val in (("reference", want[k].double()), ("hip", out[k].double().cpu())):
            res[f"{who}_vs_float64_max_rel_{name}"] = float(((val - exact).abs()[big] / exact.abs()[big]).max())
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the SAC / PPO / bandit block (BASELINE configs[2..4]) after the timed region")
    ap.add_argument("--timing-level", type=int, default=1)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("PEARL_AMD_FORCE_DP") == "1":
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    from pearl_amd import (BasicReplayBuffer, DeepQLearning, OneHotActionTensorRepresentationModule,
                           PearlAgent, _native as N)
    import ctypes as C

    torch.manual_seed(0)          # identical initial parameters on every rank
    random.seed(1000 + rank)      # rank-private sampling stream
    pl = DeepQLearning(state_dim=S, action_space=space(A), hidden_dims=HIDDEN,
                       training_rounds=args.warmup, batch_size=B,
                       action_representation_module=OneHotActionTensorRepresentationModule(A))
    rb = BasicReplayBuffer(N_REPLAY, sampler="device")
    agent = PearlAgent(pl, replay_buffer=rb, device_id=local_rank)
    fill_arena(rb, dev, seed=rank)          # rank-private shard, resident in HBM
    assert len(rb) == N_REPLAY

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    nat = pl._ensure_bound(B, A)

    def read_timers():
        out = {}
        for name in ("allreduce", "target", "target_l1", "l1_dual", "online_l1", "gather", "gather_nox", "gather_x", "sample",
                     "online_l2", "head", "bwd_dx", "rowpass", "bwd_dw", "adamw", "soft_update"):
            ms, cnt, units = C.c_double(), C.c_int64(), C.c_int64()
            N.check(N.lib().pa_dqn_get_timing(nat.handle, name.encode(), C.byref(ms), C.byref(cnt)))
            N.check(N.lib().pa_dqn_get_timing_units(nat.handle, name.encode(), C.byref(units)))
            if cnt.value:
                out[name] = {"avg_us": ms.value * 1e3, "n": cnt.value, "units": units.value}
        return out

    # Calibration pass (outside the timed region, BEFORE it and before the warm-up rounds): the same
    # target kernel with the chip to itself, i.e. the single-stream loop, so that the kernel's own
    # efficiency can be told apart from the CU sharing of the overlapped loop.  N = 1 only: multi-GPU runs stay short.  It runs
    # first because a GPU that has been idle takes its first millisecond of work at reduced clocks
    # (tools/firstcall.py: the same 20-round call costs 1.18 ms cold, 1.07 ms in steady state,
    # 1.43 ms after half a second of idling) — the line says so in `untimed_rounds_before`.
    # Like `timeit`: no cyclic-GC pass inside the timed region (a generation-2 pass over a torch
    # process's heap is ~80 ms, eighty times the 20-round region the driver times).  Collected
    # HERE, before the untimed rounds: 80 ms of host work right before the timed region would let
    # the GPU fall idle again (measured: 15.8 M transitions/s instead of 20 M for the 20-round call).
    gc.collect()
    gc.disable()
    isolated = None
    overlapped = os.environ.get("PEARL_AMD_OVERLAP", "1") != "0" and args.timing_level < 2
    calib_rounds = 0
    if world == 1 and args.timing_level == 1 and overlapped:
        calib_rounds = 300
        N.check(N.lib().pa_dqn_set_overlap(nat.handle, 0))
        N.check(N.lib().pa_dqn_enable_timing(nat.handle, 0))
        pl._training_rounds = 100    # (the GPU comes out of idle: these launches are not timed)
        agent.learn()
        N.check(N.lib().pa_dqn_enable_timing(nat.handle, 1))
        pl._training_rounds = 200
        agent.learn()
        torch.cuda.synchronize(dev)
        isolated = read_timers().get("target")
        N.check(N.lib().pa_dqn_set_overlap(nat.handle, 1))
        N.check(N.lib().pa_dqn_enable_timing(nat.handle, 0))
    # the W warm-up rounds, through the same loop the timed region uses, right before it
    if world > 1:
        barrier()      # ranks fill their arenas at different speeds: start the rounds together
        # the same number of untimed rounds as the single-GPU run's calibration pass, so that every
        # N starts its timed region from the same GPU state (clocks, RCCL channels)
        calib_rounds = 300
        pl._training_rounds = calib_rounds
        agent.learn()
    if args.warmup > 0:
        pl._training_rounds = args.warmup
        agent.learn()
    pl._training_rounds = args.steps
    N.check(N.lib().pa_dqn_enable_timing(nat.handle, args.timing_level))
    barrier()
    t0 = time.perf_counter()
    report = agent.learn()          # exactly `steps` rounds; returns after its single host sync
    barrier()
    dt = time.perf_counter() - t0
    gc.enable()
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert len(report["loss"]) == args.steps and all(x == x for x in report["loss"])

    timers = read_timers()
    N.check(N.lib().pa_dqn_enable_timing(nat.handle, 0))
    # Disclosed second figure, AFTER the timed region: the same loop as one long call (2000 rounds),
    # i.e. without the per-call fixed cost and the first-window bubble a 20-round call carries.  Not
    # `value` — the driver's line stays the K rounds it asked for.
    steady = None
    chain_timers = {}
    if world == 1 and args.steps < 1000 and args.timing_level <= 1:
        gc.disable()
        pl._training_rounds = 2000
        # level 1 | 4: one mid-window round in forty also brackets its two chain launches
        # (row pass, weight gradients + AdamW) with HIP events — roofline.chain below
        N.check(N.lib().pa_dqn_enable_timing(nat.handle, 5))
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        rep2 = agent.learn()
        torch.cuda.synchronize(dev)
        dt2 = time.perf_counter() - t1
        gc.enable()
        chain_timers = read_timers()
        N.check(N.lib().pa_dqn_enable_timing(nat.handle, 0))
        assert len(rep2["loss"]) == 2000 and all(x == x for x in rep2["loss"])
        steady = {"rounds": 2000, "value": B * 2000 / dt2, "unit": "transitions/s",
                  "ms_per_step": 1e3 * dt2 / 2000,
                  "step_frac": FLOP_PER_TRANSITION_STEP * B * 2000 / dt2 / PEAK_F32_MFMA,
                  "note": "one 2000-round learn() call run after the timed region (not `value`)"}

    if rank == 0:
        value = B * args.steps * world / dt
        line = {
            "metric": "learner transitions/sec (DQN batch=1024, [256,256] MLP)",
            "value": value, "unit": "transitions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "untimed_rounds_before": args.warmup + calib_rounds,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": target_dtype(),
            "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1]: DeepQLearning synthetic 128-dim obs / "
                                   "16 discrete actions, hidden=[256,256], replay 1M, batch=1024",
                       "global_batch": B * world, "per_gpu_batch": B, "replay_per_gpu": N_REPLAY,
                       "sampler": "device (Philox, without replacement)",
                       "parallelism": f"dp{world}" if world > 1 else "single",
                       "final_loss": report["loss"][-1]},
        }
        if "target" in timers or isolated:
            # A launch covers several rounds of a target-update window: algorithmic flops of the
            # timed launches / their summed duration.  In the overlapped loop these launches share
            # the chip with the online chain (the persistent ones keep off the chain's 64 CUs), so
            # the live figure is a fraction of the WHOLE chip's peak obtained on part of it; the
            # calibration pass gives the same kernel with the chip to itself.
            def kernel_rate(tt):
                per_launch = tt["units"] / tt["n"]
                return (FLOP_TARGET_KERNEL_PER_TRANSITION * per_launch / (tt["avg_us"] * 1e-6),
                        per_launch)

            # live = the launches sampled inside the timed region (the last, largest piece of every
            # 4th window of the call, first window included); without a live sample (single-stream
            # loop at timing level >= 2, multi-GPU runs) the calibration pass stands in and says so
            live = "target" in timers
            split_on = os.environ.get("PEARL_AMD_TARGET_SPLIT", "1") != "0"
            h2_on = split_on and os.environ.get("PEARL_AMD_TARGET_H2", "1") != "0" and os.environ.get(
                "PEARL_AMD_FUSE_U", "0") == "0"
            # products per fp32 product on the 16-bit matrix pipe: 3 (fp16x2 split) or 6 (bf16x3 split)
            nprod = 3.0 if h2_on else 6.0
            tile_name = ("target_h2_kernel (fp16x2 split MFMA, fp32 accuracy)" if h2_on
                         else "target_split_kernel (bf16x3 split MFMA, fp32 accuracy)")
            tt = timers["target"] if live else isolated
            ach, per_launch = kernel_rate(tt)
            traffic, traffic_src, traffic_kernel = pmc_traffic(per_launch, split_on)
            step_rate = FLOP_PER_TRANSITION_STEP * B * args.steps / dt     # per GPU
            line["roofline"] = {"bound": "mfma",
                                "kernel": ((tile_name if split_on else "target_pp_kernel<32>") +
                                           " (persistent launch of a target-update window, "
                                           "overlapped loop)" if live and overlapped
                                           else (tile_name if split_on else
                                                 "target_fused_kernel<32>") + " (classic grid)"),
                                "achieved": ach / 1e12, "peak": PEAK_F32_MFMA / 1e12,
                                "unit": "TFLOP/s", "frac": ach / PEAK_F32_MFMA,
                                "traffic": traffic, "traffic_source": traffic_src,
                                "traffic_kernel": traffic_kernel,
                                "avg_launch_us": tt["avg_us"],
                                "transitions_per_launch": per_launch,
                                "launches_timed": tt["n"],
                                "source": "timed region" if live else "calibration pass"}
            if split_on:
                # `achieved` / `frac` count ALGORITHMIC fp32 FLOPs against the fp32-MFMA peak — the
                # contract figure of SURVEY.md §8(d), comparable across rounds.  The kernel executes
                # its layer-2 product as `nprod` 16-bit MFMA products per fp32 product — round 6: three
                # (operands scaled into fp16's range and split two ways, target_h2_kernel); rounds 3-5: six
                # (three-way exact bf16 split) — fp32 accumulate, results at fp32 accuracy, 1e-5 Q-value
                # parity in tests/: against the dense 16-bit peak the same launches read as below.
                exe = nprod * 2 * A * 256 * 256 + 2 * A * 256 * A + 2 * A * 256
                line["roofline"]["precision"] = (
                    "fp32 results; layer 2 on v_mfma_f32_32x32x16_f16 with fp16x2 operand splits (3 products)"
                    if h2_on else
                    "fp32 results; layer 2 on v_mfma_f32_32x32x16_bf16 with bf16x3 operand splits (6 products)")
                # the figure against the pipe the kernel runs on: an fp32-accurate product costs `nprod`
                # 16-bit MFMA products, so the ceiling for ALGORITHMIC fp32 FLOPs on that pipe is
                # 2.5 PF / nprod (`frac` above can exceed 1 against the fp32 peak).  NOTE: halving the
                # executed products (round 6) doubles this ceiling — frac_pipe of round 6 is not
                # comparable with earlier rounds' (416.7 TF); `achieved` and `frac` are
                line["roofline"]["products_per_fp32_product"] = nprod
                line["roofline"]["peak_pipe"] = PEAK_BF16_MFMA / nprod / 1e12
                line["roofline"]["frac_pipe"] = ach / (PEAK_BF16_MFMA / nprod)
                line["roofline"]["executed"] = {
                    "flop_per_transition": exe, "achieved": ach / 1e12 * exe / FLOP_TARGET_KERNEL_PER_TRANSITION,
                    "peak": PEAK_BF16_MFMA / 1e12, "unit": "TFLOP/s",
                    "frac": ach * exe / FLOP_TARGET_KERNEL_PER_TRANSITION / PEAK_BF16_MFMA,
                    "note": "executed 16-bit MFMA FLOPs against the dense bf16 / fp16 peak (2.5 PFLOP/s)"}
            if isolated and live:
                iach, iper = kernel_rate(isolated)
                line["roofline"]["concurrent_with"] = (
                    "online chain kernels on the CUs the persistent launches keep off (overlapped loop; "
                    + os.environ.get("PEARL_AMD_RESERVED_CUS", "128" if split_on else "64") + " of 256)")
                line["roofline"]["isolated"] = {
                    "achieved": iach / 1e12, "frac": iach / PEAK_F32_MFMA,
                    "frac_pipe": iach / (PEAK_BF16_MFMA / nprod) if split_on else iach / PEAK_F32_MFMA,
                    "avg_launch_us": isolated["avg_us"], "transitions_per_launch": iper,
                    "launches_timed": isolated["n"],
                    "note": "same kernel, single-stream loop, chip to itself (calibration pass before the timed region)"}
            # the whole learner step against the same peak: 2.713 MFLOP per transition / wall time
            line["roofline"]["step"] = {"achieved": step_rate / 1e12,
                                        "frac": step_rate / PEAK_F32_MFMA,
                                        "flop_per_transition": FLOP_PER_TRANSITION_STEP}
        if "roofline" in line and "rowpass" in chain_timers and "bwd_dw" in chain_timers:
            # The CRITICAL PATH of a round: the online chain — row pass (forward + loss + dZ2 / dZ1),
            # then weight gradients + AdamW — runs round after round on the caller's stream while the
            # target pass above has slack on the side stream.  Event-timed live (one mid-window
            # round in forty of the 2000-round call after the timed region), algorithmic FLOPs of
            # SURVEY.md §8(d) against the fp32-MFMA peak of the WHOLE chip; the matrix-pipe busy
            # fraction comes from a committed PMC pass (counters cannot be read inside this run).
            busy, busy_src = pmc_chain()
            rp, dw = chain_timers["rowpass"], chain_timers["bwd_dw"]
            chain_us = rp["avg_us"] + dw["avg_us"]
            fl_rp, fl_dw = FLOP_ROWPASS_PER_TRANSITION * B, FLOP_DW_PER_TRANSITION * B
            line["roofline"]["chain"] = {
                "bound": "mfma (latency-bound as built: 64 + 113 workgroups on 256 CUs)",
                "rowpass": {"kernel": ("online_rowpass_h2_kernel (fp16x2 split MFMA, fp32 accuracy)"
                                       if os.environ.get("PEARL_AMD_ROWPASS_H2", "1") != "0" else "online_rowpass_kernel"),
                            "avg_launch_us": rp["avg_us"],
                            "launches_timed": rp["n"], "flop": fl_rp,
                            "achieved": fl_rp / (rp["avg_us"] * 1e-6) / 1e12,
                            "frac": fl_rp / (rp["avg_us"] * 1e-6) / PEAK_F32_MFMA,
                            "mfma_busy": busy.get("rowpass")},
                "weight_grad": {"kernel": "weight_grad_split_kernel32 (+ AdamW epilogue)",
                                "avg_launch_us": dw["avg_us"], "launches_timed": dw["n"], "flop": fl_dw,
                                "achieved": fl_dw / (dw["avg_us"] * 1e-6) / 1e12,
                                "frac": fl_dw / (dw["avg_us"] * 1e-6) / PEAK_F32_MFMA,
                                "mfma_busy": busy.get("weight_grad")},
                "us_per_round": chain_us, "flop_per_round": fl_rp + fl_dw,
                "achieved": (fl_rp + fl_dw) / (chain_us * 1e-6) / 1e12, "peak": PEAK_F32_MFMA / 1e12,
                "unit": "TFLOP/s", "frac": (fl_rp + fl_dw) / (chain_us * 1e-6) / PEAK_F32_MFMA,
                "round_us_steady": steady["ms_per_step"] * 1e3 if steady else None,
                "mfma_busy_source": busy_src,
                "source": "2000-round call after the timed region (HIP events on the learner stream)"}
        gt = timers.get("gather") or timers.get("gather_nox")
        if gt and "roofline" in line:
            # The sample + gather (+ one-hot) kernel that replaces tensor_based_replay_buffer.py:253-400:
            # HBM-bound.  Algorithmic bytes per transition of the timed launches (what the launch must
            # move: SURVEY.md §8d's 2 132 B = rows read + learner views written, plus the per-row
            # action table / mask when the arena's rows do not share one): see gather_bytes().
            with_x = "gather" in timers
            per_tr = gather_bytes(with_x, tables=not bool(getattr(rb, "shared_action_table", False)))
            per_launch = gt["units"] / gt["n"]
            gbs = per_tr * per_launch / (gt["avg_us"] * 1e-6) / 1e9
            line["roofline"]["gather"] = {
                "bound": "hbm", "kernel": "gather_kernel (window gather of the learn loop, side stream)",
                "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                "bytes_per_transition": per_tr, "transitions_per_launch": per_launch,
                "avg_launch_us": gt["avg_us"], "launches_timed": gt["n"],
                "writes_x": with_x, "source": "timed region"}
        if len(timers) > 1:
            line["stage_us"] = {k: round(v["avg_us"], 2) for k, v in timers.items()}
            line["stage_units"] = {k: v["units"] / v["n"] for k, v in timers.items() if v["units"]}
        if dist.is_initialized():
            # what the gradient exchange actually ran on: the rank count RCCL itself reports for the
            # communicator the learn loop's hooks used (ncclCommCount), not the one asked for
            comm = getattr(pl._native, "comm", None)
            from pearl_amd import _comm
            info = {"library": (_comm.comm_info().get("library", "rccl (native pa_comm_* hooks)")
                                if comm is not None else "torch.distributed all_reduce hooks"),
                    "ranks_requested": world}
            if comm is not None:
                n_seen, me = C.c_int32(-1), C.c_int32(-1)
                N.check(N.lib().pa_comm_info(comm, C.byref(n_seen), C.byref(me)))
                info["ranks_observed"] = n_seen.value
            info["allreduce_floats_per_round"] = int(pl._native.flat["grad"].numel())
            ar = timers.get("allreduce")
            if ar:
                # per-round exchange time as the learner stream sees it (HIP events around
                # allreduce_start .. allreduce_wait, sampled: one mid-window round in forty), and
                # what it is of a round: the first real N-GPU run yields the overlap figure directly
                info["exchange_us"] = ar["avg_us"]
                info["exchange_rounds_timed"] = ar["n"]
                info["exchange_frac_of_round"] = ar["avg_us"] / (1e6 * dt / args.steps)
                info["exchange_GBps_per_rank"] = 4.0 * info["allreduce_floats_per_round"] / (ar["avg_us"] * 1e-6) / 1e9
            info["exchange_stream"] = "learner stream (between the weight-gradient launch and AdamW; the " \
                                      "target pass of the following rounds runs beside it on the side stream)"
            line["comm"] = info
        if steady is not None:
            line["steady_state"] = steady
        if world == 1 and args.timing_level <= 1:
            try:
                line["parity"] = parity_probe(dev)
                # what "Q-values within 1e-5 rel tol" means here, and whether this run met it: the
                # tests' criterion (torch.allclose semantics: |hip - ref| <= atol + rtol * |ref|)
                par = line["parity"]
                if par and "criterion" not in par:
                    par["criterion"] = {
                        "rule": "|hip - ref| <= 1e-6 + 1e-5 * |ref| elementwise (rtol 1e-5 + atol 1e-6), "
                                "q, next_v and target of the reference's own 1024-row batch",
                        "rtol": 1e-5, "atol": 1e-6, "pass": bool(par.get("allclose_pass")),
                        "note": "the reference's fp32 outputs are themselves "
                                f"{par.get('reference_vs_float64_max_rel_q', float('nan')):.2e} from float64 "
                                f"(HIP: {par.get('hip_vs_float64_max_rel_q', float('nan')):.2e}): a pure "
                                "max-relative 1e-5 between two fp32 implementations is not attainable"}
            except Exception as e:
                line["parity"] = {"error": f"{type(e).__name__}: {e}"[:200]}
        if world == 1 and not args.no_other_configs and args.timing_level <= 1:
            # BASELINE.json configs[2..4] in the same record (not `value`): bench_algos.driver_block
            import bench_algos
            bench_algos.DEV = dev
            line["other_configs"] = bench_algos.driver_block(
                cpu_seconds=0.0 if args.no_cpu_baseline else 4.0)
        if world == 1 and not args.no_cpu_baseline:
            # the real reference when it is staged (oracle/_ref), the reference-pinned port otherwise
            base = reference_cpu_baseline()
            line["cpu_baseline"] = base if base is not None else cpu_baseline()
        print(json.dumps(line), flush=True)
    if os.environ.get("PEARL_AMD_DEBUG_WORKERS") == "1" and rank == 0:
        w, n = C.c_int64(), C.c_int64()
        N.check(N.lib().pa_debug_target_workers(nat.handle, C.byref(w), C.byref(n)))
        print(f"[debug] persistent target launches: {n.value}, workgroups that took tiles: "
              f"{w.value} ({w.value / max(1, n.value):.1f} per launch)", file=sys.stderr, flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
