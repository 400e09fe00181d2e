#!/bin/bash
# quick check of the dp / allreduce artefacts + bench_algos roofline lines
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
PEARL_AMD_FORCE_DP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline \
  > gpurun_out/bench_dp1.log 2> gpurun_out/bench_dp1.err
echo "bench dp1 rc=$?"; tail -1 gpurun_out/bench_dp1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('comm'))"; tail -3 gpurun_out/bench_dp1.err
timeout 300 python tools/allreduce_latency.py > gpurun_out/allreduce_latency.json 2> gpurun_out/allreduce_latency.err
echo "allreduce rc=$?"; cat gpurun_out/allreduce_latency.json; tail -3 gpurun_out/allreduce_latency.err
timeout 600 python bench_algos.py --only sac,ppo --cpu-seconds 1 > gpurun_out/bench_algos_k.jsonl 2> gpurun_out/bench_algos_k.err
echo "bench_algos rc=$?"; cut -c1-900 gpurun_out/bench_algos_k.jsonl; tail -3 gpurun_out/bench_algos_k.err
timeout 300 python tools/host_bound.py sac 2>&1 | grep -v amdgpu.ids | head -30
