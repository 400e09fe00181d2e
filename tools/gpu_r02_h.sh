#!/bin/bash
# actor-critic round: tests, host-bound check, bench_algos lines
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
K="${K:-actor_critic}" bash tools/gpu_tests.sh | tail -25
for w in ${WHICH:-sac ppo}; do
  timeout 300 python tools/host_bound.py $w 2>&1 | grep -v amdgpu.ids | tail -4
done
timeout 600 python bench_algos.py --only ${ONLY:-sac,td3,ppo} > gpurun_out/bench_algos_h.jsonl 2> gpurun_out/bench_algos_h.err
echo "bench_algos rc=$?"; cut -c1-330 gpurun_out/bench_algos_h.jsonl
