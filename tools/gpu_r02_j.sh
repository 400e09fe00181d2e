#!/bin/bash
# SAC fused-rows round: tests, phase stamps, bench line
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
K="${K:-sac}" bash tools/gpu_tests.sh | tail -15
timeout 300 python tools/prof_sac.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/prof_sac.txt
timeout 300 python tools/host_bound.py sac 2>&1 | grep "host enqueue"
timeout 600 python bench_algos.py --only ${ONLY:-sac} --cpu-seconds 1 > gpurun_out/bench_algos_j.jsonl 2> gpurun_out/bench_algos_j.err
echo "bench_algos rc=$?"; cut -c1-300 gpurun_out/bench_algos_j.jsonl
