#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_actor_critic.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -3
timeout 300 python bench_algos.py --only ppo --steps 200 --cpu-seconds 0.5 2>/dev/null | grep '^{' | cut -c1-330
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_ppo
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ppo -o a -- python $R/bench_algos.py --only ppo --steps 100 --cpu-seconds 0.1 > $R/gpurun_out/rocprof_ppo.log 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/prof_ppo/a_results.db 2>&1 | head -9
rm -f $R/gpurun_out/prof_ppo/*.db
