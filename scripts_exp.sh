#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --timing-level 0 > $R/gpurun_out/rocprof.log 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db 2>&1 | head -8
python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db target_fused 34
rm -f $R/gpurun_out/prof/*.db
