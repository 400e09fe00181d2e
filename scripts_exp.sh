#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for m in 1 0; do
echo "== forced DP inline=$m"
PEARL_AMD_COMM_INLINE=$m PEARL_AMD_FORCE_DP=1 timeout 300 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 > gpurun_out/bench_dp1.log 2>&1; echo "rc=$?"; grep '"metric"' gpurun_out/bench_dp1.log | cut -c1-230
done
echo "== torch allreduce hooks"
PEARL_AMD_TORCH_ALLREDUCE=1 PEARL_AMD_FORCE_DP=1 timeout 300 python bench.py --no-cpu-baseline --steps 1000 --warmup 100 > gpurun_out/bench_dp1.log 2>&1; echo "rc=$?"; grep '"metric"' gpurun_out/bench_dp1.log | cut -c1-230
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
PEARL_AMD_FORCE_DP=1 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 300 --warmup 50 --no-cpu-baseline --timing-level 0 > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof rc=$?"
python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db 2>&1 | head -9
python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db rowpass 22
rm -f $R/gpurun_out/prof/*.db
