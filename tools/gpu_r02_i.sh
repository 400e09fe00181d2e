#!/bin/bash
# rocprof kernel summaries of the actor-critic learners (bench_algos lines)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in ${WHICH:-sac ppo}; do
  rm -rf $R/gpurun_out/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o t -- python $R/bench_algos.py --steps 200 --only $w --cpu-seconds 0.5 > $R/gpurun_out/rocprof_$w.log 2>&1
  DB=$(ls $R/gpurun_out/prof_$w/*.db $R/gpurun_out/prof_$w/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/${w}_kernel_stats.txt 2>&1
  python $R/tools/rocpd_timeline.py $DB ${ANCHOR:-weight_grad} 40 >> $R/gpurun_out/${w}_kernel_stats.txt 2>&1
  head -${LINES_OUT:-24} $R/gpurun_out/${w}_kernel_stats.txt
  rm -f $DB
done
