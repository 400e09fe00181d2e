#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 200 python bench_algos.py --steps 300 --only dsac,bandit,double_dqn,push --cpu-seconds 3 > gpurun_out/bench_algos_rest.jsonl 2> gpurun_out/bench_algos_rest.err
echo "rc=$?"; cut -c1-200 gpurun_out/bench_algos_rest.jsonl; tail -3 gpurun_out/bench_algos_rest.err
cd /tmp && export TMPDIR=/tmp
for w in dsac; do
  rm -rf $R/gpurun_out/prof_$w
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o t -- python $R/bench_algos.py --steps 200 --only $w --cpu-seconds 0.3 > $R/gpurun_out/rocprof_$w.log 2>&1
  DB=$(ls $R/gpurun_out/prof_$w/*.db $R/gpurun_out/prof_$w/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/${w}_kernel_stats.txt 2>&1
  echo "rocprof $w rc=$?"; head -6 $R/gpurun_out/${w}_kernel_stats.txt | cut -c1-150
  rm -f $DB
done
