#!/bin/bash
# GPU round: gpu test-suite, smoke, bench (stage timers), bench (headline + cpu baseline), rocprofv3 stats.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
if [ "$SKIP_TESTS" != "1" ]; then
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
fi
timeout 600 python bench.py --timing-level 2 --no-cpu-baseline > gpurun_out/bench_l2.log 2>&1
echo "bench l2 rc=$?"; tail -1 gpurun_out/bench_l2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('stage_us'))"
timeout 900 python bench.py $BENCH_ARGS > gpurun_out/bench.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/bench.log
if [ "$1" != "noprof" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline --timing-level 0 > $R/gpurun_out/rocprof.log 2>&1
  echo "rocprof rc=$?"
  python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats.txt 2>&1
  cat $R/gpurun_out/kernel_stats.txt
  rm -f $R/gpurun_out/prof/*.db
fi
