#!/bin/bash
# GPU round: gpu test-suite, smoke, bench (stage timers, serial loop), bench (headline + cpu baseline),
# rocprofv3 kernel trace of the same command; "pmc" as $1 adds the three counter passes.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
if [ "$SKIP_TESTS" != "1" ]; then
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
fi
PEARL_AMD_OVERLAP=0 timeout 600 python bench.py --timing-level 2 --no-cpu-baseline > gpurun_out/bench_l2.log 2>&1
echo "bench l2 rc=$?"; tail -1 gpurun_out/bench_l2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('stage_us'))"
timeout 900 python bench.py $BENCH_ARGS > gpurun_out/bench.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/bench.log
if [ "$SKIP_ALGOS" != "1" ]; then
timeout 600 python bench_algos.py --steps 300 > gpurun_out/bench_algos.jsonl 2> gpurun_out/bench_algos.err
echo "bench_algos rc=$?"; cut -c1-260 gpurun_out/bench_algos.jsonl
fi
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof rc=$?"; tail -1 $R/gpurun_out/rocprof.log | cut -c1-200
python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats.txt 2>&1
python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db target_fused 40 >> $R/gpurun_out/kernel_stats.txt 2>&1
head -12 $R/gpurun_out/kernel_stats.txt
rm -f $R/gpurun_out/prof/*.db
if [ "$1" == "pmc" ]; then
  export PEARL_AMD_OVERLAP=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf $R/gpurun_out/pmc_$tag
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$tag -o dqn --output-format csv -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --timing-level 0 > $R/gpurun_out/pmc_$tag.log 2>&1
    echo "pmc $tag rc=$?"
  done
fi
