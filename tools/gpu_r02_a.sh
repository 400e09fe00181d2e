#!/bin/bash
# Round-2 GPU call A: gpu tests, the driver's own bench command, the per-call fixed cost of learn()
# and a kernel timeline of one 20-round call.
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_s20.log 2>&1
echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | cut -c1-1500
timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
echo "shortcall rc=$?"; cat gpurun_out/shortcall.jsonl; tail -3 gpurun_out/shortcall.err
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_long.log 2>&1
echo "bench long rc=$?"; tail -1 gpurun_out/bench_long.log | cut -c1-1200
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sc
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
echo "rocprof rc=$?"
DB=$(ls $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/shortcall_timeline.txt 2>&1
head -70 $R/gpurun_out/shortcall_timeline.txt
rm -f $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db
