#!/bin/bash
# One GPU-box call of round 3 (gpurun -- bash tools/gpu_call.sh <mode>); output under gpurun_out/.
#   a: split-MFMA prototype + gpu tests + smoke + the driver's bench line + long bench + per-call cost
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
MODE=${1:-a}
if [ "$MODE" == "a" ]; then
  timeout 120 tools/split_mfma_bench > gpurun_out/split_mfma_bench.txt 2>&1; echo "split rc=$?"; cat gpurun_out/split_mfma_bench.txt
  timeout 900 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | cut -c1-3000
  timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err
  echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-1500
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
  echo "shortcall rc=$?"; cat gpurun_out/shortcall.jsonl
fi
