#!/bin/bash
# First GPU contact: full gpu test-suite, smoke, bench (all stages timed).
mkdir -p gpurun_out
rocminfo 2>/dev/null | grep -E "Marketing Name|gfx" | head -4 > gpurun_out/device.txt
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -60 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -5 gpurun_out/smoke.log
timeout 600 python bench.py --timing-level 2 --no-cpu-baseline > gpurun_out/bench_l2.log 2>&1
echo "bench rc=$?"; tail -3 gpurun_out/bench_l2.log
