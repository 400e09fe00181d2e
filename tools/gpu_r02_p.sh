#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log
timeout 600 python bench_algos.py --steps 300 2>/dev/null | cut -c1-230
