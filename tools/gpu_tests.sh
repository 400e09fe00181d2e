#!/bin/bash
# gpu test-suite only (all failures), optional -k filter in $K
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${K:+-k "$K"} > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -60 gpurun_out/pytest_gpu.log
