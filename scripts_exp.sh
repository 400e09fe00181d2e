#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 gpurun_out/pytest_gpu.log
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_dqn.py -m gpu -q -p no:cacheprovider -x -k "full_size or fused_loop" 2>&1 | tail -1; done
PEARL_AMD_PINGPONG=2 timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -2
timeout 120 ./tools/corun_bench > gpurun_out/corun_bench.txt 2>&1; timeout 120 ./tools/mfma_bench > gpurun_out/mfma_bench.txt 2>&1; tail -3 gpurun_out/mfma_bench.txt
