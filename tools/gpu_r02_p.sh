#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests/test_gpu_actor_critic.py -m gpu -q --tb=short -p no:cacheprovider -x -k "native_learn_loop or td3 or sac" > gpurun_out/pytest_ac.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_ac.log
timeout 300 python bench_algos.py --steps 300 --only td3,sac --cpu-seconds 0.5 2>/dev/null | cut -c1-200
timeout 300 python bench_algos.py --steps 300 --only td3,sac --cpu-seconds 0.5 2>/dev/null | cut -c1-200
