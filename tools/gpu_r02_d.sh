#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log
timeout 200 python tools/prof_chain.py > gpurun_out/prof_chain_overlap.txt 2>&1
echo "prof overlap rc=$?"; grep -v "^$" gpurun_out/prof_chain_overlap.txt | tail -26
timeout 600 python tools/sweep.py $SWEEP_SPECS > gpurun_out/sweep_d.jsonl 2> gpurun_out/sweep_d.err
echo "sweep rc=$?"; cat gpurun_out/sweep_d.jsonl; tail -3 gpurun_out/sweep_d.err
timeout 300 python tools/shortcall.py --profile > gpurun_out/host_profile.txt 2>&1
echo "hostprof rc=$?"; head -50 gpurun_out/host_profile.txt
