#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "sac or dqn" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout 600 python tools/sweep.py default PEARL_AMD_PRIO_FIRST=1 PEARL_AMD_PRIO_FIRST=1,PEARL_AMD_SPLIT_FIRST=2 > gpurun_out/sweep_g.jsonl 2> gpurun_out/sweep_g.err
echo "sweep rc=$?"; cat gpurun_out/sweep_g.jsonl; tail -3 gpurun_out/sweep_g.err
timeout 300 python tools/host_bound.py sac 300 > gpurun_out/host_bound_sac.txt 2>&1
head -30 gpurun_out/host_bound_sac.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sac
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sac -o t -- python $R/bench_algos.py --steps 200 --only sac > $R/gpurun_out/rocprof_sac.log 2>&1
DB=$(ls $R/gpurun_out/prof_sac/*.db $R/gpurun_out/prof_sac/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/sac_kernel_stats.txt 2>&1
head -40 $R/gpurun_out/sac_kernel_stats.txt
rm -f $DB
