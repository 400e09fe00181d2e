#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
PEARL_AMD_SAC_SPLIT=0 timeout 100 python tools/host_bound.py sac 2>&1 | grep "host enqueue"
timeout 100 python tools/host_bound.py sac 2>&1 | grep "host enqueue"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sac
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sac -o t -- python $R/bench_algos.py --steps 200 --only sac --cpu-seconds 0.5 > $R/gpurun_out/rocprof_sac.log 2>&1
DB=$(ls $R/gpurun_out/prof_sac/*.db $R/gpurun_out/prof_sac/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/sac_kernel_stats.txt 2>&1
head -14 $R/gpurun_out/sac_kernel_stats.txt | cut -c1-180
rm -f $DB
