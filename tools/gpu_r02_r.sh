#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_loop
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_loop -o t -- python $R/tools/debug_loop.py > $R/gpurun_out/rocprof_loop.log 2>&1
tail -2 $R/gpurun_out/rocprof_loop.log
DB=$(ls $R/gpurun_out/prof_loop/*.db $R/gpurun_out/prof_loop/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_timeline.py $DB "5, 0, true" 30 2>&1 | cut -c1-150
rm -f $DB
