#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sc
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
DB=$(ls $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/shortcall_timeline.txt 2>&1
head -75 $R/gpurun_out/shortcall_timeline.txt | cut -c1-140
rm -f $DB
