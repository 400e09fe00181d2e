#!/bin/bash
# kernel timeline of the 20-round DQN call only (rocprofv3 kernel trace) -> gpurun_out/shortcall_timeline.txt
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sc
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
DB=$(ls $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/shortcall_timeline.txt 2>&1
rm -f $DB
cd $R && for i in 1 2 3; do timeout 300 python tools/shortcall.py 2>/dev/null | sed -n 3p; done
