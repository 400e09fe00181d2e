#!/bin/bash
# timelines for offline reading: the 20-round DQN call, PPO / SAC steps (rocprofv3 kernel traces)
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_sc
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
DB=$(ls $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/shortcall_timeline.txt 2>&1
rm -f $DB
for w in ppo sac; do
  rm -rf $R/gpurun_out/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o t -- python $R/bench_algos.py --steps 200 --only $w --cpu-seconds 0 > $R/gpurun_out/rocprof_$w.log 2>&1
  DB=$(ls $R/gpurun_out/prof_$w/*.db $R/gpurun_out/prof_$w/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/${w}_kernel_stats.txt 2>&1
  python $R/tools/rocpd_timeline.py $DB rowstep 40 >> $R/gpurun_out/${w}_kernel_stats.txt 2>&1
  rm -f $DB
done
cd $R && timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err; cat gpurun_out/shortcall.jsonl
