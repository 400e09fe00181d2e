#!/bin/bash
# tests + sweep ($SWEEP_SPECS) + per-spec timelines of a 20-round call ($TRACE_SPECS)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
timeout 600 python tools/sweep.py $SWEEP_SPECS > gpurun_out/sweep_e.jsonl 2> gpurun_out/sweep_e.err
echo "sweep rc=$?"; cat gpurun_out/sweep_e.jsonl; tail -3 gpurun_out/sweep_e.err
cd /tmp && export TMPDIR=/tmp
i=0
for spec in $TRACE_SPECS; do
  i=$((i+1))
  rm -rf $R/gpurun_out/prof_t$i
  ( if [ "$spec" != "default" ]; then export ${spec//,/ }; fi
    timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_t$i -o t -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_t$i.log 2>&1 )
  DB=$(ls $R/gpurun_out/prof_t$i/*.db $R/gpurun_out/prof_t$i/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/timeline_t$i.txt 2>&1
  echo "== timeline $spec"; head -75 $R/gpurun_out/timeline_t$i.txt
  rm -f $DB
done
