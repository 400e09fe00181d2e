#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log
timeout 200 python tools/prof_chain.py > gpurun_out/prof_chain_overlap.txt 2>&1
echo "prof overlap rc=$?"; grep -v "^$" gpurun_out/prof_chain_overlap.txt | tail -26
timeout 600 python tools/sweep.py "" "PEARL_AMD_SPLIT_FIRST=1" "PEARL_AMD_SPLIT_FIRST=2" "PEARL_AMD_RESERVED_CUS=80" > gpurun_out/sweep_c.jsonl 2> gpurun_out/sweep_c.err
echo "sweep rc=$?"; cat gpurun_out/sweep_c.jsonl; tail -3 gpurun_out/sweep_c.err
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/cu_census.hip -o /tmp/cu_census && /tmp/cu_census > gpurun_out/cu_census.txt 2>&1
head -12 gpurun_out/cu_census.txt
cd /tmp && export TMPDIR=/tmp
for tag in chain nochain; do
  rm -rf $R/gpurun_out/prof_$tag
  if [ $tag == nochain ]; then export PEARL_AMD_DEBUG_NO_CHAIN=1; fi
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$tag -o t -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_$tag.log 2>&1
  echo "rocprof $tag rc=$?"
  DB=$(ls $R/gpurun_out/prof_$tag/*.db $R/gpurun_out/prof_$tag/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/timeline_$tag.txt 2>&1
  grep -E "target|last learn" $R/gpurun_out/timeline_$tag.txt | head -12
  rm -f $DB
done
