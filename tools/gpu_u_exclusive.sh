#!/bin/bash
# A/B on one box: the remainder's U launch kept off the CUs of a resident row pass by an LDS request
# (PEARL_AMD_U_EXCLUSIVE=1) against the default; 2000-round and 20-round bench lines, interleaved.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
for rep in 1 2 3; do
for v in 0 1; do
  PEARL_AMD_U_EXCLUSIVE=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('U_EXCLUSIVE=$v rep $rep  20-round', round(d['value']/1e6,2), 'M   2000-round', round(d['steady_state']['value']/1e6,2), 'M  round_us_steady', round(d['roofline']['chain']['round_us_steady'],2))"
done
done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf $R/gpurun_out/prof_ux
  PEARL_AMD_U_EXCLUSIVE=$v timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ux -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs > /dev/null 2>&1
  echo "== kernel stats, U_EXCLUSIVE=$v"; python $R/tools/rocpd_summary.py $R/gpurun_out/prof_ux/dqn_results.db 2>&1 | head -8 | cut -c1-150
  rm -rf $R/gpurun_out/prof_ux
done
