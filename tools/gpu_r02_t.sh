#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for cfg in "2048 512" "1024 512" "1024 256"; do
  set -- $cfg
  echo "== MINB=$1 SLICE=$2"
  PEARL_AMD_DW_MINB=$1 PEARL_AMD_DW_SLICE=$2 timeout 300 python bench_algos.py --steps 300 --only td3,sac --cpu-seconds 0.3 2>/dev/null | cut -c150-215
  PEARL_AMD_DW_MINB=$1 PEARL_AMD_DW_SLICE=$2 timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | cut -c1-120
done
