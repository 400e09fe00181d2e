#!/bin/bash
# A/B of library builds on one box: gpurun -- bash tools/gpu_ab.sh <only> <steps> <lib-suffix>...
# ("-" = the default libpearl_amd.so).  One bench_algos line per build, then the row step's phase stamps.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
ONLY=$1; STEPS=$2; shift 2
for rep in 1 2; do
for v in "$@"; do
  lib=$R/pearl_amd/libpearl_amd.so; [ "$v" != "-" ] && lib=$R/pearl_amd/libpearl_amd_$v.so
  PEARL_AMD_LIB=$lib timeout 300 python bench_algos.py --only $ONLY --steps $STEPS --cpu-seconds 0.2 2>/dev/null > gpurun_out/ab_$v.jsonl
  echo "== $v (rep $rep)"; python tools/algo_line.py < gpurun_out/ab_$v.jsonl
done
done
if [ -n "$PROF" ]; then
for v in "$@"; do
  lib=$R/pearl_amd/libpearl_amd.so; [ "$v" != "-" ] && lib=$R/pearl_amd/libpearl_amd_$v.so
  echo "== phases $v"; PEARL_AMD_LIB=$lib timeout 200 python $PROF 2>&1 | grep -v amdgpu.ids | head -24
done
fi
