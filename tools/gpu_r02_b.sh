#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 200 python tools/prof_chain.py > gpurun_out/prof_chain_overlap.txt 2>&1
echo "prof overlap rc=$?"; cat gpurun_out/prof_chain_overlap.txt | grep -v "^$" | tail -32
PEARL_AMD_OVERLAP=0 timeout 200 python tools/prof_chain.py > gpurun_out/prof_chain_serial.txt 2>&1
echo "prof serial rc=$?"; cat gpurun_out/prof_chain_serial.txt | grep -v "^$" | tail -32
timeout 600 python tools/sweep.py "PEARL_AMD_SPLIT_FIRST=3" "PEARL_AMD_SPLIT_FIRST=1" "PEARL_AMD_SPLIT_FIRST=12" "PEARL_AMD_SPLIT_FIRST=2" "PEARL_AMD_SPLIT_FIRST=13" \
   "PEARL_AMD_RESERVED_CUS=48" "PEARL_AMD_RESERVED_CUS=80" "PEARL_AMD_RESERVED_CUS=96" "PEARL_AMD_RESERVED_CUS=80,PEARL_AMD_SPLIT_FIRST=1" "PEARL_AMD_PINGPONG=0" "PEARL_AMD_PINGPONG=0,PEARL_AMD_SPLIT_FIRST=1" > gpurun_out/sweep_b.jsonl 2> gpurun_out/sweep_b.err
echo "sweep rc=$?"; cat gpurun_out/sweep_b.jsonl; tail -3 gpurun_out/sweep_b.err
