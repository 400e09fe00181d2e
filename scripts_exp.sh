#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$PWD}; mkdir -p gpurun_out
echo "### classic 10 rounds"; PEARL_AMD_PERSIST=0 timeout 300 python tools/prof_target.py 10 2>&1 | tail -11
echo "### persistent 10 rounds"; PEARL_AMD_PERSIST=1 PEARL_AMD_RESERVED_CUS=0 timeout 300 python tools/prof_target.py 10 2>&1 | tail -11
