#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
for sf in 3 12 1 2 13; do
  echo "== SPLIT_FIRST=$sf"
  PEARL_AMD_SPLIT_FIRST=$sf timeout 100 python tools/shortcall.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['rounds'], round(d['wall_us'], 1), round(d['transitions_per_s'] / 1e6, 2))"
done
