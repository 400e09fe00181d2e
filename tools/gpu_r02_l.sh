#!/bin/bash
# driver-style short bench: timing level 0 vs 1, alternating
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
for i in 1 2 3; do
  for lvl in 1 0; do
    python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --timing-level $lvl 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('level $lvl', round(d['value']/1e6,2), 'M tr/s', round(d['ms_per_step']*1e3,1), 'us/step')"
  done
done
python tools/shortcall.py 2>/dev/null
