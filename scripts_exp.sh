#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -4
run() { echo "== $*"; env "$@" timeout 300 python bench.py --no-cpu-baseline $EXTRA > gpurun_out/b.log 2>&1; echo "rc=$?"; tail -1 gpurun_out/b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d.get('roofline',{}); print(round(d['value']/1e6,2),'M/s', round(d['ms_per_step']*1e3,2),'us/step  target frac', round(r.get('frac',0),3), 'launch_us', round(r.get('avg_launch_us',0),1), 'iso', r.get('isolated',{}).get('frac'), d.get('stage_us'))" || tail -5 gpurun_out/b.log; }
run A=1
EXTRA="--timing-level 2" run PEARL_AMD_OVERLAP=0
run PEARL_AMD_OVERLAP=0
