#!/bin/bash
# hand-off knobs: 1- and 20-round calls (3 reps) and 2000-round throughput per setting
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
run() {
  echo "=== $*"
  for i in 1 2 3; do env "$@" timeout 300 python tools/shortcall.py 2>/dev/null | python -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin if l.startswith('{')]
print('  ', ' | '.join('%d rounds %.1f us' % (d['rounds'], d['wall_us']) for d in rows))"; done
  env "$@" timeout 300 python bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  2000 rounds', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), 'us')"
}
for cfg in "$@"; do run $cfg; done
