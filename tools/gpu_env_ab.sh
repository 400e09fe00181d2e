#!/bin/bash
# Runtime-environment A/B for the per-call fixed cost and the kernel boundaries (round 6):
# kernel arguments in device memory, active wait in front of the interrupt, no interrupts at all.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
run() {  # <label> <env...>
  local label=$1; shift
  echo "== $label"
  for rep in 1 2; do
    env "$@" timeout 300 python tools/shortcall.py --rounds 1,20,100 --calls 40 2>/dev/null | python -c "
import sys, json
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')]
print('   ' + '  |  '.join('%3d rounds %7.1f us (enqueue %6.1f)' % (r['rounds'], r['wall_us'], r['enqueue_us']) for r in rows))"
  done
}
run "default" X=1
run "HIP_FORCE_DEV_KERNARG=1" HIP_FORCE_DEV_KERNARG=1
run "HIP_FORCE_DEV_KERNARG=0" HIP_FORCE_DEV_KERNARG=0
run "ROC_ACTIVE_WAIT_TIMEOUT=2000" ROC_ACTIVE_WAIT_TIMEOUT=2000
run "HSA_ENABLE_INTERRUPT=0" HSA_ENABLE_INTERRUPT=0
run "KERNARG=1 + ACTIVE_WAIT=2000" HIP_FORCE_DEV_KERNARG=1 ROC_ACTIVE_WAIT_TIMEOUT=2000
run "default (again)" X=1
