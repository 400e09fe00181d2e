cd $GRAFT_REPO_ROOT
for cfg in "X=0" "PEARL_AMD_HEAD_EARLY=1" "PEARL_AMD_FIRST_UNSPLIT=1" "PEARL_AMD_HEAD_EARLY=1 PEARL_AMD_FIRST_UNSPLIT=1" "X=0" "PEARL_AMD_HEAD_EARLY=1 PEARL_AMD_FIRST_UNSPLIT=1"; do
echo "== $cfg"
env $cfg timeout 300 python tools/shortcall.py --calls 40 --rounds 1,3,10,20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  rounds %3d wall %7.1f us  host %7.1f enq %6.1f  %.2f M' % (d['rounds'], d['wall_us'], d['host_us'], d['enqueue_us'], d['transitions_per_s']/1e6))"
done
