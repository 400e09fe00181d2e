cd $GRAFT_REPO_ROOT
bash tools/gpu_r6.sh quick 2>&1 | grep -E "^==|^value"
for cfg in "X=0" "PEARL_AMD_LEAD_ROWS=32" "PEARL_AMD_LEAD_ROWS=1"; do
echo "== shortcall $cfg"
env $cfg timeout 300 python tools/shortcall.py --calls 40 --rounds 1,20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  rounds %3d wall %7.1f us' % (d['rounds'], d['wall_us']))"
done
