#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_loop
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $R/gpurun_out/prof_loop -o t -- python $R/tools/host_bound.py sac > $R/gpurun_out/rocprof_loop.log 2>&1
grep "host enqueue" $R/gpurun_out/rocprof_loop.log
DB=$(ls $R/gpurun_out/prof_loop/*.db $R/gpurun_out/prof_loop/*/*.db 2>/dev/null | head -1)
python - $DB <<'PY'
import sys
sys.path.insert(0, "/root/repo/tools")
from rocpd_timeline import rows_of, show
rows = rows_of(sys.argv[1])
ks = [i for i, r in enumerate(rows) if "sample_indices_kernel" in r[0]]
print("calls at rows", ks, "of", len(rows))
for a, b in zip(ks, ks[1:] + [len(rows)]):
    seg = rows[a:b]
    print(f"call: {len(seg)} ops, span {(seg[-1][2] - seg[0][1]) / 1e6:.2f} ms, busy {sum(r[2] - r[1] for r in seg) / 1e6:.2f} ms")
    # biggest gaps
    gaps = sorted(((seg[i + 1][1] - seg[i][2]) / 1e3, i) for i in range(len(seg) - 1))[-5:]
    print("  largest gaps (us, after op):", [(round(g, 1), i, seg[i][0][:30]) for g, i in gaps])
    durs = sorted(((r[2] - r[1]) / 1e3, i, r[0][:40]) for i, r in enumerate(seg))[-5:]
    print("  longest ops:", [(round(d, 1), i, n) for d, i, n in durs])
if len(ks) > 1:
    show(rows[ks[1]:ks[1] + 30], rows[ks[1]][1])
PY
rm -f $DB
