#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
for lvl in 0 1; do
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --timing-level $lvl > gpurun_out/bench_s20_l$lvl.log 2>&1
  echo "bench s20 level $lvl:"; tail -1 gpurun_out/bench_s20_l$lvl.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
timeout 600 python bench_algos.py --steps 300 > gpurun_out/bench_algos.jsonl 2> gpurun_out/bench_algos.err
echo "bench_algos rc=$?"; python - <<'PY'
import json
for l in open('gpurun_out/bench_algos.jsonl'):
    d=json.loads(l); print(round(d['value']/1e6,3), 'M', d['ms_per_step'], d['config'][:60])
PY
PEARL_AMD_MLP_ROWPASS=0 timeout 600 python bench_algos.py --steps 300 --only sac > gpurun_out/bench_algos_norow.jsonl 2>/dev/null
python -c "
import json
for l in open('gpurun_out/bench_algos_norow.jsonl'):
    d=json.loads(l); print('no rowpass:', round(d['value']/1e6,3), 'M', d['ms_per_step'], d['config'][:60])"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_b20
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_b20 -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/rocprof_b20.log 2>&1
DB=$(ls $R/gpurun_out/prof_b20/*.db $R/gpurun_out/prof_b20/*/*.db 2>/dev/null | head -1)
python - "$DB" <<'PY' > $R/gpurun_out/timeline_b20.txt 2>&1
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, stream_id from kernels order by start").fetchall()
# the timed call = the SECOND sample_indices_kernel launch (warm-up is the first)
idx = [i for i, r in enumerate(rows) if "sample_indices_kernel" in r[0]]
k0 = idx[1] - 3
k1 = idx[2] - 1 if len(idx) > 2 else len(rows)
base = rows[k0][1]
print(f"timed call: {(rows[k1-1][2]-rows[k0][1])/1000:.1f} us")
for r in rows[k0:k1]:
    print(f"{(r[1]-base)/1000:9.2f} {(r[2]-base)/1000:9.2f} dur={(r[2]-r[1])/1000:7.2f} s={r[3]} {r[0][:50]}")
PY
head -80 $R/gpurun_out/timeline_b20.txt
rm -f $DB
