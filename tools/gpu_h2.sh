#!/bin/bash
# A/B of the fp16x2 row pass (PEARL_AMD_ROWPASS_H2): DQN parity tests, long bench with parity block,
# in-kernel phase stamps, kernel durations by rocprofv3.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
MODE=${1:-all}
summ() { tail -1 $1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
ch=d['roofline'].get('chain') or {}
print('value', round(d['value']/1e6,2), 'M  us/round', round(d['ms_per_step']*1e3,2), 'steady', round((d.get('steady_state') or {}).get('value',0)/1e6,2))
print('chain', {k: ch.get(k) for k in ('rowpass_us','weight_grad_us','frac')})
print('parity', {k: (float('%.3g' % v) if isinstance(v, float) else v) for k, v in (d.get('parity') or {}).items()})"; }
for H2 in $H2S; do
  export PEARL_AMD_ROWPASS_H2=$H2
  echo "=== PEARL_AMD_ROWPASS_H2=$H2"
  if [ "$MODE" == "all" ] || [ "$MODE" == "tests" ]; then
    timeout 900 python -m pytest tests/test_gpu_dqn.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/h2_${H2}_pytest.log 2>&1
    echo "pytest rc=$?"; tail -15 gpurun_out/h2_${H2}_pytest.log
  fi
  if [ "$MODE" == "all" ] || [ "$MODE" == "bench" ]; then
    timeout 600 python bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs > gpurun_out/h2_${H2}_bench.log 2> gpurun_out/h2_${H2}_bench.err
    echo "bench rc=$?"; summ gpurun_out/h2_${H2}_bench.log
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/h2_${H2}_bench20.log 2> gpurun_out/h2_${H2}_bench20.err
    echo "bench20 rc=$?"; summ gpurun_out/h2_${H2}_bench20.log
    timeout 300 python tools/prof_chain.py > gpurun_out/h2_${H2}_chain_stamps.txt 2>&1
    echo "prof_chain rc=$?"; head -40 gpurun_out/h2_${H2}_chain_stamps.txt
  fi
  if [ "$MODE" == "all" ] || [ "$MODE" == "prof" ]; then
    ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_h2_$H2 &&
      timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_h2_$H2 -o t -- python $R/bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs > $R/gpurun_out/h2_${H2}_rocprof.log 2>&1 )
    db=$(ls $R/gpurun_out/prof_h2_$H2/*.db $R/gpurun_out/prof_h2_$H2/*/*.db 2>/dev/null | head -1)
    python $R/tools/rocpd_summary.py $db > $R/gpurun_out/h2_${H2}_kernel_stats.txt 2>&1
    head -14 $R/gpurun_out/h2_${H2}_kernel_stats.txt | cut -c1-160
    rm -f $db
  fi
done
