#!/bin/bash
# Focused GPU-box calls of round 5 (gpurun -- bash tools/gpu_r5.sh <mode>); output under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
MODE=${1:-pair}
benchline() {   # $1 = log file
  grep '^{' $1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read())
    print('value', round(d['value']/1e6,2), 'M  us/round', round(d['ms_per_step']*1e3,2), ' steady', round(d.get('steady_state',{}).get('value',0)/1e6,2), ' step_frac', round(d.get('roofline',{}).get('step',{}).get('frac',0),3))
    p=d.get('parity') or {}
    print('parity', {k: float('%.3g' % v) for k, v in p.items() if isinstance(v, float) and k.startswith(('max_rel', 'hip', 'reference'))})
except Exception as e:
    print('no json line:', e)
"
}
if [ "$MODE" == "pair" ]; then
  # the paired row pass: its tests, the DQN file, bench lines with and without it, phase stamps
  timeout 900 python -m pytest tests/test_gpu_dqn.py -m gpu -q --tb=short -p no:cacheprovider -s -x \
    -k "paired or fullbatch or overlapped_loop or generic_loop or q_values_and_targets or first_step" > gpurun_out/pytest_pair.log 2>&1
  echo "pytest pair rc=$?"; grep -E "passed|failed|max rel Q|Error|error|assert" gpurun_out/pytest_pair.log | tail -20
  for pr in 1 0; do
    PEARL_AMD_ROWPASS_PAIR=$pr timeout 600 python bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs > gpurun_out/bench_pair$pr.log 2> gpurun_out/bench_pair$pr.err
    echo "bench 2000 pair=$pr rc=$?"; benchline gpurun_out/bench_pair$pr.log
    PEARL_AMD_ROWPASS_PAIR=$pr timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/bench_s20_pair$pr.log 2> gpurun_out/bench_s20_pair$pr.err
    echo "bench s20 pair=$pr rc=$?"; benchline gpurun_out/bench_s20_pair$pr.log
  done
  for pr in 1 0; do
    PEARL_AMD_ROWPASS_PAIR=$pr timeout 300 python tools/prof_chain.py > gpurun_out/prof_chain_pair$pr.txt 2>&1
    echo "prof_chain pair=$pr rc=$?"; grep -v amdgpu.ids gpurun_out/prof_chain_pair$pr.txt | tail -32
  done
fi
if [ "$MODE" == "diag" ]; then
  for rnd in 13 15; do
    PROF_PER_WG=1 PROF_ROUND=$rnd timeout 300 python tools/prof_chain.py > gpurun_out/prof_chain_wg_r$rnd.txt 2>&1
    echo "prof_chain round $rnd rc=$?"
  done
  grep -v amdgpu.ids gpurun_out/prof_chain_wg_r13.txt | grep -E "^wg|phase|peer|y cons|end " | awk '{ if ($1=="wg") { if ($12 > 14.5 || $13 > 14.5) print } else print }' | head -60
  for lds in 0 65536; do
    PEARL_AMD_PAIR_LDS=$lds timeout 600 python bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs > gpurun_out/bench_lds$lds.log 2>&1
    echo "lds=$lds"; benchline gpurun_out/bench_lds$lds.log
  done
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs > $R/gpurun_out/rocprof.log 2>&1
  echo "rocprof rc=$?"
  python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats_pair.txt 2>&1
  python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db rowpass 80 >> $R/gpurun_out/kernel_stats_pair.txt 2>&1
  head -14 $R/gpurun_out/kernel_stats_pair.txt | cut -c1-170
  rm -f $R/gpurun_out/prof/*.db
fi
if [ "$MODE" == "quick" ]; then
  # env "name=value ..." per line in $CFGS_FILE (default tools/r5_cfgs.txt): 2000-round bench of each
  while read -r cfg; do
    [ -z "$cfg" ] && continue
    tag=$(echo "$cfg" | tr -c 'A-Za-z0-9=\n' '_' | sed 's/PEARL_AMD_//g' | cut -c1-100)
    st=2000; wm=200
    case "$cfg" in STEPS=20*) st=20; wm=5; cfg=${cfg#STEPS=20 };; esac
    env $cfg timeout 600 python bench.py --gpus 1 --steps $st --warmup $wm --no-cpu-baseline --no-other-configs > gpurun_out/bench_q_$tag.log 2>&1
    echo "== steps=$st $cfg"; benchline gpurun_out/bench_q_$tag.log; grep "^\[debug\]" gpurun_out/bench_q_$tag.log
  done < ${CFGS_FILE:-tools/r5_cfgs.txt}
fi
if [ "$MODE" == "sweep" ]; then
  # knobs of the paired row pass: LDS footprint (co-residency), reserved CUs
  for lds in 0 83968; do for cus in 112 128 144; do
    PEARL_AMD_PAIR_LDS=$lds PEARL_AMD_RESERVED_CUS=$cus timeout 600 python bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs > gpurun_out/bench_sw_${lds}_$cus.log 2>&1
    echo "lds=$lds reserved=$cus"; benchline gpurun_out/bench_sw_${lds}_$cus.log
  done; done
fi
if [ "$MODE" == "tests" ]; then
  timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${K:+-k "$K"} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
fi
