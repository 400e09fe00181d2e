#!/bin/bash
# Focused GPU-box calls of round 6 (gpurun -- bash tools/gpu_r6.sh <mode>); output under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
MODE=${1:-new}
benchline() {   # $1 = log file
  grep '^{' $1 | tail -1 | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read())
    print('value', round(d['value']/1e6,2), 'M  us/round', round(d['ms_per_step']*1e3,2), ' steady', round(d.get('steady_state',{}).get('value',0)/1e6,2), ' step_frac', round(d.get('roofline',{}).get('step',{}).get('frac',0),3))
    c=d.get('roofline',{}).get('chain')
    if c: print('chain us', round(c['us_per_round'],2), 'rowpass', round(c['rowpass']['avg_launch_us'],2), 'dw', round(c['weight_grad']['avg_launch_us'],2), 'frac', round(c['frac'],3), 'n', c['rowpass']['launches_timed'])
    p=d.get('parity') or {}
    print('parity', {k: float('%.3g' % v) for k, v in p.items() if isinstance(v, float) and k.startswith(('max_rel', 'hip', 'reference'))}, (p.get('criterion') or {}).get('pass'))
    for r in d.get('other_configs',[]): print(' ', r.get('config'), round(r.get('value',0)/1e6,2), 'M', 'step_frac', round(r.get('step_frac',0),3))
except Exception as e:
    print('no json line:', e)
"
}
if [ "$MODE" == "new" ]; then
  # round 6's new parity tests first, then the files the python-sampler presample / eps=0 changes touch
  timeout 1500 python -m pytest tests/test_gpu_long_runs.py tests/test_gpu_numerics_range.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/pytest_new.log 2>&1
  echo "pytest new rc=$?"; grep -vE "^\s*$|amdgpu.ids" gpurun_out/pytest_new.log | tail -150
  timeout 1500 python -m pytest tests/test_gpu_actor_critic.py -m gpu -q --tb=short -p no:cacheprovider -x -k "ppo or sac or native or gae" > gpurun_out/pytest_ac.log 2>&1
  echo "pytest ac rc=$?"; tail -15 gpurun_out/pytest_ac.log
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; benchline gpurun_out/bench_s20.log; tail -3 gpurun_out/bench_s20.err
fi
if [ "$MODE" == "tests" ]; then
  timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider ${K:+-k "$K"} > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
fi
if [ "$MODE" == "quick" ]; then
  # env "name=value ..." per line in $CFGS_FILE: 2000-round bench of each ("STEPS=20 ..." = the driver's call)
  while read -r cfg; do
    [ -z "$cfg" ] && continue
    tag=$(echo "$cfg" | tr -c 'A-Za-z0-9=\n' '_' | sed 's/PEARL_AMD_//g' | cut -c1-100)
    st=2000; wm=200
    case "$cfg" in STEPS=20*) st=20; wm=5; cfg=${cfg#STEPS=20 };; esac
    env $cfg timeout 600 python bench.py --gpus 1 --steps $st --warmup $wm --no-cpu-baseline --no-other-configs > gpurun_out/bench_q_$tag.log 2>&1
    echo "== steps=$st $cfg"; benchline gpurun_out/bench_q_$tag.log; grep "^\[debug\]" gpurun_out/bench_q_$tag.log
  done < ${CFGS_FILE:-tools/r6_cfgs.txt}
fi
