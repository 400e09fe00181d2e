#!/bin/bash
# Focused GPU-box calls of round 4 (gpurun -- bash tools/gpu_r4.sh <mode>); output under gpurun_out/.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
MODE=${1:-new}
if [ "$MODE" == "new" ]; then
  # the tests round 4 added or changed, then the driver's bench line with the new blocks
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -s \
    -k "bandit or rollout64k or fullbatch or 200_round or squarecb or full_size or discrete_sac_learn_batch or iql_learn_batch or ddpg_td3_learn_batch" \
    > gpurun_out/pytest_new.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed|rounds  |^ *[0-9]+-|final Q|Error|error" gpurun_out/pytest_new.log | tail -40
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value', round(d['value']/1e6,2), 'M  steady', round(d.get('steady_state',{}).get('value',0)/1e6,2))
print('parity', d.get('parity'))
for r in d.get('other_configs',[]): print({k:(round(v,4) if isinstance(v,float) else v) for k,v in r.items() if k not in ('kernels','workload','metric','preprocess_replay_buffer')})
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('kind'))
"
  tail -3 gpurun_out/bench_s20.err
fi
if [ "$MODE" == "dw" ]; then
  # bf16x3 weight gradients: kernel tests, the PPO / bandit parity tests, bench lines with and without
  timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_actor_critic.py -m gpu -q --tb=short -p no:cacheprovider -s \
    -k "weight_grad or ppo or bandit or twin or rowstep" > gpurun_out/pytest_dw.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed|dW error|Error" gpurun_out/pytest_dw.log | tail -30
  for sp in 1 0; do
    PEARL_AMD_DW_SPLIT=$sp timeout 600 python bench_algos.py --steps 300 --only ppo,bandit --cpu-seconds 0.5 > gpurun_out/bench_algos_dw$sp.jsonl 2> gpurun_out/bench_algos_dw$sp.err
    echo "bench_algos split=$sp rc=$?"; python - <<PY
import json
for ln in open("gpurun_out/bench_algos_dw$sp.jsonl"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["config"][:24], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us/step", [(k["kernel"][:14], round(k["avg_launch_us"],1)) for k in d.get("kernels",[])])
PY
  done
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_ppo
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ppo -o t -- python $R/bench_algos.py --steps 200 --only ppo --cpu-seconds 0.2 > $R/gpurun_out/rocprof_ppo.log 2>&1
  DB=$(ls $R/gpurun_out/prof_ppo/*.db $R/gpurun_out/prof_ppo/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/ppo_kernel_stats.txt 2>&1
  head -12 $R/gpurun_out/ppo_kernel_stats.txt | cut -c1-160
  rm -f $DB
fi
if [ "$MODE" == "prof" ]; then
  timeout 300 python tools/prof_rowstep.py > gpurun_out/prof_rowstep.txt 2>&1; echo "prof_rowstep rc=$?"; grep -v amdgpu.ids gpurun_out/prof_rowstep.txt | tail -32
  PEARL_AMD_DW_SPLIT=0 timeout 300 python tools/prof_rowstep.py > gpurun_out/prof_rowstep_nosplit.txt 2>&1; grep -A9 "weight_grad_kernel of" gpurun_out/prof_rowstep_nosplit.txt
  # host API + kernel timeline of three 20-round calls
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_sc
  timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace --output-format csv -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
  echo "rocprof sc rc=$?"; ls -la $R/gpurun_out/prof_sc/* | head; 
  cd $R; python tools/host_timeline.py $(ls gpurun_out/prof_sc/*hip_api_trace.csv gpurun_out/prof_sc/*/*hip_api_trace.csv 2>/dev/null | head -1) $(ls gpurun_out/prof_sc/*kernel_trace.csv gpurun_out/prof_sc/*/*kernel_trace.csv 2>/dev/null | head -1) > gpurun_out/shortcall_host_timeline.txt 2>&1; head -90 gpurun_out/shortcall_host_timeline.txt
fi
if [ "$MODE" == "p2p" ]; then
  timeout 120 tools/valu_rate_bench > gpurun_out/valu_rate_bench.txt 2>&1; echo "valu bench rc=$?"; cat gpurun_out/valu_rate_bench.txt
  timeout 900 python -m pytest tests/test_gpu_dp.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_dp.log 2>&1
  echo "pytest dp rc=$?"; tail -15 gpurun_out/pytest_dp.log
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err; echo "shortcall rc=$?"; cat gpurun_out/shortcall.jsonl
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/bench_s20b.log 2>&1; tail -1 gpurun_out/bench_s20b.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('value', d['value'], 'steady', d['steady_state']['value'])"
fi
if [ "$MODE" == "pyprof" ]; then
  timeout 300 python tools/shortcall.py --profile > gpurun_out/shortcall_profile.txt 2>&1; grep -v amdgpu gpurun_out/shortcall_profile.txt | head -60
fi
if [ "$MODE" == "hostcost" ]; then
  timeout 300 python tools/host_cost.py > gpurun_out/host_cost.txt 2>&1; grep -v amdgpu gpurun_out/host_cost.txt
fi
if [ "$MODE" == "rs" ]; then
  timeout 900 python -m pytest tests/test_gpu_actor_critic.py -m gpu -q --tb=short -p no:cacheprovider -x \
    -k "ppo or bandit or twin or rowstep or dsac or discrete" > gpurun_out/pytest_rs.log 2>&1
  echo "pytest rc=$?"; tail -12 gpurun_out/pytest_rs.log
  for sp in 1 0; do
    PEARL_AMD_ROWSTEP_SPLIT=$sp timeout 600 python bench_algos.py --steps 300 --only ppo,bandit --cpu-seconds 0.5 > gpurun_out/bench_algos_rs$sp.jsonl 2> gpurun_out/bench_algos_rs$sp.err
    echo "bench_algos rowstep split=$sp rc=$?"; python - <<PY
import json
for ln in open("gpurun_out/bench_algos_rs$sp.jsonl"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["config"][:24], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us/step", [(k["kernel"][:14], round(k["avg_launch_us"],1), k["pipe"][:6]) for k in d.get("kernels",[])])
PY
  done
  timeout 300 python tools/prof_rowstep.py > gpurun_out/prof_rowstep_rs.txt 2>&1; grep -v amdgpu.ids gpurun_out/prof_rowstep_rs.txt | sed -n 3,16p
fi
if [ "$MODE" == "rs2" ]; then
  timeout 600 python -m pytest tests/test_gpu_actor_critic.py -m gpu -q --tb=short -p no:cacheprovider -x -s \
    -k "ppo_rowstep or ppo_learn_trajectory or ppo_preprocess or bandit_learn_batch or ppo_heads" > gpurun_out/pytest_rs2.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed|error vs float64|fp32 MFMA forward|bf16x3 forward|Error" gpurun_out/pytest_rs2.log | tail -20
fi
if [ "$MODE" == "rs3" ]; then
  timeout 600 python -m pytest tests/test_gpu_actor_critic.py tests/test_gpu_dp.py -m gpu -q --tb=short -p no:cacheprovider -x -s \
    -k "ppo_rowstep_bf16x3 or p2p" > gpurun_out/pytest_rs3.log 2>&1
  echo "pytest rc=$?"; grep -E "passed|failed|error vs float64|fp32 MFMA forward|bf16x3 forward|Error" gpurun_out/pytest_rs3.log | tail -20
fi
if [ "$MODE" == "bandit" ]; then
  cd /tmp && export TMPDIR=/tmp
  for w in bandit sac; do
    rm -rf $R/gpurun_out/prof_$w
    timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o t -- python $R/bench_algos.py --steps 200 --only $w --cpu-seconds 0.2 > $R/gpurun_out/rocprof_$w.log 2>&1
    DB=$(ls $R/gpurun_out/prof_$w/*.db $R/gpurun_out/prof_$w/*/*.db 2>/dev/null | head -1)
    python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/${w}_kernel_stats.txt 2>&1
    echo "rocprof $w rc=$?"; head -16 $R/gpurun_out/${w}_kernel_stats.txt | cut -c1-150
    python $R/tools/rocpd_timeline.py $DB mlp_rowstep 30 > $R/gpurun_out/${w}_timeline.txt 2>&1; head -40 $R/gpurun_out/${w}_timeline.txt | cut -c1-150
    rm -f $DB
  done
fi
if [ "$MODE" == "bandit2" ]; then
  timeout 600 python -m pytest tests/test_gpu_actor_critic.py -m gpu -q --tb=short -p no:cacheprovider -x \
    -k "bandit or linreg or squarecb or dsac_rowsteps or twin_adam or qnet" > gpurun_out/pytest_b2.log 2>&1
  echo "pytest rc=$?"; tail -5 gpurun_out/pytest_b2.log
  timeout 600 python -m pytest tests/test_gpu_dqn.py -m gpu -q --tb=short -p no:cacheprovider -x -k "qnet or generic or deep3 or dueling or multihead" > gpurun_out/pytest_b3.log 2>&1
  echo "pytest rc=$?"; tail -3 gpurun_out/pytest_b3.log
  timeout 600 python bench_algos.py --steps 300 --only bandit,dsac --cpu-seconds 0.5 > gpurun_out/bench_algos_b2.jsonl 2> gpurun_out/bench_algos_b2.err
  python - <<PY
import json
for ln in open("gpurun_out/bench_algos_b2.jsonl"):
    if ln.startswith("{"):
        d=json.loads(ln); print(d["config"][:30], round(d["value"]/1e6,2), "M", round(d["ms_per_step"]*1e3,1), "us/step", [(k["kernel"][:14], round(k["avg_launch_us"],1), k["pipe"][:6]) for k in d.get("kernels",[])])
PY
fi
if [ "$MODE" == "alltests" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log | cut -c1-300
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
fi
if [ "$MODE" == "t32" ]; then
  timeout 900 python -m pytest tests/test_gpu_dqn.py -m gpu -q --tb=short -p no:cacheprovider -x \
    -k "tile_shapes or q_values_and_targets or variants or overlapped_loop or generic_loop or fullbatch or double" > gpurun_out/pytest_t32.log 2>&1
  echo "pytest rc=$?"; tail -6 gpurun_out/pytest_t32.log | cut -c1-220
  for rows in 32 64; do
    PEARL_AMD_TARGET_ROWS=$rows timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > gpurun_out/bench_rows$rows.log 2> gpurun_out/bench_rows$rows.err
    echo "rows=$rows rc=$?"; tail -1 gpurun_out/bench_rows$rows.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print('value', round(d['value']/1e6,2), 'steady', round(d['steady_state']['value']/1e6,2), 'live frac', round(r['frac'],3), 'us', round(r['avg_launch_us'],1), 'isolated', round(r['isolated']['achieved'],1), 'TF frac_pipe', round(r['isolated'].get('frac_pipe',0),3))"
    PEARL_AMD_TARGET_ROWS=$rows timeout 300 python bench_algos.py --steps 200 --only double_dqn,dsac --cpu-seconds 0.3 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print('  ', d['config'][:40], round(d['value']/1e6,2), 'M')"
  done
fi
if [ "$MODE" == "bandit3" ]; then
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_bandit
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bandit -o t -- python $R/bench_algos.py --steps 200 --only bandit --cpu-seconds 0.2 > $R/gpurun_out/rocprof_bandit.log 2>&1
  DB=$(ls $R/gpurun_out/prof_bandit/*.db $R/gpurun_out/prof_bandit/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/bandit_kernel_stats.txt 2>&1
  head -14 $R/gpurun_out/bandit_kernel_stats.txt | cut -c1-150
  python $R/tools/rocpd_timeline.py $DB mlp_rowstep 26 > $R/gpurun_out/bandit_timeline.txt 2>&1; sed -n 5,32p $R/gpurun_out/bandit_timeline.txt | cut -c1-140
  rm -f $DB
  cd $R; TOPN=22 timeout 300 python tools/host_bound.py bandit 2>&1 | grep -v amdgpu | tail -34
fi
if [ "$MODE" == "bandit4" ]; then
  cd $R
  timeout 900 python -m pytest tests/test_gpu_actor_critic.py -q -x -k "bandit or squarecb or neural_linear" 2>&1 | tail -8
  timeout 300 python bench_algos.py --steps 300 --only bandit --cpu-seconds 0.2 2>$R/gpurun_out/bench_bandit4.err | tee $R/gpurun_out/bench_bandit4.jsonl | python tools/algo_line.py 2>/dev/null || tail -3 $R/gpurun_out/bench_bandit4.err
  TOPN=24 timeout 300 python tools/host_bound.py bandit 2>&1 | grep -v amdgpu | tail -36
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_bandit
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bandit -o t -- python $R/bench_algos.py --steps 200 --only bandit --cpu-seconds 0.2 > $R/gpurun_out/rocprof_bandit.log 2>&1
  DB=$(ls $R/gpurun_out/prof_bandit/*.db $R/gpurun_out/prof_bandit/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/bandit_kernel_stats.txt 2>&1
  head -14 $R/gpurun_out/bandit_kernel_stats.txt | cut -c1-150
  python $R/tools/rocpd_timeline.py $DB mlp_rowstep 26 > $R/gpurun_out/bandit_timeline.txt 2>&1; sed -n 5,24p $R/gpurun_out/bandit_timeline.txt | cut -c1-140
  rm -f $DB
fi
if [ "$MODE" == "bandit5" ]; then
  cd $R
  timeout 900 python -m pytest tests/test_gpu_actor_critic.py -q -x -k "bandit or squarecb or neural_linear or linreg" 2>&1 | tail -8
  timeout 300 python bench_algos.py --steps 300 --only bandit --cpu-seconds 0.2 2>$R/gpurun_out/bench_bandit5.err | tee $R/gpurun_out/bench_bandit5.jsonl | python tools/algo_line.py
  for w in ppo sac; do TOPN=14 timeout 300 python tools/host_bound.py $w 2>&1 | grep -v amdgpu | tail -24; done
fi
if [ "$MODE" == "ppo2" ]; then
  cd $R
  timeout 900 python -m pytest tests/test_gpu_actor_critic.py -q -x -k "ppo" 2>&1 | tail -8
  timeout 300 python bench_algos.py --steps 300 --only ppo --cpu-seconds 0.2 2>$R/gpurun_out/bench_ppo2.err | tee $R/gpurun_out/bench_ppo2.jsonl | python tools/algo_line.py
  PEARL_AMD_AC_LOOP=0 timeout 300 python bench_algos.py --steps 300 --only ppo --cpu-seconds 0.2 2>/dev/null | python tools/algo_line.py
  TOPN=10 timeout 300 python tools/host_bound.py ppo 2>&1 | grep -v amdgpu | tail -18
fi
if [ "$MODE" == "ppo3" ]; then
  cd $R
  timeout 900 python -m pytest tests/test_gpu_actor_critic.py -q -x -s -k "ppo or rowstep" 2>&1 | grep -v "^$" | tail -22
  for m in 1 0; do
    PEARL_AMD_ROWSTEP_SPLIT_BWD=$m timeout 300 python bench_algos.py --steps 300 --only ppo --cpu-seconds 0.2 2>$R/gpurun_out/bench_ppo3_$m.err | tee $R/gpurun_out/bench_ppo3_$m.jsonl | python tools/algo_line.py
  done
fi
if [ "$MODE" == "slots" ]; then
  cd $R
  for sl in 256 576 1152 2304; do
    echo "== PEARL_AMD_DW_SLOTS=$sl"
    PEARL_AMD_DW_SLOTS=$sl timeout 300 python bench_algos.py --steps 300 --only ppo,bandit --cpu-seconds 0.2 2>/dev/null | python tools/algo_line.py
  done
fi
if [ "$MODE" == "hb2" ]; then
  cd $R
  for w in dsac iql ddqn td3; do TOPN=1 timeout 300 python tools/host_bound.py $w 200 2>&1 | grep "host enqueue"; done
fi
if [ "$MODE" == "dsacprof" ]; then
  cd /tmp && export TMPDIR=/tmp
  for w in dsac; do
    rm -rf $R/gpurun_out/prof_$w
    timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o t -- python $R/bench_algos.py --steps 200 --only $w --cpu-seconds 0.2 > $R/gpurun_out/rocprof_$w.log 2>&1
    DB=$(ls $R/gpurun_out/prof_$w/*.db $R/gpurun_out/prof_$w/*/*.db 2>/dev/null | head -1)
    python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/${w}_kernel_stats.txt 2>&1
    head -24 $R/gpurun_out/${w}_kernel_stats.txt | cut -c1-150
    python $R/tools/rocpd_timeline.py $DB dsac_actor 40 > $R/gpurun_out/${w}_timeline.txt 2>&1; sed -n 5,50p $R/gpurun_out/${w}_timeline.txt | cut -c1-150
    rm -f $DB
  done
fi
if [ "$MODE" == "dsac2" ]; then
  cd $R
  timeout 900 python -m pytest tests/test_gpu_actor_critic.py tests/test_gpu_replay.py -q -x -k "sac or gather or replay" 2>&1 | tail -8
  timeout 300 python bench_algos.py --steps 300 --only dsac --cpu-seconds 0.2 2>$R/gpurun_out/bench_dsac2.err | tee $R/gpurun_out/bench_dsac2.jsonl | python tools/algo_line.py
  PEARL_AMD_AC_LOOP=0 PEARL_AMD_DSAC_ONE_CALL=0 timeout 300 python bench_algos.py --steps 300 --only dsac --cpu-seconds 0.2 2>/dev/null | python tools/algo_line.py
fi
if [ "$MODE" == "iql2" ]; then
  cd $R
  timeout 900 python -m pytest tests/test_gpu_actor_critic.py -q -x -k "iql" 2>&1 | tail -8
  for w in iql; do TOPN=1 timeout 300 python tools/host_bound.py $w 200 2>&1 | grep "host enqueue"; PEARL_AMD_AC_LOOP=0 PEARL_AMD_IQL_ONE_CALL=0 TOPN=1 timeout 300 python tools/host_bound.py $w 200 2>&1 | grep "host enqueue"; done
fi
if [ "$MODE" == "stress2" ]; then
  cd $R
  PYTORCH_NO_CUDA_MEMORY_CACHING=1 AMD_SERIALIZE_KERNEL=3 timeout 900 python -m pytest tests/test_gpu_actor_critic.py tests/test_gpu_kernels.py tests/test_gpu_dp.py -m gpu -q -x -p no:cacheprovider \
    -k "rowstep or weight_grad or bandit_learn_batch or ppo_learn_trajectory or p2p_exchange_sums or one_call or native_learn_loop or two_solves" > gpurun_out/nocache_tests.txt 2>&1
  echo "stress tests rc=$?"; tail -3 gpurun_out/nocache_tests.txt
  PEARL_AMD_P2P=1 PEARL_AMD_FORCE_DP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs \
    > gpurun_out/bench_dp1_p2p.log 2> gpurun_out/bench_dp1_p2p.err
  echo "bench dp1 p2p rc=$?"; tail -1 gpurun_out/bench_dp1_p2p.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('comm'))"
fi
if [ "$MODE" == "ddqnprof" ]; then
  cd /tmp && export TMPDIR=/tmp
  w=double_dqn
  rm -rf $R/gpurun_out/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o t -- python $R/bench_algos.py --steps 200 --only $w --cpu-seconds 0.2 > $R/gpurun_out/rocprof_$w.log 2>&1
  DB=$(ls $R/gpurun_out/prof_$w/*.db $R/gpurun_out/prof_$w/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/${w}_kernel_stats.txt 2>&1
  head -16 $R/gpurun_out/${w}_kernel_stats.txt | cut -c1-150
  python $R/tools/rocpd_timeline.py $DB weight_grad 40 > $R/gpurun_out/${w}_timeline.txt 2>&1; sed -n 1,50p $R/gpurun_out/${w}_timeline.txt | cut -c1-150
  rm -f $DB
fi
if [ "$MODE" == "ddqn2" ]; then
  cd $R
  timeout 900 python -m pytest tests/test_gpu_dqn.py -q -x -k "double or ddqn or qnet or sarsa" 2>&1 | tail -5
  for w in 1 0; do PEARL_AMD_DDQN_OVERLAP=$w timeout 300 python bench_algos.py --steps 300 --only double_dqn --cpu-seconds 0.2 2>/dev/null | python tools/algo_line.py; done
fi
if [ "$MODE" == "final" ]; then
  cd $R
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value', round(d['value']/1e6,2), 'M  steady', round(d.get('steady_state',{}).get('value',0)/1e6,2), ' frac', round(d['roofline']['frac'],3), 'frac_pipe', round(d['roofline'].get('frac_pipe',0),3), 'traffic src', d['roofline'].get('traffic_source'))
for r in d.get('other_configs',[]): print(' ', r.get('config'), round(r.get('value',0)/1e6,2), 'M', 'step_frac', round(r.get('step_frac',0),3), 'kernel_us', round(r.get('kernel_us',0),1), 'cpu', round((r.get('cpu_baseline') or {}).get('value',0)))"
  timeout 600 python bench_algos.py --steps 300 --cpu-seconds 2 > gpurun_out/bench_algos.jsonl 2> gpurun_out/bench_algos.err
  echo "bench_algos rc=$?"; python tools/algo_line.py < gpurun_out/bench_algos.jsonl
fi
if [ "$MODE" == "chainsplit" ]; then
  cd $R
  timeout 600 python -m pytest tests/test_gpu_kernels.py -q -x -s -k "weight_grad" 2>&1 | tail -12
  for cfg in "PEARL_AMD_DW_SPLIT=2" "default"; do
    echo "== $cfg"
    if [ "$cfg" == "default" ]; then E=""; else E="$cfg"; fi
    env $E timeout 300 python bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('value', round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,2), 'us/round', 'final_loss', d['config'].get('final_loss'))"
  done
fi
if [ "$MODE" == "split32tests" ]; then
  cd $R
  timeout 1500 python -m pytest tests/test_gpu_dqn.py tests/test_gpu_dp.py tests/test_reference_binding.py tests/test_gpu_kernels.py -q -x --tb=short 2>&1 | tail -15
  timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('value', round(d['value']/1e6,2), 'steady', round(d['steady_state']['value']/1e6,2))"
fi
if [ "$MODE" == "prof2" ]; then
  cd $R
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err; cat gpurun_out/shortcall.jsonl
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof
  timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs > $R/gpurun_out/rocprof.log 2>&1
  echo "rocprof rc=$?"; tail -1 $R/gpurun_out/rocprof.log | cut -c1-200
  python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats.txt 2>&1
  python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db target_split 60 >> $R/gpurun_out/kernel_stats.txt 2>&1
  head -12 $R/gpurun_out/kernel_stats.txt | cut -c1-150
  rm -f $R/gpurun_out/prof/*.db
fi
if [ "$MODE" == "prof3" ]; then
  cd $R
  timeout 600 python bench_algos.py --steps 300 --cpu-seconds 2 > gpurun_out/bench_algos.jsonl 2> gpurun_out/bench_algos.err
  echo "bench_algos rc=$?"; python tools/algo_line.py < gpurun_out/bench_algos.jsonl
  cd /tmp && export TMPDIR=/tmp
  for w in sac ppo bandit dsac double_dqn; do
    rm -rf $R/gpurun_out/prof_$w
    timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o t -- python $R/bench_algos.py --steps 200 --only $w --cpu-seconds 0.5 > $R/gpurun_out/rocprof_$w.log 2>&1
    DB=$(ls $R/gpurun_out/prof_$w/*.db $R/gpurun_out/prof_$w/*/*.db 2>/dev/null | head -1)
    python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/${w}_kernel_stats.txt 2>&1
    echo "rocprof $w rc=$?"; head -5 $R/gpurun_out/${w}_kernel_stats.txt | cut -c1-150
    rm -f $DB
  done
fi
if [ "$MODE" == "bandit6" ]; then
  cd $R
  timeout 900 python -m pytest tests/test_gpu_actor_critic.py -q -x -k "bandit" 2>&1 | tail -8
fi
if [ "$MODE" == "gae" ]; then
  cd $R
  timeout 900 python -m pytest tests/test_gpu_actor_critic.py tests/test_gpu_kernels.py -q -x -k "ppo or gae" 2>&1 | tail -4
  timeout 300 python bench_algos.py --steps 100 --only ppo --cpu-seconds 0.2 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(round(d['value']/1e6,2), 'M', d.get('preprocess_replay_buffer'))"
fi
if [ "$MODE" == "gaeprof" ]; then
  cd /tmp && export TMPDIR=/tmp
  w=ppo
  rm -rf $R/gpurun_out/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o t -- python $R/bench_algos.py --steps 50 --only $w --cpu-seconds 0.2 > $R/gpurun_out/rocprof_$w.log 2>&1
  DB=$(ls $R/gpurun_out/prof_$w/*.db $R/gpurun_out/prof_$w/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/${w}_kernel_stats2.txt 2>&1
  head -22 $R/gpurun_out/${w}_kernel_stats2.txt | cut -c1-150
  rm -f $DB
fi
if [ "$MODE" == "final2" ]; then
  cd $R
  timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
  echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value', round(d['value']/1e6,2), 'M  steady', round(d.get('steady_state',{}).get('value',0)/1e6,2), ' frac', round(d['roofline']['frac'],3), 'frac_pipe', round(d['roofline'].get('frac_pipe',0),3))
for r in d.get('other_configs',[]): print(' ', r.get('config'), round(r.get('value',0)/1e6,2), 'M', 'step_frac', round(r.get('step_frac',0),3), 'kernel_us', round(r.get('kernel_us',0),1), 'cpu', round((r.get('cpu_baseline') or {}).get('value',0)))"
fi
if [ "$MODE" == "ppo4" ]; then
  cd $R
  for st in 100 300 100; do timeout 300 python bench_algos.py --steps $st --only ppo --cpu-seconds 0.2 2>/dev/null | python tools/algo_line.py; done
fi
if [ "$MODE" == "dp" ]; then
  cd $R
  timeout 900 python -m pytest tests/test_gpu_dp.py tests/test_vector_env.py tests/test_gpu_replay.py -m gpu -q -x 2>&1 | tail -4
fi
