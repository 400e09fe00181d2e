#!/bin/bash
# Focused GPU-box calls of round 3 (gpurun -- bash tools/gpu_call.sh <mode>); output under
# gpurun_out/.  scripts_gpu_round.sh is the full round; these are the smaller experiments whose
# results are kept under profiles/r03_*:
#   tests   gpu test-suite + smoke + the driver's bench line
#   pmc     separate rocprofv3 --pmc passes over the single-stream loop (also: scripts_gpu_round.sh pmc)
#   chain   phase stamps of the two chain kernels + kernel boundary vs grid barrier (r03_g_*)
#   feeder  batched-observe tests and bench line (r03_h_*)
#   dsac    discrete SAC: parity tests, bench line, kernel stats
#   rowstep gpu suite + PPO / discrete SAC with and without the fused row step (r03_j_*)
#   variants  target-kernel builds A/B: bf16x3 split (default) | fp32 MFMA | U formed in the tile
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
MODE=${1:-tests}
line() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,2),'M tr/s', round(d['ms_per_step']*1e3,2),'us/round  steady', round(d.get('steady_state',{}).get('value',0)/1e6,2),'M  target launches timed', d['roofline']['launches_timed'])"; }
stats() {  # <name> <command...>: rocprofv3 kernel trace of the command -> gpurun_out/<name>_kernel_stats.txt
  local name=$1; shift
  ( cd /tmp && export TMPDIR=/tmp && rm -rf $R/gpurun_out/prof_$name &&
    timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$name -o t -- "$@" > $R/gpurun_out/rocprof_$name.log 2>&1 )
  local db=$(ls $R/gpurun_out/prof_$name/*.db $R/gpurun_out/prof_$name/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $db > $R/gpurun_out/${name}_kernel_stats.txt 2>&1
  head -24 $R/gpurun_out/${name}_kernel_stats.txt | cut -c1-150
  rm -f $db
}
if [ "$MODE" == "tests" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; line gpurun_out/bench_s20.log
fi
if [ "$MODE" == "pmc" ]; then
  # separate counter passes (kernel trace + --pmc only) over the single-stream loop
  export PEARL_AMD_OVERLAP=0
  cd /tmp && export TMPDIR=/tmp
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf $R/gpurun_out/pmc_$tag
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$tag -o dqn --output-format csv -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-other-configs --timing-level 0 > $R/gpurun_out/pmc_$tag.log 2>&1
    echo "pmc $tag rc=$?"
  done
  F=$(ls $R/gpurun_out/pmc_FETCH_SIZE/*counter_collection.csv $R/gpurun_out/pmc_FETCH_SIZE/*/*counter_collection.csv 2>/dev/null | head -1)
  W=$(ls $R/gpurun_out/pmc_WRITE_SIZE/*counter_collection.csv $R/gpurun_out/pmc_WRITE_SIZE/*/*counter_collection.csv 2>/dev/null | head -1)
  TK=target_h2_kernel; PL="262 KB"
  if [ "$PEARL_AMD_TARGET_H2" == "0" ]; then TK=target_split_kernel; PL="393 KB"; fi
  python $R/tools/pmc_traffic.py $F $W --kernel $TK --transitions 10240 --algorithmic 1033 --note "U row 1024 + reward 4 + term 1 + y 4 per transition; the shared [A, AD] action table and W2' planes ($PL) stay in L2" > $R/gpurun_out/pmc_target.json; cat $R/gpurun_out/pmc_target.json
  python $R/tools/pmc_traffic.py $F $W --kernel gather_kernel --transitions 10240 --algorithmic 2130 --note "window gather: next_state + reward + term read and written, state + action read, x written" > $R/gpurun_out/pmc_gather.json; cat $R/gpurun_out/pmc_gather.json
  python $R/tools/pmc_summary.py --chain-json $R/gpurun_out/pmc_chain.json $(ls $R/gpurun_out/pmc_*/*counter_collection.csv $R/gpurun_out/pmc_*/*/*counter_collection.csv 2>/dev/null) > $R/gpurun_out/pmc_summary.txt 2>&1; head -40 $R/gpurun_out/pmc_summary.txt | cut -c1-260; cat $R/gpurun_out/pmc_chain.json
  rm -f $R/gpurun_out/pmc_*/*kernel_trace.csv
fi
if [ "$MODE" == "chain" ]; then
  for r in 13 14 26; do
    PROF_ROUND=$r timeout 300 python tools/prof_chain.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_chain_r$r.txt
  done
  cat gpurun_out/prof_chain_r13.txt; grep -h "kernel boundary" gpurun_out/prof_chain_r14.txt gpurun_out/prof_chain_r26.txt
  timeout 120 tools/chain_boundary_bench 400 > gpurun_out/chain_boundary_bench.txt 2>&1
  cat gpurun_out/chain_boundary_bench.txt
fi
if [ "$MODE" == "feeder" ]; then
  timeout 600 python -m pytest tests/test_vector_env.py -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_vec.log 2>&1
  echo "pytest rc=$?"; tail -5 gpurun_out/pytest_vec.log
  timeout 600 python bench_algos.py --only feeder --steps 200 > gpurun_out/bench_feeder.jsonl 2> gpurun_out/bench_feeder.err
  echo "feeder rc=$?"; cut -c1-900 gpurun_out/bench_feeder.jsonl
fi
if [ "$MODE" == "dsac" ]; then
  timeout 900 python -m pytest tests/test_gpu_actor_critic.py -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_ac.log 2>&1
  echo "pytest rc=$?"; tail -5 gpurun_out/pytest_ac.log
  timeout 600 python bench_algos.py --only dsac --steps 200 > gpurun_out/bench_dsac.jsonl 2> gpurun_out/bench_dsac.err
  echo "dsac rc=$?"; cut -c1-400 gpurun_out/bench_dsac.jsonl
  stats dsac python $R/bench_algos.py --steps 100 --only dsac --cpu-seconds 0.3
fi
if [ "$MODE" == "variants" ]; then
  for cfg in "split PEARL_AMD_TARGET_SPLIT=1" "fp32 PEARL_AMD_TARGET_SPLIT=0" "fused_u PEARL_AMD_FUSE_U=1"; do
    set -- $cfg
    env $2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_$1.log 2> gpurun_out/bench_$1.err
    echo "$1 rc=$?"; line gpurun_out/bench_$1.log
  done
fi
if [ "$MODE" == "rowstep" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log
  for rs in 1 0; do
    PEARL_AMD_ROWSTEP=$rs timeout 600 python bench_algos.py --only ppo,dsac --steps 300 --cpu-seconds 1 > gpurun_out/bench_rowstep$rs.jsonl 2>/dev/null
    echo "ROWSTEP=$rs"; python -c "
import json
for l in open('gpurun_out/bench_rowstep$rs.jsonl'):
    d=json.loads(l); print(' ', d['config'][:50], round(d['value']/1e6,2), 'M', round(d['ms_per_step']*1e3,1), 'us')"
  done
  stats ppo_rowstep python $R/bench_algos.py --steps 200 --only ppo --cpu-seconds 0.3
fi
