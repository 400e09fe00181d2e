#!/bin/bash
# One GPU-box call of round 3 (gpurun -- bash tools/gpu_call.sh <mode>); output under gpurun_out/.
#   a: split-MFMA prototype + gpu tests + smoke + the driver's bench line + long bench + per-call cost
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R; mkdir -p gpurun_out
MODE=${1:-a}
if [ "$MODE" == "a" ]; then
  timeout 120 tools/split_mfma_bench > gpurun_out/split_mfma_bench.txt 2>&1; echo "split rc=$?"; cat gpurun_out/split_mfma_bench.txt
  timeout 900 python -m pytest tests -m gpu -q --tb=short -x -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | cut -c1-3000
  timeout 600 python bench.py --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err
  echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-1500
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
  echo "shortcall rc=$?"; cat gpurun_out/shortcall.jsonl
fi
if [ "$MODE" == "b" ]; then
  # the bf16x3 target kernel: parity suite, then throughput against the fp32 kernels and with more
  # CUs for the chain
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -30 gpurun_out/pytest_gpu.log
  run() { # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$name.log 2> gpurun_out/bench_$name.err
    echo "bench $name rc=$? $(tail -1 gpurun_out/bench_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2),'M tr/s  us/step',round(d['ms_per_step']*1e3,2),' target: us',round(r['avg_launch_us'],1),'tr/launch',r['transitions_per_launch'],'frac',round(r['frac'],3),' iso',round(r.get('isolated',{}).get('frac',0),3))" 2>&1)"
  }
  run fp32 PEARL_AMD_TARGET_SPLIT=0
  run split64 PEARL_AMD_TARGET_SPLIT=1
  run split96 PEARL_AMD_TARGET_SPLIT=1 PEARL_AMD_RESERVED_CUS=96
  run split128 PEARL_AMD_TARGET_SPLIT=1 PEARL_AMD_RESERVED_CUS=128
  run split128ks2 PEARL_AMD_TARGET_SPLIT=1 PEARL_AMD_RESERVED_CUS=128 PEARL_AMD_DW_MINB=1024
  run split112ks2 PEARL_AMD_TARGET_SPLIT=1 PEARL_AMD_RESERVED_CUS=112 PEARL_AMD_DW_MINB=1024
  run split64sf1 PEARL_AMD_TARGET_SPLIT=1 PEARL_AMD_SPLIT_FIRST=1
  run split64sf12 PEARL_AMD_TARGET_SPLIT=1 PEARL_AMD_SPLIT_FIRST=12
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | cut -c1-2600
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
  echo "shortcall rc=$?"; cat gpurun_out/shortcall.jsonl
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
  echo "rocprof rc=$?"
  python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats.txt 2>&1
  python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db target_split 60 >> $R/gpurun_out/kernel_stats.txt 2>&1
  head -14 $R/gpurun_out/kernel_stats.txt
  rm -f $R/gpurun_out/prof/*.db
fi
if [ "$MODE" == "c" ]; then
  # 32-row weight-gradient tiles with more CUs for the chain
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -30 gpurun_out/pytest_gpu.log
  run() { # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$name.log 2> gpurun_out/bench_$name.err
    echo "bench $name rc=$? $(tail -1 gpurun_out/bench_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2),'M tr/s  us/step',round(d['ms_per_step']*1e3,2),' target: us',round(r['avg_launch_us'],1),'tr/launch',r['transitions_per_launch'],'frac',round(r['frac'],3),' iso',round(r.get('isolated',{}).get('frac',0),3))" 2>&1)"
  }
  run tm64r64 PEARL_AMD_DW_TM=64
  run tm32r64 PEARL_AMD_DW_TM=32
  run tm32r112 PEARL_AMD_DW_TM=32 PEARL_AMD_RESERVED_CUS=112
  run tm32r128 PEARL_AMD_DW_TM=32 PEARL_AMD_RESERVED_CUS=128
  run tm32r144 PEARL_AMD_DW_TM=32 PEARL_AMD_RESERVED_CUS=144
  run tm32r160 PEARL_AMD_DW_TM=32 PEARL_AMD_RESERVED_CUS=160
  run tm64r128 PEARL_AMD_DW_TM=64 PEARL_AMD_RESERVED_CUS=128
  PEARL_AMD_DW_TM=32 PEARL_AMD_RESERVED_CUS=128 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof
  PEARL_AMD_DW_TM=32 PEARL_AMD_RESERVED_CUS=128 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
  echo "rocprof rc=$?"
  python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats.txt 2>&1
  python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db target_split 60 >> $R/gpurun_out/kernel_stats.txt 2>&1
  head -8 $R/gpurun_out/kernel_stats.txt
  rm -f $R/gpurun_out/prof/*.db
fi
if [ "$MODE" == "d" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -30 gpurun_out/pytest_gpu.log
  run() { # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$name.log 2> gpurun_out/bench_$name.err
    echo "bench $name rc=$? $(tail -1 gpurun_out/bench_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2),'M tr/s  us/step',round(d['ms_per_step']*1e3,2),' target: us',round(r['avg_launch_us'],1),'tr/launch',r['transitions_per_launch'],'frac',round(r['frac'],3),' iso',round(r.get('isolated',{}).get('frac',0),3))" 2>&1)"
  }
  run dflt X=1
  run nosplitrp PEARL_AMD_ROWPASS_SPLIT=0
  run leadp PEARL_AMD_LEAD_PERSIST=1
  run sf1 PEARL_AMD_SPLIT_FIRST=1
  run sf12 PEARL_AMD_SPLIT_FIRST=12
  run sf2 PEARL_AMD_SPLIT_FIRST=2
  run r128 PEARL_AMD_RESERVED_CUS=128
  run r64 PEARL_AMD_RESERVED_CUS=64 PEARL_AMD_DW_TM=32
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
  echo "shortcall rc=$?"; cat gpurun_out/shortcall.jsonl
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
  echo "rocprof rc=$?"
  python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats.txt 2>&1
  python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db target_split 70 >> $R/gpurun_out/kernel_stats.txt 2>&1
  head -9 $R/gpurun_out/kernel_stats.txt
  rm -f $R/gpurun_out/prof/*.db
fi
if [ "$MODE" == "e" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -30 gpurun_out/pytest_gpu.log
  run() { # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$name.log 2> gpurun_out/bench_$name.err
    echo "bench $name rc=$? $(tail -1 gpurun_out/bench_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2),'M tr/s  us/step',round(d['ms_per_step']*1e3,2),' target: us',round(r['avg_launch_us'],1),'tr/launch',r['transitions_per_launch'],'frac',round(r['frac'],3),' iso',round(r.get('isolated',{}).get('frac',0),3))" 2>&1)"
  }
  run dflt X=1
  run sf12 PEARL_AMD_SPLIT_FIRST=12
  run sf2 PEARL_AMD_SPLIT_FIRST=2
  run leadp PEARL_AMD_LEAD_PERSIST=1
  run r120 PEARL_AMD_RESERVED_CUS=120
  run r136 PEARL_AMD_RESERVED_CUS=136
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
  PEARL_AMD_SPLIT_FIRST=12 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20_sf12.log 2> gpurun_out/bench_s20_sf12.err
  echo "bench s20 sf12 rc=$?"; tail -1 gpurun_out/bench_s20_sf12.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
  echo "shortcall rc=$?"; cat gpurun_out/shortcall.jsonl
  timeout 300 python bench_algos.py --steps 200 --only ppo --cpu-seconds 1 > gpurun_out/ppo_single.jsonl 2> gpurun_out/ppo_single.err
  echo "ppo single rc=$?"; cut -c1-400 gpurun_out/ppo_single.jsonl
  PEARL_AMD_FORCE_DP=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench_algos.py --steps 200 --only ppo --cpu-seconds 1 > gpurun_out/ppo_dp1.jsonl 2> gpurun_out/ppo_dp1.err
  echo "ppo dp1 rc=$?"; cut -c1-600 gpurun_out/ppo_dp1.jsonl; tail -3 gpurun_out/ppo_dp1.err
fi
if [ "$MODE" == "f" ]; then
  # does device ASan work here at all?
  HSA_XNACK=1 timeout 60 tools/asan_probe > gpurun_out/asan_probe.txt 2>&1; echo "asan probe rc=$?"; tail -15 gpurun_out/asan_probe.txt | cut -c1-200
  timeout 900 bash tools/asan_run.sh python tools/stress_ppo.py 3 > gpurun_out/asan_stress_ppo.txt 2>&1; echo "asan stress rc=$?"; tail -25 gpurun_out/asan_stress_ppo.txt | cut -c1-250
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
  PEARL_AMD_LEAD_PERSIST=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20_lp.log 2> gpurun_out/bench_s20_lp.err
  echo "bench s20 leadp rc=$?"; tail -1 gpurun_out/bench_s20_lp.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state'))"
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
  echo "shortcall rc=$?"; cat gpurun_out/shortcall.jsonl
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_sc
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
  DB=$(ls $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/shortcall_timeline.txt 2>&1
  head -80 $R/gpurun_out/shortcall_timeline.txt | cut -c1-130
  rm -f $DB
fi
if [ "$MODE" == "g" ]; then
  # (1) poor man's sanitizer for the PPO fault: every torch tensor its own hipMalloc (page-granular,
  # unmapped neighbours) + serialized kernels; (2) standalone device-ASan probe; (3) bandit async solve
  export LD_LIBRARY_PATH=$(dirname $(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)):$LD_LIBRARY_PATH
  HSA_XNACK=1 timeout 60 tools/asan_probe > gpurun_out/asan_probe.txt 2>&1; echo "asan probe rc=$?"; tail -12 gpurun_out/asan_probe.txt | cut -c1-200
  PYTORCH_NO_CUDA_MEMORY_CACHING=1 AMD_SERIALIZE_KERNEL=3 timeout 900 python -X faulthandler tools/stress_ppo.py 12 > gpurun_out/nocache_stress_ppo.txt 2>&1; echo "nocache stress rc=$?"; tail -4 gpurun_out/nocache_stress_ppo.txt | cut -c1-200
  PYTORCH_NO_CUDA_MEMORY_CACHING=1 AMD_SERIALIZE_KERNEL=3 timeout 1500 python -m pytest tests/test_gpu_actor_critic.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/nocache_pytest_ac.log 2>&1; echo "nocache pytest ac rc=$?"; tail -5 gpurun_out/nocache_pytest_ac.log
  PYTORCH_NO_CUDA_MEMORY_CACHING=1 timeout 900 python bench_algos.py --steps 30 --only ppo,sac,td3,dsac,bandit --cpu-seconds 0.3 > gpurun_out/nocache_bench_algos.jsonl 2> gpurun_out/nocache_bench_algos.err; echo "nocache bench_algos rc=$?"; cut -c1-160 gpurun_out/nocache_bench_algos.jsonl; tail -3 gpurun_out/nocache_bench_algos.err
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
  timeout 300 python bench_algos.py --steps 300 --only bandit --cpu-seconds 1 > gpurun_out/bandit_async.jsonl 2> gpurun_out/bandit_async.err; echo "bandit rc=$?"; cut -c1-300 gpurun_out/bandit_async.jsonl
  PEARL_AMD_BANDIT_ASYNC_SOLVE=0 timeout 300 python bench_algos.py --steps 300 --only bandit --cpu-seconds 1 > gpurun_out/bandit_sync.jsonl 2> gpurun_out/bandit_sync.err; echo "bandit sync rc=$?"; cut -c1-300 gpurun_out/bandit_sync.jsonl
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_sc
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
  DB=$(ls $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/shortcall_timeline.txt 2>&1
  head -90 $R/gpurun_out/shortcall_timeline.txt | cut -c1-130
  rm -f $DB
fi
if [ "$MODE" == "pmc" ]; then
  # separate counter passes (kernel trace + --pmc only) over the single-stream loop
  export PEARL_AMD_OVERLAP=0
  cd /tmp && export TMPDIR=/tmp
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | cut -d' ' -f1)
    rm -rf $R/gpurun_out/pmc_$tag
    timeout 600 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmc_$tag -o dqn --output-format csv -- python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --timing-level 0 > $R/gpurun_out/pmc_$tag.log 2>&1
    echo "pmc $tag rc=$?"
  done
  F=$(ls $R/gpurun_out/pmc_FETCH_SIZE/*counter_collection.csv $R/gpurun_out/pmc_FETCH_SIZE/*/*counter_collection.csv 2>/dev/null | head -1)
  W=$(ls $R/gpurun_out/pmc_WRITE_SIZE/*counter_collection.csv $R/gpurun_out/pmc_WRITE_SIZE/*/*counter_collection.csv 2>/dev/null | head -1)
  python $R/tools/pmc_traffic.py $F $W --kernel target_split_kernel --transitions 10240 --algorithmic 1033 --note "U row 1024 + reward 4 + term 1 + y 4 per transition; the shared [A, AD] action table and W2' planes (393 KB) stay in L2" > $R/gpurun_out/pmc_target.json; cat $R/gpurun_out/pmc_target.json
  python $R/tools/pmc_traffic.py $F $W --kernel gather_kernel --transitions 10240 --algorithmic 2130 --note "window gather: next_state + reward + term read and written, state + action read, x written" > $R/gpurun_out/pmc_gather.json; cat $R/gpurun_out/pmc_gather.json
  python $R/tools/pmc_summary.py $(ls $R/gpurun_out/pmc_*/*counter_collection.csv $R/gpurun_out/pmc_*/*/*counter_collection.csv 2>/dev/null) > $R/gpurun_out/pmc_summary.txt 2>&1; head -40 $R/gpurun_out/pmc_summary.txt | cut -c1-260
  rm -f $R/gpurun_out/pmc_*/*kernel_trace.csv
fi
if [ "$MODE" == "h" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
  for i in 1 2 3; do
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20_$i.log 2> gpurun_out/bench_s20_$i.err
    echo "bench s20 #$i rc=$?"; tail -1 gpurun_out/bench_s20_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state',{}).get('value'))"
  done
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
  echo "shortcall rc=$?"; cut -c1-200 gpurun_out/shortcall.jsonl
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_sc
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
  DB=$(ls $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/shortcall_timeline.txt 2>&1
  head -24 $R/gpurun_out/shortcall_timeline.txt | cut -c1-130
  rm -f $DB
  cd $R
  bash tools/gpu_call.sh pmc
fi
if [ "$MODE" == "i" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -6 gpurun_out/pytest_gpu.log
  for i in 1 2; do
    timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20_$i.log 2> gpurun_out/bench_s20_$i.err
    echo "bench s20 #$i rc=$?"; tail -1 gpurun_out/bench_s20_$i.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state',{}).get('value'), d['roofline'].get('executed'))"
  done
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
  echo "shortcall rc=$?"; cut -c1-200 gpurun_out/shortcall.jsonl
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_sc
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
  DB=$(ls $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/shortcall_timeline.txt 2>&1
  head -20 $R/gpurun_out/shortcall_timeline.txt | cut -c1-130
  rm -f $DB
fi
if [ "$MODE" == "j" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log
  run() { # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$name.log 2> gpurun_out/bench_$name.err
    echo "bench $name rc=$? $(tail -1 gpurun_out/bench_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2),'M tr/s  us/step',round(d['ms_per_step']*1e3,2),' target: us',round(r['avg_launch_us'],1),'tr/launch',r['transitions_per_launch'],'frac',round(r['frac'],3),' iso',round(r.get('isolated',{}).get('frac',0),3))" 2>&1)"
  }
  run fuse X=1
  run nofuse PEARL_AMD_FUSE_U=0
  run fuse2 X=1
  for v in 1 0; do
    PEARL_AMD_FUSE_U=$v timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20_f$v.log 2> gpurun_out/bench_s20_f$v.err
    echo "bench s20 fuse=$v rc=$?"; tail -1 gpurun_out/bench_s20_f$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state',{}).get('value'))"
  done
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
  echo "shortcall rc=$?"; cut -c1-200 gpurun_out/shortcall.jsonl
  timeout 300 python bench_algos.py --steps 300 --only sac,td3,double_dqn --cpu-seconds 0.5 > gpurun_out/bench_algos_j.jsonl 2> gpurun_out/bench_algos_j.err; echo "algos rc=$?"; cut -c1-260 gpurun_out/bench_algos_j.jsonl
fi
if [ "$MODE" == "k" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log
  run() { # name, env...
    local name=$1; shift
    env "$@" timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$name.log 2> gpurun_out/bench_$name.err
    echo "bench $name rc=$? $(tail -1 gpurun_out/bench_$name.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value']/1e6,2),'M tr/s  us/step',round(d['ms_per_step']*1e3,2),' target: us',round(r['avg_launch_us'],1),'tr/launch',r['transitions_per_launch'],'frac',round(r['frac'],3),' iso',round(r.get('isolated',{}).get('frac',0),3), ' gather', round(r.get('gather',{}).get('avg_launch_us',0),1))" 2>&1)"
  }
  run hyb X=1
  run nofuse PEARL_AMD_FUSE_U=0
  run hyb2 X=1
  for v in 1 0 1; do
    PEARL_AMD_FUSE_U=$v timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20_f$v.log 2> gpurun_out/bench_s20_f$v.err
    echo "bench s20 fuse=$v rc=$?"; tail -1 gpurun_out/bench_s20_f$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state',{}).get('value'))"
  done
  timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
  echo "shortcall rc=$?"; cut -c1-200 gpurun_out/shortcall.jsonl
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
  python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats.txt 2>&1
  python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db target_split 40 >> $R/gpurun_out/kernel_stats.txt 2>&1
  head -12 $R/gpurun_out/kernel_stats.txt | cut -c1-150
  rm -f $R/gpurun_out/prof/*.db
fi
if [ "$MODE" == "l" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
  echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state',{}).get('value'), d['roofline']['launches_timed'], d['roofline'].get('gather'))"
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_dsac
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dsac -o t -- python $R/bench_algos.py --steps 100 --only dsac --cpu-seconds 0.3 > $R/gpurun_out/rocprof_dsac.log 2>&1
  DB=$(ls $R/gpurun_out/prof_dsac/*.db $R/gpurun_out/prof_dsac/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/dsac_kernel_stats.txt 2>&1
  head -24 $R/gpurun_out/dsac_kernel_stats.txt | cut -c1-150
  rm -f $DB
fi
if [ "$MODE" == "m" ]; then
  for ring in 8 16 32; do
    PEARL_AMD_DW_RING=$ring timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_ring$ring.log 2> gpurun_out/bench_ring$ring.err
    echo "ring $ring rc=$?"; tail -1 gpurun_out/bench_ring$ring.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state',{}).get('value'))"
  done
  timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "tile_shapes or overlapped or generic_loop or full_size or fixtures" > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
  cd /tmp && export TMPDIR=/tmp
  for ring in 16 32; do
    rm -rf $R/gpurun_out/prof_ring$ring
    PEARL_AMD_DW_RING=$ring timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_ring$ring -o t -- python $R/bench.py --gpus 1 --steps 400 --warmup 20 --no-cpu-baseline > $R/gpurun_out/rocprof_ring$ring.log 2>&1
    DB=$(ls $R/gpurun_out/prof_ring$ring/*.db $R/gpurun_out/prof_ring$ring/*/*.db 2>/dev/null | head -1)
    python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/ring${ring}_kernel_stats.txt 2>&1
    head -8 $R/gpurun_out/ring${ring}_kernel_stats.txt | cut -c1-150
    rm -f $DB
  done
fi
if [ "$MODE" == "n" ]; then
  for cfg in "8 0" "8 1" "16 1"; do
    set -- $cfg
    PEARL_AMD_DW_RING=$1 PEARL_AMD_DW_XCD_ORDER=$2 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/bench_x$1_$2.log 2> gpurun_out/bench_x$1_$2.err
    echo "ring $1 xcd $2 rc=$?"; tail -1 gpurun_out/bench_x$1_$2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('steady_state',{}).get('value'))"
  done
  timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x -k "tile_shapes or overlapped or generic_loop or full_size or fixtures or weight" > gpurun_out/pytest_gpu.log 2>&1
  echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_x
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_x -o t -- python $R/bench.py --gpus 1 --steps 400 --warmup 20 --no-cpu-baseline > $R/gpurun_out/rocprof_x.log 2>&1
  DB=$(ls $R/gpurun_out/prof_x/*.db $R/gpurun_out/prof_x/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/xcd_kernel_stats.txt 2>&1
  head -8 $R/gpurun_out/xcd_kernel_stats.txt | cut -c1-150
  rm -f $DB
fi
if [ "$MODE" == "o" ]; then
  for r in 13 25; do
    PROF_ROUND=$r timeout 300 python tools/prof_chain.py > gpurun_out/prof_chain_r$r.txt 2>&1
    cat gpurun_out/prof_chain_r$r.txt
  done
  PEARL_AMD_DW_XCD_ORDER=0 PROF_ROUND=13 timeout 300 python tools/prof_chain.py > gpurun_out/prof_chain_r13_noxcd.txt 2>&1
  tail -12 gpurun_out/prof_chain_r13_noxcd.txt
  PEARL_AMD_OVERLAP=0 PROF_ROUND=13 timeout 300 python tools/prof_chain.py > gpurun_out/prof_chain_r13_serial.txt 2>&1
  tail -12 gpurun_out/prof_chain_r13_serial.txt
fi
if [ "$MODE" == "p" ]; then
  timeout 120 tools/chain_boundary_bench 400 > gpurun_out/chain_boundary_bench.txt 2>&1
  cat gpurun_out/chain_boundary_bench.txt
  PROF_ROUND=13 timeout 300 python tools/prof_chain.py 2>&1 | grep -v amdgpu.ids > gpurun_out/prof_chain_r13.txt
  head -3 gpurun_out/prof_chain_r13.txt
  PROF_ROUND=14 timeout 300 python tools/prof_chain.py 2>&1 | grep "kernel boundary"
  PROF_ROUND=26 timeout 300 python tools/prof_chain.py 2>&1 | grep "kernel boundary"
fi
if [ "$MODE" == "s" ]; then
  timeout 600 python -m pytest tests/test_vector_env.py -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_vec.log 2>&1
  echo "pytest rc=$?"; tail -12 gpurun_out/pytest_vec.log
  timeout 600 python bench_algos.py --only feeder --steps 200 > gpurun_out/bench_feeder.jsonl 2> gpurun_out/bench_feeder.err
  echo "feeder rc=$?"; cat gpurun_out/bench_feeder.jsonl | cut -c1-900; tail -5 gpurun_out/bench_feeder.err
fi
