#!/bin/bash
# GPU round (rounds 4-5): a fresh-box stress FIRST (every torch tensor its own page-granular hipMalloc,
# kernels serialized: a stray access faults with context instead of landing in allocator slack —
# DESIGN.md §14.0), the gpu test-suite, smoke, the driver's own bench command (20 steps; carries
# parity, other_configs and the reference CPU baselines), the long bench, per-call fixed cost of
# learn(), bench_algos, the 1-rank RCCL and P2P readiness runs, rocprofv3 kernel summaries + the
# short call's host / kernel timeline; "pmc" as $1 adds the counter passes.  Everything lands in
# gpurun_out/; copy what should be judged into profiles/ (tracked).
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
if [ "$SKIP_STRESS" != "1" ]; then
PYTORCH_NO_CUDA_MEMORY_CACHING=1 AMD_SERIALIZE_KERNEL=3 timeout 300 python tools/stress_ppo.py 8 > gpurun_out/nocache_stress.txt 2>&1
echo "stress rc=$?"; tail -2 gpurun_out/nocache_stress.txt
# (not tests/test_gpu_dqn.py: the overlapped DQN loop hands data between two streams inside a call,
#  which a serialized queue turns into an expired bounded wait — by construction, not a fault)
PYTORCH_NO_CUDA_MEMORY_CACHING=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -m pytest tests/test_gpu_actor_critic.py tests/test_gpu_kernels.py tests/test_gpu_dp.py -m gpu -q -x -p no:cacheprovider \
  -k "rowstep or weight_grad or bandit_learn_batch or ppo_learn_trajectory or p2p_exchange_sums or one_call or native_learn_loop or two_solves" > gpurun_out/nocache_tests.txt 2>&1
echo "stress tests rc=$?"; tail -2 gpurun_out/nocache_tests.txt
# round 5's new kernels that do not need the two-stream loop: the paired row pass (stand-alone step),
# LayerNorm / activation kernels of the generic engine, the CQL rows, caller-supplied optimizers
PYTORCH_NO_CUDA_MEMORY_CACHING=1 AMD_SERIALIZE_KERNEL=3 timeout 600 python -m pytest tests/test_gpu_dqn.py -m gpu -q -x -p no:cacheprovider \
  -k "qnet or paired or caller_supplied or conservative" >> gpurun_out/nocache_tests.txt 2>&1
echo "stress tests (round 5 kernels) rc=$?"; tail -2 gpurun_out/nocache_tests.txt
fi
if [ "$SKIP_TESTS" != "1" ]; then
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
# the long-run / dynamic-range tables of round 6 (printed by the tests themselves)
timeout 900 python -m pytest tests/test_gpu_long_runs.py tests/test_gpu_numerics_range.py tests/test_gpu_perf_report.py -m gpu -q --tb=short -p no:cacheprovider -s > gpurun_out/long_runs.txt 2>&1
echo "long runs rc=$?"; tail -2 gpurun_out/long_runs.txt
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('value', round(d['value']/1e6,2), 'M  steady', round(d.get('steady_state',{}).get('value',0)/1e6,2), ' frac', round(d['roofline']['frac'],3), 'frac_pipe', round(d['roofline'].get('frac_pipe',0),3))
print('parity', {k: (float('%.3g' % v) if isinstance(v, float) else v) for k, v in (d.get('parity') or {}).items() if k.startswith(('max_rel', 'hip', 'reference'))})
for r in d.get('other_configs',[]): print(' ', r.get('config'), round(r.get('value',0)/1e6,2), 'M', 'step_frac', round(r.get('step_frac',0),3), 'kernel_us', round(r.get('kernel_us',0),1), 'cpu', round((r.get('cpu_baseline') or {}).get('value',0)))
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('kind'))"
timeout 900 python bench.py $BENCH_ARGS --no-cpu-baseline --no-other-configs > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-400
timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
echo "shortcall rc=$?"; cat gpurun_out/shortcall.jsonl
if [ "$SKIP_ALGOS" != "1" ]; then
timeout 900 python bench_algos.py --steps 300 --cpu-seconds 2 > gpurun_out/bench_algos.jsonl 2> gpurun_out/bench_algos.err
echo "bench_algos rc=$?"; cut -c1-260 gpurun_out/bench_algos.jsonl
fi
# multi-GPU readiness on one GPU: the driver's torchrun command line with the data-parallel loop
# forced on through a 1-rank RCCL communicator, and the same through the one-shot P2P exchange
for mode in rccl p2p; do
  P2P=0; [ "$mode" == "p2p" ] && P2P=1
  PEARL_AMD_P2P=$P2P PEARL_AMD_FORCE_DP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs \
    > gpurun_out/bench_dp1_$mode.log 2> gpurun_out/bench_dp1_$mode.err
  echo "bench dp1 $mode rc=$?"; tail -1 gpurun_out/bench_dp1_$mode.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('comm'))"
done
PEARL_AMD_FORCE_DP=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29541 bench_algos.py --steps 200 --only ppo --cpu-seconds 1 \
  > gpurun_out/ppo_dp1.jsonl 2> gpurun_out/ppo_dp1.err
echo "ppo dp1 rc=$?"; grep '^{' gpurun_out/ppo_dp1.jsonl | cut -c1-300
cd /tmp && export TMPDIR=/tmp
for w in sac ppo bandit dsac; do
  rm -rf $R/gpurun_out/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o t -- python $R/bench_algos.py --steps 200 --only $w --cpu-seconds 0.5 > $R/gpurun_out/rocprof_$w.log 2>&1
  DB=$(ls $R/gpurun_out/prof_$w/*.db $R/gpurun_out/prof_$w/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/${w}_kernel_stats.txt 2>&1
  echo "rocprof $w rc=$?"; head -6 $R/gpurun_out/${w}_kernel_stats.txt | cut -c1-150
  rm -f $DB
done
rm -rf $R/gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof rc=$?"; tail -1 $R/gpurun_out/rocprof.log | cut -c1-200
python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats.txt 2>&1
python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db target_h2 60 >> $R/gpurun_out/kernel_stats.txt 2>&1
head -12 $R/gpurun_out/kernel_stats.txt | cut -c1-150
rm -f $R/gpurun_out/prof/*.db
rm -rf $R/gpurun_out/prof_sc
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_sc -o sc -- python $R/tools/shortcall.py --trace > $R/gpurun_out/rocprof_sc.log 2>&1
DB=$(ls $R/gpurun_out/prof_sc/*.db $R/gpurun_out/prof_sc/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_timeline.py $DB --last-call > $R/gpurun_out/shortcall_timeline.txt 2>&1
head -3 $R/gpurun_out/shortcall_timeline.txt
rm -f $DB
if [ "$1" == "pmc" ]; then
  cd $R && bash tools/gpu_call.sh pmc
fi
