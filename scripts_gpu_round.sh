#!/bin/bash
# GPU round: gpu test-suite, smoke, the driver's own bench command (20 steps), the long bench with
# the CPU baseline, per-call fixed cost of learn(), bench_algos, rocprofv3 kernel trace + timeline
# of the driver's command, the 1-rank RCCL run + all-reduce latency, SAC / PPO kernel summaries;
# "pmc" as $1 adds the counter passes.  Everything lands in gpurun_out/;
# copy what should be judged into profiles/ (tracked).
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd $R
if [ "$SKIP_TESTS" != "1" ]; then
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
fi
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_s20.log 2> gpurun_out/bench_s20.err
echo "bench s20 rc=$?"; tail -1 gpurun_out/bench_s20.log | cut -c1-1400
timeout 900 python bench.py $BENCH_ARGS --no-cpu-baseline > gpurun_out/bench.log 2> gpurun_out/bench.err
echo "bench rc=$?"; tail -1 gpurun_out/bench.log | cut -c1-1400
PEARL_AMD_OVERLAP=0 timeout 600 python bench.py --timing-level 2 --no-cpu-baseline > gpurun_out/bench_l2.log 2>&1
echo "bench l2 rc=$?"; tail -1 gpurun_out/bench_l2.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d.get('stage_us'))"
timeout 300 python tools/shortcall.py > gpurun_out/shortcall.jsonl 2> gpurun_out/shortcall.err
echo "shortcall rc=$?"; cat gpurun_out/shortcall.jsonl
if [ "$SKIP_ALGOS" != "1" ]; then
timeout 600 python bench_algos.py --steps 300 > gpurun_out/bench_algos.jsonl 2> gpurun_out/bench_algos.err
echo "bench_algos rc=$?"; cut -c1-300 gpurun_out/bench_algos.jsonl
fi
# multi-GPU readiness on one GPU: the driver's torchrun command line with the data-parallel loop
# forced on (1-rank RCCL communicator: every round's gradient goes through ncclAllReduce), and
# the latency of that one collective
PEARL_AMD_FORCE_DP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline \
  > gpurun_out/bench_dp1.log 2> gpurun_out/bench_dp1.err
echo "bench dp1 rc=$?"; tail -1 gpurun_out/bench_dp1.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d.get('comm'))"
timeout 300 python tools/allreduce_latency.py > gpurun_out/allreduce_latency.json 2> gpurun_out/allreduce_latency.err
echo "allreduce rc=$?"; cat gpurun_out/allreduce_latency.json
# PPO (BASELINE config 4) through the driver-style torchrun line with a 1-rank RCCL communicator:
# actor + critic gradients as ONE ncclAllReduce per step
PEARL_AMD_FORCE_DP=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29541 bench_algos.py --steps 200 --only ppo --cpu-seconds 1 \
  > gpurun_out/ppo_dp1.jsonl 2> gpurun_out/ppo_dp1.err
echo "ppo dp1 rc=$?"; grep '^{' gpurun_out/ppo_dp1.jsonl | cut -c1-300
# the fused SAC step: in-kernel phase stamps
timeout 300 python tools/prof_sac.py > gpurun_out/prof_sac.txt 2>&1
echo "prof_sac rc=$?"; grep -E "whole launch|^end" gpurun_out/prof_sac.txt
cd /tmp && export TMPDIR=/tmp
for w in sac ppo; do
  rm -rf $R/gpurun_out/prof_$w
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o t -- python $R/bench_algos.py --steps 200 --only $w --cpu-seconds 0.5 > $R/gpurun_out/rocprof_$w.log 2>&1
  DB=$(ls $R/gpurun_out/prof_$w/*.db $R/gpurun_out/prof_$w/*/*.db 2>/dev/null | head -1)
  python $R/tools/rocpd_summary.py $DB > $R/gpurun_out/${w}_kernel_stats.txt 2>&1
  echo "rocprof $w rc=$?"; head -8 $R/gpurun_out/${w}_kernel_stats.txt
  rm -f $DB
done
rm -rf $R/gpurun_out/prof
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline > $R/gpurun_out/rocprof.log 2>&1
echo "rocprof rc=$?"; tail -1 $R/gpurun_out/rocprof.log | cut -c1-200
python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats.txt 2>&1
python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db target_split 60 >> $R/gpurun_out/kernel_stats.txt 2>&1
head -12 $R/gpurun_out/kernel_stats.txt
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
