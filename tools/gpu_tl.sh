cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof
PEARL_AMD_ROWPASS_PAIR=1 PEARL_AMD_PAIR_LDS=0 PEARL_AMD_PERSIST_OFFER=4096 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof -o dqn -- python $R/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs > $R/gpurun_out/rocprof.log 2>&1
python $R/tools/rocpd_summary.py $R/gpurun_out/prof/dqn_results.db > $R/gpurun_out/kernel_stats_pair2.txt 2>&1
python $R/tools/rocpd_timeline.py $R/gpurun_out/prof/dqn_results.db rowpass 70 >> $R/gpurun_out/kernel_stats_pair2.txt 2>&1
head -9 $R/gpurun_out/kernel_stats_pair2.txt | cut -c1-150
grep "stream=" $R/gpurun_out/kernel_stats_pair2.txt | tail -64 | cut -c1-130
rm -f $R/gpurun_out/prof/*.db
