cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_long_runs.py tests/test_reference_binding.py tests/test_gpu_replay.py -m gpu -q --tb=short -p no:cacheprovider -k "bandit or tracks or presampled" 2>&1 | tail -15
bash tools/gpu_r6.sh quick
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2; do
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_dw$m
PEARL_AMD_DEBUG_DW_ONLY=$m timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_dw$m -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 500 --warmup 50 --no-cpu-baseline --no-other-configs > $GRAFT_REPO_ROOT/gpurun_out/rocprof_dw$m.log 2>&1
DB=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_dw$m/*.db $GRAFT_REPO_ROOT/gpurun_out/prof_dw$m/*/*.db 2>/dev/null | head -1)
echo "== rocprof DW_ONLY=$m"; python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $DB 2>&1 | head -7 | cut -c1-160
rm -f $DB
done
