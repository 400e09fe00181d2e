cd $GRAFT_REPO_ROOT
timeout 600 python bench_algos.py --only gather --steps 30 2>&1 | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['learn_loop_form']
print('gather all columns: %.1f GB/s median (%.1f best) frac %.3f of 8 TB/s, %.3f of 6.3; %d B/row, %.1f MB/launch, %.3f ms' % (r['achieved'], r['best_GBps'], r['frac'], r['frac_of_achievable_6300'], r['bytes_per_transition'], r['launch_bytes']/1e6, d['ms_per_step']))
print('learn-loop form: %.1f GB/s (%.1f best), %d B/row, %.3f ms' % (l['GBps'], l['best_GBps'], l['bytes_per_transition'], l['ms']))"
PEARL_AMD_FORCE_DP=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 2000 --warmup 200 --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('dp1 rccl', round(d['value']/1e6,2), d.get('comm'))"
timeout 900 python -m pytest tests/test_gpu_dp.py -m gpu -q --tb=short -p no:cacheprovider 2>&1 | tail -3
