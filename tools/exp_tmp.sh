cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_replay.py -m gpu -q --tb=short -p no:cacheprovider -k "large_gathers or checkpoint or state_dict" 2>&1 | tail -2
timeout 600 python bench_algos.py --only gather --steps 30 2>&1 | grep '^{' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']; l=d['learn_loop_form']
print('gather all columns: %.1f GB/s median (%.1f best) frac %.3f of 8 TB/s, %.3f of 6.3; %.3f ms' % (r['achieved'], r['best_GBps'], r['frac'], r['frac_of_achievable_6300'], d['ms_per_step']))
print('learn-loop form: %.1f GB/s (%.1f best), %.3f ms' % (l['GBps'], l['best_GBps'], l['ms']))"
