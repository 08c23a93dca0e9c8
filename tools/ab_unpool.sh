for i in 1 2; do
for f in 0 1; do
MAUA_FUSE_UNPOOL=$f python bench.py --steps 100 --no_cpu_baseline --no_exact_split --no_repeats 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('unpool=$f', d['value'], r['per_kernel_ms_per_step']['conv3x3_split_fwd'], r['per_kernel_ms_per_step']['conv3x3_split_bwd'], [(o.get('image_size'), o['iterations_per_s']) for o in d['extra']['other_sizes']])"
done; done
