for i in 1 2; do
for v in 0 1; do
for s in 512 256; do
MAUA_FUSE_POOL_SPLIT=$v python bench.py --size $s --steps 200 --no_cpu_baseline --no_exact_split --no_repeats --no_extra_sizes 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('pool_in_finish=$v size=$s', d['value'])"
done; done; done
