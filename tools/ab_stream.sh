for i in 1 2; do for v in 0 1; do
MAUA_STYLE_STREAM=$v python bench.py --size 2048 --steps 40 --no_cpu_baseline --no_exact_split --no_repeats --no_extra_sizes 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('vgg 2048 stream=$v', d['value'])"
MAUA_STYLE_STREAM=$v python bench.py --model nin --steps 200 --no_cpu_baseline --no_exact_split --no_repeats --no_extra_sizes 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('nin 1024 stream=$v', d['value'])"
done; done
