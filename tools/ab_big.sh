for i in 1 2; do for v in 0 1; do
MAUA_IMAGE_GRAM=$v python bench.py --size 2048 --steps 40 --no_cpu_baseline --no_exact_split --no_repeats --no_extra_sizes 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('2048 image_gram=$v', d['value'])"
MAUA_IMAGE_GRAM=$v python bench.py --size 1448 --steps 60 --no_cpu_baseline --no_exact_split --no_repeats --no_extra_sizes 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('1448 image_gram=$v', d['value'])"
done; done
