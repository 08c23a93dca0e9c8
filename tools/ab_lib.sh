#!/bin/bash
# A/B of two library builds on one box (alternating rounds): tools/ab_lib.sh LIB_A LIB_B "sizes" [steps]
a=$1; b=$2; sizes=${3:-"1024"}; steps=${4:-200}
for i in 1 2 3; do
for s in $sizes; do
for l in $a $b; do
MAUA_HIP_LIB=$l python bench.py --size $s --steps $steps --no_cpu_baseline --no_exact_split --no_repeats --no_extra_sizes --no_accuracy_probe 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('size $s', '$l'.split('/')[-1], d['value'], 'frac', d['roofline']['frac'])"
done; done; done
