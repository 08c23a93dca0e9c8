#!/bin/bash
# A/B of one environment switch on one box: tools/ab_env.sh VAR "v0 v1 ..." "sizes" [steps]
var=$1; vals=${2:-"0 1"}; sizes=${3:-"512 256"}; steps=${4:-200}
for i in 1 2; do
for v in $vals; do
for s in $sizes; do
env $var=$v python bench.py --size $s --steps $steps --no_cpu_baseline --no_exact_split --no_repeats --no_extra_sizes 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$var=$v size=$s', d['value'])"
done; done; done
