#!/bin/bash
# A/B of one environment switch over image sizes on one box: tools/ab_sizes.sh VAR "v0 v1" "sizes" [steps]   (two alternating rounds)
var=$1; vals=${2:-"0 1"}; sizes=${3:-"512 724 1024 1448"}; steps=${4:-100}
for i in 1 2; do
for s in $sizes; do
for v in $vals; do
env $var=$v python bench.py --size $s --steps $steps --no_cpu_baseline --no_exact_split --no_repeats --no_extra_sizes --no_accuracy_probe 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('size $s $var=$v', d['value'], 'frac', d['roofline']['frac'])"
done; done; done
