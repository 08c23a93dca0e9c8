#!/bin/bash
# average duration of the kernels whose name matches $1 in a short eager bench run: tools/trace_kernels.sh PATTERN [bench args]
pat=$1; shift
cd /tmp; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; rm -rf /tmp/tk
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tk -o p -- python3 $R/bench.py --steps 10 --warmup 2 --no_prefill --no_cpu_baseline --no_extra_sizes --no_exact_split --no_repeats --no_hip_graph "$@" > /dev/null 2>&1
grep -h -E "$pat" /tmp/tk/*stats.csv /tmp/tk/*/*stats.csv 2>/dev/null | cut -c1-170
