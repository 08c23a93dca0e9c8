#!/bin/bash
# Per-kernel durations of the 1024 x 1024 L-BFGS step in stream order (eager launches), as a short table: tools/prof_quick.sh OUTDIR [bench flags]
R=$PWD; O=$R/$1; shift; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o p -- python3 $R/bench.py --steps 20 --warmup 2 --no_cpu_baseline --no_extra_sizes --no_exact_split --no_repeats --no_hip_graph "$@" > $O/bench_under_rocprof.json 2>/dev/null
cd $R
rm -f $O/stats/*kernel_trace.csv
python - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/stats/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
it = [int(r["Calls"]) for r in rows if "lbfgs_pair_kernel" in r["Name"]][0]
for r in rows[:22]:
    print(f"{float(r['TotalDurationNs']) / 1e6 / it:7.3f} ms/it  {int(r['Calls']) / it:5.1f} x {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:110]}")
PY
