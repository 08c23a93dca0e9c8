#!/bin/bash
# Timeline of one steady-state L-BFGS iteration replayed from the hipGraph WITH the idle gaps in front of every launch (median over the last iterations
# of a kernel trace):   tools/trace_gaps.sh OUTDIR [bench args]      e.g.  tools/trace_gaps.sh gpurun_out/x --model nin --steps 130
R=$PWD; O=$R/$1; shift; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/t -o p -- python3 $R/bench.py --steps 12 --warmup 2 --no_prefill --no_cpu_baseline --no_extra_sizes --no_exact_split --no_repeats --hip_graph "$@" > /dev/null 2>&1
cd $R
python - "$O" "$@" <<'PY'
import csv, glob, statistics, sys
K = 12
for i, a in enumerate(sys.argv):
    if a == "--steps":
        K = int(sys.argv[i + 1])
f = glob.glob(f"{sys.argv[1]}/t/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "lbfgs_combine" in r["Kernel_Name"]]
# bench.py runs K graph replays (the timed region), then K eager iterations with per-launch events: take the last replays
last = len(idx) - 1 - K
its = [rows[idx[k]:idx[k + 1] + 1] for k in range(last - 9, last - 1)]  # from the previous combine (its end = this iteration's start)
n = len(its[0])
assert all(len(i) == n for i in its), [len(i) for i in its]
busy = idle = 0.0
for j in range(1, n):
    d = statistics.median((int(i[j]["End_Timestamp"]) - int(i[j]["Start_Timestamp"])) / 1e3 for i in its)
    g = statistics.median((int(i[j]["Start_Timestamp"]) - int(i[j - 1]["End_Timestamp"])) / 1e3 for i in its)
    busy += d; idle += g
    name = its[0][j]["Kernel_Name"].replace("maua::", "").replace("void ", "").split("(")[0][:44]
    print(f"{name:44s} {its[0][j]['Grid_Size_X'] + 'x' + its[0][j]['Grid_Size_Y'] + 'x' + its[0][j]['Grid_Size_Z']:>16s}  gap {g:6.1f}  run {d:7.1f}")
print(f"launches {n - 1}  busy {busy:.1f} us  idle {idle:.1f} us  iteration {busy + idle:.1f} us")
PY
rm -rf $O/t
