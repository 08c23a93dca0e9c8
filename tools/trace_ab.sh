#!/bin/bash
# Per-dispatch timeline of one 1024 x 1024 L-BFGS iteration (median over the last iterations of the trace) for each value of an
# planner field, side by side:   tools/trace_ab.sh OUTDIR FIELD "v0 v1"   (FIELD = a name of plan.FIELDS, e.g. conv_x3p; it is set
# through MAUA_PLAN="FIELD=value" - any other MAUA_* variable would be ignored by the planner; a full MAUA_* name of an honoured
# variable, e.g. MAUA_CONV_X3, is exported as it is)
R=$PWD; O=$R/$1; VAR=$2; VALS=$3; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in $VALS; do
  case $VAR in
    MAUA_PLAN) export MAUA_PLAN=$v ;;
    MAUA_CONV_X3|MAUA_CONV_X6|MAUA_HIP_GRAPH|MAUA_DEBUG_POISON|MAUA_HOST_THREADS) export $VAR=$v ;;
    MAUA_*) echo "trace_ab.sh: $VAR is not an honoured variable; pass the planner field's name instead" >&2; exit 2 ;;
    *) export MAUA_PLAN="${BASE_PLAN:+$BASE_PLAN,}$VAR=$v" ;;
  esac
  rocprofv3 --kernel-trace --output-format csv -d $O/t_$v -o p -- python3 $R/bench.py --steps 12 --warmup 2 --no_prefill --no_cpu_baseline --no_extra_sizes --no_exact_split --no_repeats ${GRAPH:---no_hip_graph} $BENCH_ARGS > /dev/null 2>&1
done
cd $R
python - "$O" $VALS <<'PY'
import csv, glob, os, statistics, sys
out = {}
for v in sys.argv[2:]:
    f = glob.glob(f"{sys.argv[1]}/t_{v}/**/*kernel_trace.csv", recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "lbfgs_combine" in r["Kernel_Name"]]
    its = [rows[idx[k] + 1:idx[k + 1] + 1] for k in range(len(idx) - 9, len(idx) - 1)]
    n = len(its[0])
    assert all(len(i) == n for i in its), [len(i) for i in its]
    out[v] = [(its[0][j]["Kernel_Name"].replace("maua::", "").replace("void ", "").split("(")[0][:44], its[0][j]["Grid_Size_X"] + "x" + its[0][j]["Grid_Size_Y"] + "x" + its[0][j]["Grid_Size_Z"],
               statistics.median((int(i[j]["End_Timestamp"]) - int(i[j]["Start_Timestamp"])) / 1e3 for i in its)) for j in range(n)]
vals = sys.argv[2:]
if len({len(out[v]) for v in vals}) == 1:
    tot = [0.0] * len(vals)
    for j in range(len(out[vals[0]])):
        ds = [out[v][j][2] for v in vals]
        if max(ds) < float(os.environ.get('MIN_US', 30)):  # (MIN_US=0: every launch)
            continue
        for k, d in enumerate(ds):
            tot[k] += d
        print("  ".join(f"{out[v][j][0]:44s} {out[v][j][1]:>14s} {out[v][j][2]:7.1f}" for v in vals))
    print("sums of the listed launches:", [round(t, 1) for t in tot])
else:
    for v in vals:
        print("==", v, len(out[v]))
        for nme, g, d in out[v]:
            if d >= 30:
                print(f"{nme:44s} {g:>14s} {d:7.1f}")
PY
rm -rf $O/t_*
