"""Print the per-dispatch timeline of the last complete iteration in a rocprofv3 kernel-trace CSV."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f))); rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marker = sys.argv[2] if len(sys.argv) > 2 else "lbfgs_combine"
idx = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
a, b = idx[-2] + 1, idx[-1] + 1
tot = 0
for r in rows[a:b]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; tot += d
    name = r["Kernel_Name"].replace("maua::", "").replace("void ", "").split("(")[0][:48]
    print("%8.1f us grid %8s,%4s %s" % (d, r["Grid_Size_X"], r["Grid_Size_Y"], name))
print("sum", round(tot, 1), "span", (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e3)
