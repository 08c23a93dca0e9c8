"""rocprofv3 kernel_stats.csv -> compact per-iteration table.   python tools/prof_table.py STATS.csv ITERATIONS [TOP]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total GPU time {tot / 1e6:.1f} ms over {iters} iterations = {tot / 1e6 / iters:.3f} ms / iteration")
for r in rows[:top]:
    name = r["Name"].split("(")[0].replace("void ", "").replace("maua::", "")[:60]
    print(f"{name:60s} calls/it {int(r['Calls']) / iters:6.2f}  avg {float(r['AverageNs']) / 1e3:8.1f} us  ms/it {float(r['TotalDurationNs']) / 1e6 / iters:7.3f}")
