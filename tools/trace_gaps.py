"""rocprofv3 --kernel-trace CSV -> where the time of a steady-state window goes: per-kernel busy time, idle time between
kernels (launch boundaries), overlap.      python tools/trace_gaps.py KERNEL_TRACE.csv [WINDOW_FRACTION=0.3] [TOP=25]
The window is the last WINDOW_FRACTION of the trace (the timed region of bench.py comes last)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
t_end = max(e for _, e, _ in ev)
t0 = ev[0][0]
lo = t_end - int((t_end - t0) * frac)
ev = [e for e in ev if e[0] >= lo]
span = ev[-1][1] - ev[0][0]
busy = 0
cur_s, cur_e = ev[0][0], ev[0][1]
gaps = []
for s, e, _ in ev[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append(s - cur_e)
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
per = defaultdict(lambda: [0, 0])
for s, e, n in ev:
    k = n.split("(")[0].replace("void ", "").replace("maua::", "")[:56]
    per[k][0] += 1
    per[k][1] += e - s
ksum = sum(v[1] for v in per.values())
print(f"window {span / 1e6:.2f} ms, {len(ev)} launches: GPU busy {busy / 1e6:.2f} ms ({100 * busy / span:.1f} %), idle {(span - busy) / 1e6:.2f} ms "
      f"in {len(gaps)} gaps (median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, mean {sum(gaps) / max(len(gaps), 1) / 1e3:.2f} us); "
      f"sum of kernel durations {ksum / 1e6:.2f} ms (overlap {100 * (ksum - busy) / max(busy, 1):.1f} %)")
for k, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{k:56s} {c:6d} launches  avg {t / c / 1e3:8.2f} us  {100 * t / span:5.1f} % of the window")
