"""Summarise rocprofv3 --pmc passes (counter_collection.csv) per kernel: average counter value per launch.

    python tools/pmc_summary.py [--last N] OUT.json DIR_OR_CSV [DIR_OR_CSV ...]

--last N: only the last N launches of every kernel count (a run that fills the L-BFGS history first: the sweeps at full history).

Each pass is a separate `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -- python3 bench.py ...` run
(MI355X_MICROARCH.md, section HBM: FETCH_SIZE / WRITE_SIZE come from the L2's memory-side request counters; on gfx950
FETCH_SIZE reports half the bytes of wide coalesced reads).  Kernel names are shortened to the function name plus
template arguments.  The output feeds bench.py's `roofline.traffic`.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

csv.field_size_limit(1 << 30)


def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"((?:maua::)?[A-Za-z0-9_]+(?:<[^>]*>)?)", name)
    return m.group(1) if m and name.startswith("maua::") else name[:60]


def main():
    argv = sys.argv[1:]
    last = 0
    if argv and argv[0] == "--last":
        last, argv = int(argv[1]), argv[2:]
    out, srcs = argv[0], argv[1:]
    files = []
    for s in srcs:
        files += [s] if s.endswith(".csv") else glob.glob(os.path.join(s, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))   # kernel -> counter -> [sum, launches]
    dur = defaultdict(lambda: [0.0, 0])
    for f in files:
        seen = set()
        with open(f, newline="") as fh:
            rows = list(csv.DictReader(fh))
        if last:  # keep the rows of each kernel's last `last` dispatches
            ids = defaultdict(set)
            for row in rows:
                ids[short(row["Kernel_Name"])].add(int(row["Dispatch_Id"]))
            keep = {k: set(sorted(v)[-last:]) for k, v in ids.items()}
            rows = [row for row in rows if int(row["Dispatch_Id"]) in keep[short(row["Kernel_Name"])]]
        if True:
            for row in rows:
                k = short(row["Kernel_Name"])
                a = acc[k][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
                key = (row["Dispatch_Id"],)
                if key not in seen:
                    seen.add(key)
                    d = dur[k]
                    d[0] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
                    d[1] += 1
    res = {}
    for k, counters in acc.items():
        e = {"launches_seen": max(v[1] for v in counters.values()),
             "avg_us_under_pmc": round(dur[k][0] / max(1, dur[k][1]), 2)}
        for c, (s, n) in counters.items():
            e[c] = round(s / n, 3)
        # bytes per launch from the raw request counters (64 B per request, 32 B for the _32B subset)
        if "TCC_EA0_RDREQ_sum" in e:
            r32 = e.get("TCC_EA0_RDREQ_32B_sum", 0.0)
            e["read_bytes_raw"] = round((e["TCC_EA0_RDREQ_sum"] - r32) * 64 + r32 * 32)
        if "TCC_EA0_WRREQ_sum" in e:
            w64 = e.get("TCC_EA0_WRREQ_64B_sum", 0.0)
            e["write_bytes_raw"] = round(w64 * 64 + (e["TCC_EA0_WRREQ_sum"] - w64) * 32)
        if "FETCH_SIZE" in e:
            e["fetch_bytes"] = round(e["FETCH_SIZE"] * 1024)       # the derived metric is in KiB
        if "WRITE_SIZE" in e:
            e["write_bytes"] = round(e["WRITE_SIZE"] * 1024)
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e and e["TCC_HIT_sum"] + e["TCC_MISS_sum"] > 0:
            e["l2_hit_rate"] = round(e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"]), 4)
        res[k] = e
    with open(out, "w") as fh:
        json.dump({"source_files": [os.path.basename(f) for f in files], "kernels": res}, fh, indent=1, sort_keys=True)
    for k in sorted(res, key=lambda k: -res[k].get("fetch_bytes", res[k].get("read_bytes_raw", 0))):
        print(k, {a: b for a, b in res[k].items() if "bytes" in a or a in ("launches_seen", "l2_hit_rate")})


if __name__ == "__main__":
    main()
