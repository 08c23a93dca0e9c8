"""Sweep planner fields at the smaller image sizes: one bench.py run per (size, MAUA_PLAN) in fresh processes on ONE box, the default first and
last (drift check).     python tools/sweep_plan.py SIZE[,SIZE...] "field=v,field=v" "field=v" ...     (prints it/s per configuration)"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(v) for v in sys.argv[1].split(",")]
plans = [""] + sys.argv[2:] + [""]
steps = int(os.environ.get("STEPS", "200"))
for S in sizes:
    for pl in plans:
        env = dict(os.environ)
        if pl:
            env["MAUA_PLAN"] = pl
        else:
            env.pop("MAUA_PLAN", None)
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--size", str(S), "--steps", str(steps), "--no_cpu_baseline",
                              "--no_extra_sizes", "--no_exact_split", "--no_accuracy_probe", "--no_repeats"], env=env, capture_output=True, text=True)
        try:
            d = json.loads(out.stdout.strip().splitlines()[-1])
            print(f"{S:5d}  {d['value']:9.2f} it/s  {d['ms_per_step']:.4f} ms   {pl or '(default)'}", flush=True)
        except Exception:
            print(f"{S:5d}  FAILED  {pl}  {out.stderr.strip()[-300:]}", flush=True)
