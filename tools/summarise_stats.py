"""rocprofv3 kernel_stats.csv -> markdown table (profiles/rocprof_rNN_summary.md).
    python tools/summarise_stats.py STATS.csv ITERATIONS [ROUND] > profiles/rocprof_rNN_summary.md"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
iters = int(sys.argv[2])
calls = [int(r["Calls"]) for r in rows if "lbfgs_pair_kernel" in r["Name"]]
if calls:  # one launch per iteration: the trace itself says how many iterations it holds (bench.py's repeated regions included)
    iters = calls[0]
rnd = sys.argv[3] if len(sys.argv) > 3 else "01"
print(f"# rocprofv3 --kernel-trace --stats, round {int(rnd)} (final state of the round)\n")
print("Command (on the MI355X box, from /tmp with TMPDIR=/tmp):")
print("`rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/... -o p -- python3 bench.py --steps 20 --warmup 2 "
      "--no_cpu_baseline --no_extra_sizes --no_hip_graph`   (tools/profile_round.sh)")
print(f"({iters} iterations in total: 100 history-filling + 2 warm-up + 20 timed" + (" + 5 x 20 of `extra.repeats`" if iters == 222 else "") +
      "; 1024x1024, L-BFGS, eager launches so that every kernel "
      f"appears under its own name; the JSON line of this run is `bench_r{rnd}_under_rocprof.json`, the unprofiled runs "
      f"`bench_r{rnd}_final_1024_lbfgs.json` (hipGraph replay, the product default) and `bench_r{rnd}_final_1024_lbfgs_eager.json`).  "
      "The averages of the dominant convolution kernel here are the figures `bench.py`'s `roofline.avg_launch_ms` must agree with.\n")
print("| kernel | calls | total ms | avg us | % | ms / iteration |")
print("|---|---|---|---|---|---|")
for r in rows[:28]:
    name = r["Name"].replace("|", "\\|")
    if len(name) > 96:
        name = name[:96] + "..."
    print(f"| `{name}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.2f} | {float(r['AverageNs']) / 1e3:.1f} | "
          f"{float(r['Percentage']):.2f} | {float(r['TotalDurationNs']) / 1e6 / iters:.3f} |")
print("\nPMC passes (separate `rocprofv3 --pmc` runs of `bench.py --steps 4 --warmup 1 --no_prefill --no_cpu_baseline --no_hip_graph`): "
      f"`pmc_r{rnd}_traffic.json` (memory-side request counters, L2 hit rate, SQ wait / busy counters per kernel, "
      "`tools/pmc_summary.py`); calibration of the request counters: `pmc_r01_calibration.json`.")
