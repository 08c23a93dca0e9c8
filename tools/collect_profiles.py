"""gpurun_out/r2fin (written by tools/profile_round.sh on the GPU box) -> the tracked files under profiles/ for round RR.
    python tools/collect_profiles.py [RR=02] [SRC=gpurun_out/r2fin]"""
import csv
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rr = sys.argv[1] if len(sys.argv) > 1 else "06"
src = os.path.join(REPO, sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r6fin")
dst = os.path.join(REPO, "profiles")


def last_json(path):
    with open(path) as f:
        return json.loads([l for l in f.read().splitlines() if l.strip().startswith("{")][-1])


def conv_avg(stats_csv):
    rows = [r for r in csv.DictReader(open(stats_csv)) if "conv_x3w_kernel" in r["Name"] or "conv_x3q_kernel" in r["Name"] or "conv_x3p_kernel" in r["Name"]]
    t, c = sum(float(r["TotalDurationNs"]) for r in rows), sum(int(r["Calls"]) for r in rows)
    return t / c / 1e3, c


def kernel_avg(stats_csv, name):
    for r in csv.DictReader(open(stats_csv)):
        if name in r["Name"]:
            return float(r["AverageNs"]) / 1e3
    return float("nan")


for a, b in (("bench_graph.json", f"bench_r{rr}_final_1024_lbfgs.json"), ("bench_eager.json", f"bench_r{rr}_final_1024_lbfgs_eager.json"),
             ("bench_under_rocprof.json", f"bench_r{rr}_under_rocprof.json")):
    with open(os.path.join(dst, b), "w") as f:
        f.write(json.dumps(last_json(os.path.join(src, a))) + "\n")
for a, b in (("sizes_lbfgs.jsonl", f"bench_r{rr}_sizes_lbfgs.jsonl"), ("sizes_adam.jsonl", f"bench_r{rr}_sizes_adam.jsonl"),
             ("configs.json", f"configs_r{rr}_final.json"), ("pmc_traffic.json", f"pmc_r{rr}_traffic.json"),
             ("pmc_traffic_nin.json", f"pmc_r{rr}_traffic_nin.json"), ("launches_graph_1024.txt", f"probe_r{rr}_launch_table_1024_graph.txt"),
             ("launches_graph_512.txt", f"probe_r{rr}_launch_table_512_graph.txt"), ("launches_graph_256.txt", f"probe_r{rr}_launch_table_256_graph.txt"),
             ("launches_graph_nin.txt", f"probe_r{rr}_launch_table_nin_graph.txt"), ("graph_host_cost.txt", f"probe_r{rr}_graph_host_cost.txt"),
             ("check_x3p.txt", f"probe_r{rr}_x3p_vs_x3q_x3w.txt"),
             ("clock_x3p_conv1_2.txt", f"probe_r{rr}_clock_x3p_conv1_2.txt"), ("soak.txt", f"probe_r{rr}_soak.txt"),
             ("stage_bw.txt", f"probe_r{rr}_stage_bw.txt"), ("gram128_zero_lanes.txt", f"probe_r{rr}_gram128_zero_lanes_matrix.txt")):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, b))
if os.path.exists(os.path.join(src, "bench_nin.json")):
    with open(os.path.join(dst, f"bench_r{rr}_nin_config5.json"), "w") as f:
        f.write(json.dumps(last_json(os.path.join(src, "bench_nin.json"))) + "\n")
serial = os.path.join(src, "stats", "p_kernel_stats.csv")
shutil.copy(serial, os.path.join(dst, f"rocprof_r{rr}_kernel_stats_1024_lbfgs.csv"))
table = subprocess.run([sys.executable, os.path.join(REPO, "tools", "summarise_stats.py"), serial, "122", rr], capture_output=True, text=True, check=True).stdout
s_avg, s_calls = conv_avg(serial)
under = last_json(os.path.join(src, "bench_under_rocprof.json"))
graph = last_json(os.path.join(src, "bench_graph.json"))
note = f"""
Everything of an iteration is in stream order at this size (the Gram / loss chains moved off the side stream in round 3: their partial
kernels go out in two launches behind the forward pass; images of 1536² and more still overlap them with the convolutions).  `bench.py`
brackets every convolution launch with HIP events in a pass of its own, and this trace is what its `roofline.avg_launch_ms` has to agree with:
**{s_avg:.1f} µs here ({s_calls} launches of the `conv_x3w_kernel`, `conv_x3q_kernel` and `conv_x3p_kernel` variants) against {under['roofline']['avg_launch_ms'] * 1e3:.1f} µs in `bench_r{rr}_under_rocprof.json`** (the JSON
line of this very run: the events also see the launch gaps of an eager run under the profiler).  Unprofiled, the same figure is
{graph['roofline']['avg_launch_ms'] * 1e3:.1f} µs (`bench_r{rr}_final_1024_lbfgs.json`, {graph['value']:.1f} it/s with graph replay).

Which layer runs which kernel: `extra.routes` of the bench lines (round 5: the persistent `conv_x3p_kernel<OM, POOL, UNPOOL, GRAM>` takes the launches
of two work items and more per workgroup - at 1024 x 1024 conv1_2 ... conv3_4 forward, conv3_4 / conv3_3 / conv2_1 backward).
What the variants of the other two are (template arguments ACC, OM, POOL, UNPOOL; `conv_x3q_kernel` = the 256+ channel passes since round 4, `conv_x3w_kernel` the others): `<false,false,false,false>` plain forward / backward-data launches;
`<false,true,false,false>` backward-data with the ReLU mask of the produced gradient in the epilogue, one of them carrying the Gram backward of
relu3_1 (`maua_conv3x3_x3w_gram`); `<false,true,false,true>` the backward passes of conv1_2 / 2_2 / 3_4 / 4_4 staged from the POOLED map's gradient
and the pool's decision bytes (`maua_conv3x3_x3w_unpool`; conv1_2 / conv2_2 with the Gram backward of relu1_1 / relu2_1 along);
`<false,false,true,false>` forward with ReLU + 2x2 max pool in the epilogue (conv1_2, 2_2, 3_4, 4_4).
L-BFGS kernels: the averages include the 100 history-filling iterations (sweeps grow linearly with the history: ~450 µs each at
full history, see `probes_r{rr}.md`).
"""
i = table.index("\n| kernel |")
with open(os.path.join(dst, f"rocprof_r{rr}_summary.md"), "w") as f:
    f.write(table[:i] + note + table[i:])
print("serial conv avg", round(s_avg, 1), "bench under rocprof", under["roofline"]["avg_launch_ms"], "graph", graph["value"], graph["roofline"]["frac"])
for l in open(os.path.join(src, "sizes_lbfgs.jsonl")):
    d = json.loads(l)
    print(d["config"]["image_size"], d["value"], d["roofline"]["frac"])
for l in open(os.path.join(src, "sizes_adam.jsonl")):
    d = json.loads(l)
    print("adam", d["config"]["image_size"], d["value"])
for r in json.load(open(os.path.join(src, "configs.json")))["results"]:
    print(r["config"], r.get("wall_seconds"), r.get("frames_per_second"), r.get("iterations_per_second_incl_setup"))
