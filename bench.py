#!/usr/bin/env python3
"""Benchmark of the image-optimisation hot path on MI355X.

A "step" is one optimizer iteration of the reference's loop (reference optim.py:201-241): one function
evaluation (VGG-19 forward, Gram/content/TV losses, backward to the pixels) plus one L-BFGS (or Adam) pixel
update, on ONE synthetic 1024x1024 image per GPU (BASELINE.json metric: "optimizer iterations/sec at 1024x1024
VGG-19").  With N GPUs every rank optimises its own image (independent frames, the natural shard axis), after
one RCCL broadcast of the conv weights and style targets; there is no per-iteration collective (weak scaling).

Before the timed region the L-BFGS history (100 pairs) is filled by running `history` real iterations, so the
timed steps are steady-state iterations (the two-loop cost grows until the history is full).

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the split-precision 3x3 convolution: conv_x3w.hip,
with conv_x3.hip on maps below 64x64): algorithmic fp32-equivalent FLOPs per launch / average launch duration measured with
HIP events on the launch stream, against the dense 16-bit MFMA peak divided by the MFMAs one product block costs (2500 / 3
for fp16x3, 2500 / 6 for bf16x6; 157.3 TFLOP/s when the fp32 matrix cores are selected with MAUA_CONV_X6=0); `traffic` comes
from the committed PMC passes.  `cpu_baseline` times the CPU oracle on this host's cores at the benchmarked size; `extra`
carries the 512x512 figure.
"""
import argparse
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]

import torch  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (same guide)
X6_MFMAS_PER_PRODUCT = 6        # bf16x6: six bf16 MFMAs carry one fp32-accurate product block (conv_x6.hip)
HBM_PEAK_GBS = 8000.0


def algorithmic_work(S, history=100):
    """FLOPs and bytes of one iteration at S x S (SURVEY.md Appendix B calculator, restated)."""
    chans = [64, 64, "P", 128, 128, "P", 256, 256, 256, 256, "P", 512, 512, 512, 512, "P", 512]
    names = ["conv1_1", "conv1_2", "pool1", "conv2_1", "conv2_2", "pool2", "conv3_1", "conv3_2", "conv3_3", "conv3_4",
             "pool3", "conv4_1", "conv4_2", "conv4_3", "conv4_4", "pool4", "conv5_1"]
    style, content = {"conv1_1", "conv2_1", "conv3_1", "conv4_1", "conv5_1"}, {"conv4_2"}
    H = W = S
    cin, macs, bf, bb, gram = 3, 0, 0, 0, 0
    for n, c in zip(names, chans):
        if c == "P":
            bi, bo = cin * H * W * 4, cin * (H // 2) * (W // 2) * 4
            bf += bi + bo
            bb += bo + 2 * bi
            H //= 2
            W //= 2
            continue
        mac = 9 * cin * c * H * W
        wb, ib, ob = (9 * cin * c + c) * 4, cin * H * W * 4, c * H * W * 4
        macs += mac
        bf += ib + wb + ob
        bb += 2 * ob + wb + ib
        if n in style:
            gram += 2 * (2 * c * c * H * W)
            bf += ob + c * c * 4
            bb += ob + c * c * 4 + 2 * ob
        if n in content:
            bf += 2 * ob
            bb += 4 * ob
        cin = c
    npx = 3 * S * S
    return dict(flops=4 * macs + gram, bytes_feval=bf + bb, bytes_lbfgs=(4 * history + 10) * npx * 4)


def _host_description():
    """(nproc, CPU model string) of this host: os.cpu_count() and the 'model name' line of /proc/cpuinfo (what lscpu prints)."""
    model = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return os.cpu_count() or 1, model


def cpu_baseline(S, optimizer, iters=5, repeats=3, model="vgg19"):
    """The CPU oracle (oracle/, a restatement of the reference's arithmetic on torch CPU ops) timed on this host's cores AT
    THE BENCHMARKED SIZE, no extrapolation: `repeats` runs of `iters` L-BFGS iterations (evaluate + two-loop update) from
    the same start, median.  The intra-op thread count is the fastest of a sweep of single evaluations up to nproc
    (MKL-DNN oversubscribes badly on many-core hosts, SURVEY.md section 8c: more threads is not faster)."""
    import argparse as ap
    import statistics

    import synth
    from oracle import OracleNet, build_spec
    from oracle.style_oracle import _LbfgsState, _lbfgs_step
    nproc, cpu_model = _host_description()
    cfg = ap.Namespace(model_file="vgg19", pooling="max", content_layers="relu4_2",
                       style_layers="relu1_1,relu2_1,relu3_1,relu4_1,relu5_1", tv_weight=1e-3, temporal_weight=50.0,
                       content_weight=5.0, style_weight=100.0, use_covariance=False, normalize_gradients=True,
                       video_style_factor=100.0)
    if model == "nin":  # BASELINE config 5
        cfg.model_file, cfg.content_layers, cfg.style_layers, cfg.use_covariance = "nin", "relu8", "relu1,relu3,relu5,relu7,relu9,relu11", True
    net = OracleNet(build_spec(cfg), synth.nin_state_dict() if model == "nin" else synth.vgg19_state_dict())
    content, style, init = synth.images(S)
    net.capture_content(content)
    net.capture_style([style], [1.0])
    shape = init.shape
    sweep = {}
    best = (float("inf"), 1)
    counts, c = [], 1
    while c < nproc:  # {1, 2, 4, ..., nproc}
        counts.append(c)
        c *= 2
    counts.append(nproc)

    def one(k):
        torch.set_num_threads(k)
        t0 = time.perf_counter()
        net.feval(init)
        sweep[k] = round(time.perf_counter() - t0, 3)
        return sweep[k]
    # Order: from 4 threads upwards first (the optimum on many-core hosts is 8 - 32; the upward walk stops once a count is twice as slow as
    # the best so far: larger counts only oversubscribe further), then 2 and 1 - a one-thread evaluation at 1024 x 1024 is ~40 s - while the
    # sweep's 150 s budget lasts.  Counts that were not run are listed as such.
    budget_s = 150.0
    t_sweep = time.perf_counter()
    for k in [v for v in counts if v >= 4]:
        dt = one(k)
        best = min(best, (dt, k))
        if k > best[1] and dt > 2.0 * best[0]:
            break
    for k in [v for v in (2, 1) if v in counts]:
        if time.perf_counter() - t_sweep + 2.2 * sweep.get(2 * k, 0.0) > budget_s:
            break  # (an evaluation on k threads takes about twice the one on 2 k)
        dt = one(k)
        best = min(best, (dt, k))
    sweep = dict(sorted(sweep.items()))
    not_run = [k for k in counts if k not in sweep]
    torch.set_num_threads(best[1])

    def closure(x):
        total, _, g = net.feval(x.reshape(shape))
        return float(total), g.flatten()

    times = []
    for _ in range(repeats):
        st = _LbfgsState()
        t0 = time.perf_counter()
        _lbfgs_step(init.flatten().clone(), closure, st, iters, 100)
        times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    return {"value": round(iters / dt, 5), "unit": "iterations/s", "cores": best[1], "threads_used": best[1], "nproc": nproc,
            "cpu_model": cpu_model, "model": model, "kind": "port", "thread_sweep_s_per_feval": sweep,
            "thread_counts_not_run": not_run,  # (beyond twice the best time, or the sweep's 150 s budget)
            "sample": f"{iters} L-BFGS iterations of the CPU oracle at {S}x{S} (the benchmarked size, no scaling), median of "
                      f"{repeats} runs: {dt:.2f} s on {best[1]} threads ({nproc} logical CPUs)"}


def accuracy_probe():
    """Measured in THIS run (nothing quoted from elsewhere): the timed convolution kernels on a conv4_2-shaped layer (512 -> 512 channels,
    128 x 128, post-ReLU-like input, seeded) against `F.conv2d` in fp64 on a 32 x 32 output crop, next to the same crop computed by
    the reference's own arithmetic (fp32 `F.conv2d` on the CPU).  rel-L2 errors; the fp16x3 claim is "as close to fp64 as fp32"."""
    import torch.nn.functional as F
    import hip
    g = torch.Generator().manual_seed(11)
    cin = cout = 512
    x = torch.relu(torch.randn(1, cin, 128, 128, generator=g))
    w = torch.randn(cout, cin, 3, 3, generator=g) * (2.0 / (9 * cin)) ** 0.5
    ref = F.conv2d(x[:, :, :34, :34].double(), w.double(), padding=1)[:, :, :32, :32]
    rel = lambda y: float((y.double() - ref).norm() / ref.norm())
    out = {"layer": "3x3, 512 -> 512 channels, 128 x 128 (conv4_2's shape), seeded post-ReLU-like input, 32 x 32 output crop vs fp64",
           "fp32_cpu_conv": rel(F.conv2d(x[:, :, :34, :34], w, padding=1)[:, :, :32, :32])}
    xd, wd = x.cuda(), w.cuda()
    for name, pack, run in (("conv_x3w", hip.conv_pack_filters_x3w, hip.conv3x3_x3w), ("conv_x3q", hip.conv_pack_filters_x3q, hip.conv3x3_x3q),
                            ("conv_x3p", hip.conv_pack_filters_x3q, hip.conv3x3_x3p)):
        bank, _, wsc = pack(wd)
        out[name] = rel(run(xd, bank, wsc, None, cout, 1, False)[:, :, :32, :32].cpu())
    return {k: (round(v, 10) if isinstance(v, float) else v) for k, v in out.items()}


def pmc_traffic(prefix):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC pass (profiles/pmc_r0N_traffic.json, the newest; written
    by tools/pmc_summary.py from separate `rocprofv3 --pmc` runs of this same command).  Reads: request counters x 64 B,
    doubled as MI355X_MICROARCH.md prescribes for gfx950 (our own calibration, profiles/pmc_r01_calibration.json: x2.0
    for 16 B/lane streams, x1.2-1.6 for 4 B/lane patterns, so this is an upper bound); writes are exact."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = next((os.path.join(here, f) for f in ("pmc_r06_traffic.json", "pmc_r05_traffic.json", "pmc_r04_traffic.json", "pmc_r03_traffic.json", "pmc_r02_traffic.json")
                 if os.path.exists(os.path.join(here, f))), None)
    if path is None:
        return None
    with open(path) as f:
        kernels = json.load(f)["kernels"]
    n = rd = wr = 0
    for name, e in kernels.items():
        if name.startswith(tuple(prefix) if isinstance(prefix, (tuple, list)) else prefix) and "read_bytes_raw" in e:
            n += e["launches_seen"]
            rd += e["read_bytes_raw"] * e["launches_seen"]
            wr += e["write_bytes_raw"] * e["launches_seen"]
    if not n:
        return None
    return {"bytes": round((2 * rd + wr) / n),
            "note": f"NOT measured by this run: per launch, read from the committed profiles/{os.path.basename(path)} (separate rocprofv3 --pmc passes of this command): "
                    "2 x TCC_EA0_RDREQ x 64 B (gfx950 correction, upper bound for 4 B/lane loads) + write requests"}


def pmc_traffic_lbfgs():
    """Memory-side bytes of one maua_lbfgs_iterate at full history (its five launches added up) from the committed PMC pass of
    `bench.py --model nin` (profiles/pmc_r0N_traffic_nin.json, the newest: tools/profile_round.sh, the last four launches of each kernel)."""
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles")
    path = next((os.path.join(here, f) for f in ("pmc_r06_traffic_nin.json", "pmc_r05_traffic_nin.json", "pmc_r04_traffic_nin.json", "pmc_r03_traffic_nin.json")
                 if os.path.exists(os.path.join(here, f))), None)
    if path is None:
        return None
    with open(path) as f:
        kernels = json.load(f)["kernels"]
    total = sum(2 * e["read_bytes_raw"] + e["write_bytes_raw"] for name, e in kernels.items() if name.startswith(("maua::lbfgs_pair", "maua::lbfgs_finish_dots", "maua::lbfgs_coeffs", "maua::lbfgs_combine")) and "read_bytes_raw" in e)
    if not total:
        return None
    return {"bytes": round(total), "note": f"per update (the five launches), from profiles/{os.path.basename(path)} (a separate rocprofv3 --pmc pass of "
                                           "this command, last four launches of each kernel = full history): 2 x TCC_EA0_RDREQ x 64 B + write requests"}


def _visible_gpus():
    """Device count WITHOUT initialising the GPU in this process (torch.cuda.device_count() does not, on this image)."""
    try:
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def launch_ranks(n, argv, allow_fewer=False):
    """`python bench.py --gpus N` outside torchrun: start N fresh rank processes (one per GPU) with torchrun's environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT) and wait for them.  This parent never touches the GPU and
    never re-execs itself; rank 0's JSON line is relayed on stdout.  Fewer than N visible devices is an error unless
    `--allow_fewer` (run min(N, visible) ranks and say so in the line: `n_gpus` is what ran, `requested_gpus` what was asked)
    or MAUA_DIST_BACKEND=gloo (several ranks per GPU, the 1-GPU test box).  All children are polled: when one exits non-zero
    (a GPU fault, a failed rendezvous) the others are terminated and the launcher returns non-zero instead of waiting in a
    collective."""
    import socket
    import subprocess
    have = _visible_gpus()
    if have < n and os.environ.get("MAUA_DIST_BACKEND") != "gloo":
        if allow_fewer and have >= 1:
            sys.stderr.write(f"bench.py: --gpus {n} requested, {have} device(s) visible: running {have} rank(s) (--allow_fewer)\n")
            argv = _with_flag_value(argv, "--gpus", str(have)) + ["--requested_gpus", str(n)]
            n = have
        else:
            sys.stderr.write(f"bench.py: --gpus {n} but only {have} device(s) visible (--allow_fewer runs on what is there; "
                             "MAUA_DIST_BACKEND=gloo shares GPUs between ranks)\n")
            return 2
    if n == 1:  # one rank: no process group needed, but still a fresh child (this parent must stay off the GPU)
        return subprocess.call([sys.executable, os.path.abspath(__file__)] + argv + ["--_launched"])
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile(mode="w+")
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
                       "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, text=True))
    rcs = [None] * n
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
        if any(rc not in (None, 0) for rc in rcs):
            for r, p in enumerate(procs):  # a rank failed: the others would sit in rendezvous or a collective holding their GPUs
                if rcs[r] is None:
                    p.terminate()
            deadline = time.time() + 20
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=max(0.1, deadline - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[r] = p.wait()
            break
        time.sleep(0.2)
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write(f"bench.py: ranks failed or were stopped (rank, exit code): {bad}\n")
        return 1
    return 0


def _with_flag_value(argv, flag, value):
    """argv with `flag value` / `flag=value` replaced (appended when absent)."""
    out, i, seen = [], 0, False
    while i < len(argv):
        if argv[i] == flag and i + 1 < len(argv):
            out += [flag, value]
            i += 2
            seen = True
        elif argv[i].startswith(flag + "="):
            out.append(f"{flag}={value}")
            i += 1
            seen = True
        else:
            out.append(argv[i])
            i += 1
    return out if seen else out + [flag, value]


def start_exact_split_child(argv_size, steps, warmup, history, optimizer):
    """The same workload on the EXACT three-way bf16 split (MAUA_CONV_X3=0: bf16x6, 24-bit operands) in a fresh child process.
    It is started here - before this process makes its first GPU call - and then waits on its stdin until the parent has
    finished its own measurement (two jobs timing each other would measure neither)."""
    import subprocess
    env = dict(os.environ, MAUA_CONV_X3="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.Popen([sys.executable, os.path.abspath(__file__), "--gpus", "1", "--size", str(argv_size), "--steps", str(steps),
                             "--warmup", str(warmup), "--history", str(history), "--optimizer", optimizer, "--no_cpu_baseline",
                             "--no_extra_sizes", "--no_exact_split", "--no_repeats", "--no_accuracy_probe", "--_wait_for_go"], env=env, stdin=subprocess.PIPE,
                            stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)


def finish_exact_split_child(proc):
    try:
        out, _ = proc.communicate("go\n", timeout=900)
    except Exception as e:  # noqa: BLE001 - the headline must not die with its side figure
        proc.kill()
        return {"error": repr(e)}
    lines = [l for l in out.splitlines() if l.startswith("{")]
    if proc.returncode != 0 or not lines:
        return {"error": f"child exit code {proc.returncode}"}
    d = json.loads(lines[-1])
    r = d.get("roofline") or {}
    return {"env": "MAUA_CONV_X3=0", "dtype": d["dtype"], "iterations_per_s": d["value"], "ms_per_step": d["ms_per_step"],
            "steps": d["steps"], "roofline_kernel": r.get("kernel"), "roofline_achieved_tflops": r.get("achieved"),
            "roofline_peak_tflops": r.get("peak"), "roofline_frac": r.get("frac"),
            "note": "same workload, same process layout, every 3x3 convolution on exact bf16x6 products (three bf16 parts per operand = 24 "
                    "significand bits, six MFMAs per product block); measured in a fresh child process after the headline"}


def lbfgs_still_moving(opt, expect_iters, what):
    """{n_iter, stopped, history_len} of the optimiser right behind a timed region - and the run FAILS if a stop rule fired inside it:
    a raised stop flag (g . d > -tolerance, max |g|, ...) makes lbfgs_pair_dots and the kernels behind it return early (lbfgs.hip), which
    would inflate iterations/s.  The line must prove by itself that every timed iteration did its whole work."""
    st = opt.state.status()
    out = {"n_iter": int(st["n_iter"]), "stopped": bool(st["stopped"]), "history_len": int(st["history_len"]), "expected_n_iter": int(expect_iters)}
    if out["stopped"] or out["n_iter"] != out["expected_n_iter"]:
        raise RuntimeError(f"bench.py: the L-BFGS optimiser stopped moving inside the timed region of {what}: {out} - the measured rate is not valid")
    return out


def steady_rate(size, steps, optimizer="lbfgs", history=100, warmup=5):
    """iterations/s of one more single-image job at another size on this rank's GPU (the 512x512 figure north_star asks for
    next to the 1024x1024 headline): same construction as the main workload, history prefilled, graph replay."""
    import config
    import models
    import optim
    import synth
    tmp = tempfile.mkdtemp(prefix="maua_bench_")
    wfile = os.path.join(tmp, "vgg19_synth.pth")
    torch.save(synth.vgg19_state_dict(), wfile)
    scaling = os.path.join(tmp, "scaling.json")
    with open(scaling, "w") as f:
        json.dump({"100000": {"gpu": "0", "multidevice": False}}, f)
    args = config.get_args(["--content", "c.png", "--style", "s.png", "--model_file", wfile, "--disable_check",
                            "--scaling_args", scaling, "--optimizer", optimizer, "--image_sizes", str(size),
                            "--num_iters", str(steps), "--seed", "0", "--no_hist_match", "--lbfgs_num_correction", str(history)])
    args.hip_graph = True
    optim.set_model_args(args, size)
    net, losses = models.load_model(args)
    content, style, init = synth.images(size)
    optim.set_content_targets(net, content, args)
    optim.set_style_targets(net, [style], args)
    for m in losses:
        m.mode = "loss"
    opt = optim.PixelOptimizer(net, losses, init, args)
    for _ in range((history if optimizer == "lbfgs" else 0) + warmup):
        opt.step()
    import dist as _dist
    _dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        opt.step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _dist.barrier()
    moving = lbfgs_still_moving(opt, history + warmup + steps, f"the {size}x{size} figure") if optimizer == "lbfgs" else None
    work = algorithmic_work(size, history)
    import dist
    world = dist.group_size()
    dt_all = dist.max_over_ranks(dt)  # whole job: every rank ran `steps` iterations on its own image; the slowest rank's time
    res = {"image_size": size, "optimizer": optimizer, "steps": steps, "n_gpus": world,
           "iterations_per_s": round(world * steps / dt_all, 2), "ms_per_step": round(dt_all / steps * 1e3, 4),
           "model_tflops": round(world * work["flops"] / (dt_all / steps) / 1e12, 2)}
    if moving is not None:
        res["lbfgs"] = moving
    conv = opt.engine is not None and _conv_roofline_of(opt, steps)
    if conv:
        res["conv_roofline_frac"] = conv
    return res


def _conv_roofline_of(opt, steps):
    """roofline.frac of the split-precision 3x3 launches at this size: a short eager pass with HIP events around every launch."""
    import models
    timer = []
    opt.engine.timer = timer
    try:
        for _ in range(min(steps, 20)):
            opt.step()
        torch.cuda.synchronize()
    finally:
        opt.engine.timer = None
    conv = [(fl, e0.elapsed_time(e1)) for tag, fl, nb, e0, e1 in timer if tag.startswith("conv3x3_split")]
    if not conv:
        return None
    per_product = 3 if models._x3_enabled() else X6_MFMAS_PER_PRODUCT
    return round(sum(c[0] for c in conv) / (sum(c[1] for c in conv) * 1e-3) / 1e12 / (BF16_MFMA_PEAK_TFLOPS / per_product), 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--optimizer", default="lbfgs", choices=["lbfgs", "adam"])
    ap.add_argument("--history", type=int, default=100)
    ap.add_argument("--model", default="vgg19", choices=["vgg19", "nin"],
                    help="nin: BASELINE config 5 (NIN + --use_covariance, reference models.py:74-113, loss.py:87-89) instead of the headline")
    ap.add_argument("--no_prefill", action="store_true", help="do not fill the L-BFGS history before timing")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--hip_graph", action="store_true", help="(default) replay each iteration from a captured hipGraph")
    ap.add_argument("--no_hip_graph", action="store_true", help="launch every kernel eagerly in the timed region too")
    ap.add_argument("--no_extra_sizes", action="store_true", help="skip the 512x512 / 256x256 figures in `extra`")
    ap.add_argument("--extra_sizes", default=None, help="comma-separated image sizes for `extra.other_sizes` (default: 512,256 next to a 1024 run)")
    ap.add_argument("--no_exact_split", action="store_true", help="skip the bf16x6 (MAUA_CONV_X3=0) figure in `extra`")
    ap.add_argument("--no_repeats", action="store_true", help="skip the repeated timed regions in `extra.repeats`")
    ap.add_argument("--no_accuracy_probe", action="store_true", help="skip the in-run fp64 check of the timed convolution kernels")
    ap.add_argument("--repeats", type=int, default=5, help="further K-step regions timed after the headline one")
    ap.add_argument("--allow_fewer", action="store_true", help="with fewer than --gpus devices visible: run on those and report it")
    ap.add_argument("--requested_gpus", type=int, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--_launched", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--_wait_for_go", action="store_true", help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a.gpus, [v for v in sys.argv[1:] if v != "--allow_fewer"], a.allow_fewer))  # before anything here touches the GPU
    if a._wait_for_go and sys.stdin.readline().strip() != "go":  # exact-split child: the parent says when the GPU is free
        sys.exit(3)  # (the parent went away without asking)
    exact_child = None
    if a.gpus == 1 and "WORLD_SIZE" not in os.environ and a.size == 1024 and not a.no_exact_split and a.model == "vgg19":
        exact_child = start_exact_split_child(a.size, a.steps, a.warmup, a.history, a.optimizer)  # before the first GPU call

    import config
    import dist
    import hip
    import models
    import optim
    import synth

    rank, local_rank, world = dist.init()
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the process group has WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU implementation")
    hip.lib()
    dev = torch.device("cuda", torch.cuda.current_device())
    S = a.size

    tmp = tempfile.mkdtemp(prefix="maua_bench_")
    nin = a.model == "nin"
    wfile = os.path.join(tmp, "nin_synth.pth" if nin else "vgg19_synth.pth")
    torch.save(synth.nin_state_dict() if nin else synth.vgg19_state_dict(), wfile)
    scaling = os.path.join(tmp, "scaling.json")
    with open(scaling, "w") as f:
        json.dump({"100000": {"gpu": "0", "multidevice": False}}, f)
    args = config.get_args(["--content", "c.png", "--style", "s.png", "--model_file", wfile, "--disable_check",
                            "--scaling_args", scaling, "--optimizer", a.optimizer, "--image_sizes", str(S),
                            "--num_iters", str(a.steps), "--seed", "0", "--no_hist_match", "--lbfgs_num_correction",
                            str(a.history)] +
                           (["--use_covariance", "--style_layers", "relu1,relu3,relu5,relu7,relu9,relu11", "--content_layers", "relu8"]
                            if nin else []))
    a.hip_graph = not a.no_hip_graph
    args.hip_graph = a.hip_graph
    optim.set_model_args(args, S)
    net, losses = models.load_model(args)
    dist.broadcast_network(net, src=0)  # one RCCL broadcast of the replica (no-op at N=1)

    content, style, _ = synth.images(S)
    init = synth.images(S, seed=100 + rank)[2]  # every rank optimises its own image
    optim.set_content_targets(net, synth.images(S, seed=200 + rank)[0] if world > 1 else content, args)
    if rank == 0:
        optim.set_style_targets(net, [style], args)
    dist.broadcast_style_targets(net, src=0)
    for m in net.style_losses:
        m.mode = "none"
    for m in losses:
        m.mode = "loss"

    opt = optim.PixelOptimizer(net, losses, init, args)
    prefill = 0 if (a.no_prefill or a.optimizer != "lbfgs") else a.history
    for _ in range(prefill + a.warmup):
        opt.step()
    torch.cuda.synchronize()

    # Timed region: K iterations the way the product runs them (one hipGraph replay per iteration unless --no_hip_graph).
    # A graph's kernels cannot be bracketed one by one, so the per-launch HIP events of the roofline come from a second
    # pass of K eager iterations right after it (same kernels, same stream); with --no_hip_graph both are the same pass.
    timer = []
    if opt.engine is not None and not a.hip_graph:
        opt.engine.timer = timer
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        opt.step()
    torch.cuda.synchronize()
    dist.barrier()
    mine = time.perf_counter() - t0
    # (every rank checks its own optimiser: a stop flag raised inside the timed region voids the rate)
    moving = lbfgs_still_moving(opt, prefill + a.warmup + a.steps, "the headline") if a.optimizer == "lbfgs" else None
    elapsed = dist.max_over_ranks(mine)
    per_rank = dist.gather_floats(a.steps / mine)  # iterations/s of every rank (rank order)
    # The same K-step region again, `--repeats` times (each bracketed like the headline one): its spread inside THIS run.
    repeats = []
    if not a.no_repeats:
        for _ in range(max(0, a.repeats)):
            dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(a.steps):
                opt.step()
            torch.cuda.synchronize()
            dist.barrier()
            repeats.append(dist.max_over_ranks(time.perf_counter() - t1))
    eager_ms = None
    lbfgs_events = []
    if opt.engine is not None and a.hip_graph and rank == 0:
        opt.engine.timer = timer
        if a.optimizer == "lbfgs":  # the five launches of one maua_lbfgs_iterate call, bracketed together (HIP events, launch stream)
            real_iterate = opt.state.iterate

            def timed_iterate(*args_, **kw):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                out_ = real_iterate(*args_, **kw)
                e1.record()
                lbfgs_events.append((e0, e1))
                return out_
            opt.state.iterate = timed_iterate
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            opt.step()
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t1) / a.steps * 1e3
        if a.optimizer == "lbfgs":
            opt.state.iterate = real_iterate
    if opt.engine is not None:
        opt.engine.timer = None

    want_sizes = not a.no_extra_sizes and a.model == "vgg19" and (S == 1024 or a.extra_sizes is not None)
    extra_sizes = [int(v) for v in (a.extra_sizes or "512,256").split(",") if v]
    if rank != 0:
        if want_sizes:  # every rank runs the other sizes on its own GPU (north_star: 512x512 at 1, 2, 4 and 8 GPUs too)
            del opt
            for sz in extra_sizes:
                steady_rate(sz, max(a.steps, 100) if a.extra_sizes is None else a.steps, a.optimizer, a.history)
        return
    status = opt.state.status() if a.optimizer == "lbfgs" else {}
    work = algorithmic_work(S, a.history)
    ms = elapsed / a.steps * 1e3
    # dominant kernel: the MFMA convolution (forward + backward-data launches)
    x6 = opt.engine is not None and opt.engine.x6_fwd and opt.engine.x6_bwd
    dominant = "conv3x3_split" if x6 else "conv"  # with bf16x6 on, conv1_1 (3 channels) runs other kernels: not counted here
    conv = [(fl, e0.elapsed_time(e1)) for tag, fl, nb, e0, e1 in timer if tag.startswith(dominant)]
    roofline = None
    x3w = x6 and models._x3_enabled() and models._x3w_enabled()
    x3q = x3w and models._x3q_min_channels() > 0
    pmc = pmc_traffic(((("maua::conv_x3w_kernel<", "maua::conv_x3q_kernel<", "maua::conv_x3p_kernel<") if x3w else "maua::conv_x3_kernel<") if models._x3_enabled() else
                       "maua::conv_x6_kernel<") if x6 else "maua::conv_mfma2_kernel<") if S == 1024 else None
    if conv:
        tot_fl, tot_ms = sum(c[0] for c in conv), sum(c[1] for c in conv)
        achieved = tot_fl / (tot_ms * 1e-3) / 1e12
        by_tag = {}
        for tag, fl, nb, e0, e1 in timer:
            d = by_tag.setdefault(tag, [0, 0.0, 0, 0])
            d[0] += 1
            d[1] += e0.elapsed_time(e1)
            d[2] += fl
            d[3] += nb
        # fp32-accurate products on the bf16 matrix cores cost six MFMAs each: the attainable rate of ALGORITHMIC
        # (fp32-equivalent) FLOPs is the dense bf16 peak / 6; with the fp32 matrix cores it is the fp32 MFMA peak
        x3 = x6 and models._x3_enabled()
        per_product = 3 if x3 else X6_MFMAS_PER_PRODUCT
        peak = BF16_MFMA_PEAK_TFLOPS / per_product if x6 else FP32_MFMA_PEAK_TFLOPS
        roofline = {"bound": "mfma",
                    "kernel": ((("conv_x3w_kernel + conv_x3q_kernel + conv_x3p_kernel (3x3 conv fwd + bwd-data, fp16x3: 16-channel chunks on 32x32x16 MFMAs "
                                 "or 32-channel chunks on 16x16x32 MFMAs, per launch as `extra.routes` lists)" if x3q else
                                 "conv_x3w_kernel (3x3 conv fwd + bwd-data, fp16x3, 16-channel chunks)") if x3w else
                                "conv_x3_kernel (3x3 conv fwd + bwd-data, fp16x3)") if x3 else
                               "conv_x6_kernel (3x3 conv fwd + bwd-data, bf16x6)") if x6 else "conv_mfma2_kernel (fwd + bwd-data)",
                    "achieved": round(achieved, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
                    "peak_note": (f"algorithmic fp32-equivalent FLOPs; peak = 2500 TFLOP/s dense {'fp16' if x3 else 'bf16'} / "
                                  f"{per_product} MFMAs per product block; 16-bit multiply-accumulate work = achieved x {per_product}"
                                  + ("; every tap is a full-K MFMA step, so that is also the matrix-pipe time; the kernels are power-bound "
                                     "(in-kernel clock stamps: profiles/probes_r02.md, probes_r04.md)" if x3w else
                                     f"; matrix-pipe time = achieved x {per_product} x 10/9 (the ninth tap's K=8 MFMA holds the pipe as "
                                     "long as a K=16 one: 5 steps for 4.5)")) if x6 else "fp32 MFMA peak",
                    "hw_16bit_tflops": round(achieved * per_product, 1) if x6 else None,
                    "hw_pipe_equiv_tflops": round(achieved * per_product * (1.0 if x3w else 10 / 9), 1) if x6 else None,
                    "traffic": pmc["bytes"] if pmc else None, "traffic_note": pmc["note"] if pmc else None,
                    "algorithmic_bytes_per_launch": round(sum(nb for tag, fl, nb, e0, e1 in timer if tag.startswith(dominant)) / len(conv)),
                    "events_from": ("second pass of K eager iterations (the timed region replays a hipGraph)" if eager_ms is not None
                                    else "the timed region"),
                    "eager_ms_per_step_with_events": round(eager_ms, 4) if eager_ms is not None else None,
                    "launches": len(conv), "avg_launch_ms": round(tot_ms / len(conv), 4),
                    "flops_per_launch_avg": tot_fl / len(conv),
                    "per_kernel_ms_per_step": {k: round(v[1] / a.steps, 4) for k, v in by_tag.items()},
                    "per_kernel_tflops": {k: round(v[2] / (v[1] * 1e-3) / 1e12, 2) for k, v in by_tag.items() if v[1] > 0},
                    "per_kernel_alg_gbs": {k: round(v[3] / (v[1] * 1e-3) / 1e9, 1) for k, v in by_tag.items() if v[1] > 0}}
    out = {
        "metric": "optimizer iterations/sec at 1024x1024 VGG-19" if S == 1024 else f"optimizer iterations/sec at {S}x{S} VGG-19",
        "value": round(a.steps * world / elapsed, 4), "unit": "iterations/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": (("f32 (fp16x3 split-precision MFMA, fp32 accumulate)" if models._x3_enabled() else
                   "f32 (bf16x6 split-precision MFMA, fp32 accumulate)")
                  if (opt.engine is not None and opt.engine.x6_fwd) else "f32"),
        "dtype_note": ((("3x3 convs: power-of-two-scaled 2-way fp16 split of both operands = 22 of 24 significand bits, three fp16 MFMAs "
                         "per product block, fp32 accumulate; image layer on exact bf16x6 products; accuracy of the timed kernels against "
                         "fp64, measured in this run: `accuracy_probe`; whole-network pixel gradient: "
                         "tests/test_engine_gpu.py::test_pixel_gradient_is_as_close_to_fp64_as_the_reference_fp32") if models._x3_enabled() else
                        "3x3 convs: exact 3-way bf16 split of both operands on the bf16 matrix cores, fp32 accumulate")
                       if (opt.engine is not None and opt.engine.x6_fwd) else None),
        "data": "synthetic",
        "config": {"workload": f"{S}x{S} single-scale VGG-19 Gram style transfer, {a.optimizer.upper()}"
                               f"{' history ' + str(a.history) + ' (full)' if prefill else ''}, one image per GPU, "
                               "content 5 / style 100 / tv 1e-3, normalize_gradients, seeded synthetic weights and images",
                   "image_size": S, "optimizer": a.optimizer, "lbfgs_history_len": status.get("history_len"),
                   "lbfgs": moving,  # {n_iter, stopped, ...} read right behind the timed region (the run fails if a stop rule fired in it)
                   "parallelism": f"frames x{world} (replicas, one broadcast, no per-iteration collective)",
                   "hip_graph": bool(a.hip_graph)},
        "rccl_ranks": dist.group_size(), "dist_backend": dist.backend_name(),
        "per_rank_iterations_per_s": [round(v, 3) for v in per_rank],
        "model_flops_per_step": work["flops"],
        "whole_step": {"tflops": round(work["flops"] / (ms * 1e-3) / 1e12, 2),
                       "frac_fp32_mfma_peak": round(work["flops"] / (ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4),
                       "alg_gbs": round((work["bytes_feval"] + (work["bytes_lbfgs"] if a.optimizer == "lbfgs" else 0)) / (ms * 1e-3) / 1e9, 1),
                       "frac_hbm_peak": round((work["bytes_feval"] + (work["bytes_lbfgs"] if a.optimizer == "lbfgs" else 0)) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
        "roofline": roofline,
    }
    if nin:
        # BASELINE config 5.  Its dominant kernels are the two L-BFGS history sweeps (a third of the iteration: profiles/): HBM roofline.
        # Algorithmic bytes of one update: (4 m + 10) n 4 B (the two-loop recursion touches 2 m vectors twice + ~10
        # vector passes), n = 3 S^2, m = pairs held; measured with HIP events around the five launches of maua_lbfgs_iterate.
        n_px, m_hist = 3 * S * S, int(status.get("history_len") or a.history)
        lb_bytes = (4 * m_hist + 10) * n_px * 4
        out["metric"] = f"optimizer iterations/sec at {S}x{S} NIN + covariance (BASELINE config 5)"
        out["config"]["workload"] = (f"{S}x{S} NIN (models.py:74-113) + --use_covariance, style relu1,3,5,7,9,11 / content relu8, "
                                     f"{a.optimizer.upper()} history {m_hist}, one image per GPU, seeded synthetic weights and images")
        flops5, bytes5 = (115.9e9, 1.291e9) if S == 1024 else (None, None)  # SURVEY.md section 8d, config 5 work
        out["model_flops_per_step"] = flops5
        out["whole_step"] = None if flops5 is None else {
            "tflops": round(flops5 / (ms * 1e-3) / 1e12, 2), "alg_gbs": round((bytes5 + lb_bytes) / (ms * 1e-3) / 1e9, 1),
            "frac_hbm_peak": round((bytes5 + lb_bytes) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        conv_part = roofline
        roofline = None
        if lbfgs_events:
            lb_ms = sum(e0.elapsed_time(e1) for e0, e1 in lbfgs_events) / len(lbfgs_events)
            ach = lb_bytes / (lb_ms * 1e-3) / 1e9
            roofline = {"bound": "hbm", "kernel": "lbfgs_pair_dots_kernel + lbfgs_combine_v4_kernel (the two history sweeps; the five launches of "
                                                  "one maua_lbfgs_iterate bracketed together)",
                        "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                        "algorithmic_bytes_per_launch": lb_bytes, "avg_launch_ms": round(lb_ms, 4), "launches": len(lbfgs_events),
                        "share_of_step": round(lb_ms / ms, 3), "traffic": (pmc_traffic_lbfgs() or {}).get("bytes"),
                        "traffic_note": (pmc_traffic_lbfgs() or {"note": "no PMC pass committed for this workload"})["note"],
                        "events_from": "second pass of K eager iterations (the timed region replays a hipGraph)",
                        "eager_ms_per_step_with_events": round(eager_ms, 4) if eager_ms is not None else None,
                        "matrix_kernels": None if conv_part is None else {
                            k: conv_part[k] for k in ("per_kernel_ms_per_step", "per_kernel_tflops", "per_kernel_alg_gbs") if k in conv_part}}
        out["roofline"] = roofline
    extra = {}
    # What ran, so that the line describes itself: the kernel every convolution launch of one evaluation took (as the engine planned and
    # launched it: one more eager evaluation with the route log on) and the planner settings that are not at their defaults, with any
    # MAUA_* environment variable that was found and ignored (plan.py).
    import plan
    if opt.engine is not None:
        extra["routes"] = [" ".join([r["pass"], r["kernel"], f"{r['consumed']}->{r['produced']}", f"@{r['plane'][0]}x{r['plane'][1]}",
                                     f"[{r['tile']}]", f"ks{r['ksplit']}"] + [k for k in ("relu", "mask", "pool", "unpool", "gram", "gram_slabs", "accumulate") if r.get(k)])
                           for r in opt.engine.describe_routes(opt.x if hasattr(opt, "x") else init.to(dev))]
    extra["env_overrides"] = plan.env_overrides()
    if repeats:
        import statistics
        rates = sorted(a.steps * world / t for t in repeats)
        extra["repeats"] = {"regions": len(repeats), "steps_each": a.steps,
                            "iterations_per_s": {"median": round(statistics.median(rates), 3), "min": round(rates[0], 3),
                                                 "max": round(rates[-1], 3)},
                            "ms_per_step": {"median": round(statistics.median(repeats) / a.steps * 1e3, 4),
                                            "min": round(min(repeats) / a.steps * 1e3, 4),
                                            "max": round(max(repeats) / a.steps * 1e3, 4)},
                            "note": "the K-step timed region repeated after the headline region (which stays `value`), each "
                                    "bracketed by barrier + synchronize, max over ranks"}
    if a.requested_gpus:
        out["requested_gpus"] = a.requested_gpus
        out["note_gpus"] = f"{a.requested_gpus} GPUs requested, {world} visible: ran on {world} (--allow_fewer)"
    if want_sizes:
        del opt  # frees the 2.5 GB history slab before the next job allocates its own
        extra["other_sizes"] = [steady_rate(sz, max(a.steps, 100) if a.extra_sizes is None else a.steps, a.optimizer, a.history)
                                for sz in extra_sizes]
    if exact_child is not None:
        torch.cuda.empty_cache()
        extra["exact_split"] = finish_exact_split_child(exact_child)
    if extra:
        out["extra"] = extra
    if world == 1 and not a.no_accuracy_probe and a.model == "vgg19" and models._x3_enabled() and models._x3w_enabled():
        out["accuracy_probe"] = accuracy_probe()
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(S, a.optimizer, model=a.model)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
