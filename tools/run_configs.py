"""Run BASELINE.json's configs 2-5 end to end on ONE MI355X through the product's own entry points and time them.

    python tools/run_configs.py [--configs 2,3,4,5] [--frames 64] [--out gpurun_out/configs.json]

Unlike bench.py (steady-state iterations of one size), these are whole jobs as a user of the reference would run them:
model build, target capture, every scale, PNG input / output and histogram matching included.  Synthetic inputs as in
SURVEY.md section 8(d): seeded weights (synth.py), seeded PNGs / frames.  Prints one JSON object per config.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "maua-style_amd")
sys.path.insert(0, PKG)

import numpy as np  # noqa: E402
import torch  # noqa: E402
from PIL import Image  # noqa: E402

import config  # noqa: E402
import optim  # noqa: E402
import style  # noqa: E402
import synth  # noqa: E402


def _png(path, side, seed):
    g = torch.Generator().manual_seed(seed)
    Image.fromarray((torch.rand(side, side, 3, generator=g) * 255).byte().numpy()).save(path)


class Timed:
    """Wraps optim.optimize / optim.optimize_frames: wall time (device-synchronised) and iteration count of every call
    (a batch of B frames counts B function evaluations per iteration)."""

    def __init__(self):
        self.calls = []
        self._orig = optim.optimize
        self._orig_frames = optim.optimize_frames

    def _record(self, init, num_iters, args, t0):
        steps = optim.lbfgs_moves(num_iters) if args.optimizer == "lbfgs" else num_iters + 1
        self.calls.append({"size": int(max(init.shape[2:])), "frames": int(init.shape[0]), "optimizer": args.optimizer,
                           "num_iters": int(num_iters), "fevals": int(steps) * int(init.shape[0]),
                           "seconds": round(time.perf_counter() - t0, 4)})

    def __enter__(self):
        def wrapped(content, styles, init, num_iters, args, net=None, losses=None, **kw):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = self._orig(content, styles, init, num_iters, args, net, losses, **kw)
            torch.cuda.synchronize()
            self._record(init, num_iters, args, t0)
            return out

        def wrapped_frames(contents, styles, inits, num_iters, args, net, losses, **kw):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            out = self._orig_frames(contents, styles, inits, num_iters, args, net, losses, **kw)
            torch.cuda.synchronize()
            self._record(inits, num_iters, args, t0)
            return out
        optim.optimize = wrapped
        optim.optimize_frames = wrapped_frames
        return self

    def __exit__(self, *exc):
        optim.optimize = self._orig
        optim.optimize_frames = self._orig_frames


def setup(tmp):
    paths = {"vgg19": os.path.join(tmp, "vgg19_synth.pth"), "nin": os.path.join(tmp, "nin_synth.pth")}
    torch.save(synth.vgg19_state_dict(), paths["vgg19"])
    torch.save(synth.nin_state_dict(), paths["nin"])
    with open(os.path.join(PKG, "config", "scaling-img.json")) as f:
        table = json.load(f)
    for entry in table.values():  # the stock table (L-BFGS <= 1456 px, Adam above) with the synthetic checkpoint
        entry["model_file"] = paths["vgg19"]
    paths["scaling"] = os.path.join(tmp, "scaling-img-synth.json")
    with open(paths["scaling"], "w") as f:
        json.dump(table, f)
    paths["scaling_adam"] = os.path.join(tmp, "scaling-adam-synth.json")
    with open(paths["scaling_adam"], "w") as f:
        json.dump({"100000": {"model_file": paths["vgg19"], "optimizer": "adam", "multidevice": False, "gpu": "0"}}, f)
    paths["scaling_nin"] = os.path.join(tmp, "scaling-nin-synth.json")
    with open(paths["scaling_nin"], "w") as f:
        json.dump({"100000": {"model_file": paths["nin"], "optimizer": "lbfgs", "multidevice": False, "gpu": "0"}}, f)
    paths["content"] = os.path.join(tmp, "content.png")
    paths["style"] = os.path.join(tmp, "style.png")
    _png(paths["content"], 256, 7)
    _png(paths["style"], 256, 8)
    return paths


def summarise(calls, wall):
    fevals = sum(c["fevals"] for c in calls)
    loop = sum(c["seconds"] for c in calls)
    return {"wall_seconds": round(wall, 3), "optimize_seconds": round(loop, 3), "fevals": fevals,
            "iterations_per_second_incl_setup": round(fevals / loop, 2) if loop else None}


def config2(p, tmp):
    args = config.get_args(["--content", "c.png", "--style", "s.png", "--model_file", p["vgg19"], "--disable_check",
                            "--scaling_args", p["scaling"], "--image_sizes", "512", "--num_iters", "500", "--seed", "0",
                            "--no_hist_match", "--output_dir", tmp])
    content, sty, init = synth.images(512)
    with Timed() as t:
        t0 = time.perf_counter()
        out = optim.optimize(content, [sty], init, 500, args)
        wall = time.perf_counter() - t0
    return {"config": 2, "workload": "512x512 single-scale VGG-19 Gram style transfer, L-BFGS 500 iters", "calls": t.calls,
            "finite": bool(torch.isfinite(out).all()), **summarise(t.calls, wall)}


def config3(p, tmp, scaling_key, label):
    out_dir = os.path.join(tmp, "c3_" + scaling_key)
    os.makedirs(out_dir, exist_ok=True)
    args = config.get_args(["--content", p["content"], "--style", p["style"], "--model_file", p["vgg19"], "--disable_check",
                            "--scaling_args", p[scaling_key], "--image_sizes", "256,512,1024,2048", "--num_iters",
                            "500,400,300,200", "--seed", "0", "--init", "content", "--output_dir", out_dir])
    torch.manual_seed(0)
    with Timed() as t:
        t0 = time.perf_counter()
        out = style.img_img(args)
        wall = time.perf_counter() - t0
    pngs = sorted(f for f in os.listdir(out_dir) if f.endswith(".png"))
    return {"config": 3, "workload": "multi-resolution 256->512->1024->2048, " + label + ", histogram matching on, PNG per scale",
            "calls": t.calls, "pngs_written": pngs, "finite": bool(torch.isfinite(out).all()), **summarise(t.calls, wall)}


def config4(p, tmp, frames):
    fdir = os.path.join(tmp, "frames")
    os.makedirs(fdir, exist_ok=True)
    g = torch.Generator().manual_seed(9)
    vid = (torch.rand(frames, 512, 512, 3, generator=g) * 255).byte().numpy()
    for i in range(frames):
        Image.fromarray(vid[i]).save(os.path.join(fdir, "%04d.png" % i))
    out_dir = os.path.join(tmp, "c4")
    os.makedirs(out_dir, exist_ok=True)
    args = config.get_args(["--load_args", "config/args-vid.json", "--content", fdir, "--style", p["style"], "--model_file",
                            p["vgg19"], "--disable_check", "--scaling_args", p["scaling"], "--image_sizes", "512",
                            "--num_iters", "200", "--seed", "0", "--output_dir", out_dir])
    torch.manual_seed(0)
    with Timed() as t:
        t0 = time.perf_counter()
        style.vid_img(args)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
    n_png = sum(len([f for f in fs if f.endswith(".png")]) for _, _, fs in os.walk(out_dir))
    s = summarise(t.calls, wall)
    return {"config": 4, "workload": f"{frames} synthetic 512x512 frames, args-vid.json (no optical flow), 200 iters/frame in 4 passes, "
                                    "L-BFGS, one GPU (all frames on this rank)", "frames_per_batch": style.frames_per_batch(512), "optimize_calls": len(t.calls),
            "pngs_written": n_png,
            "frames_per_second": round(frames / wall, 3), **s}


def config5(p, tmp):
    args = config.get_args(["--content", "c.png", "--style", "s.png", "--model_file", p["nin"], "--disable_check",
                            "--scaling_args", p["scaling_nin"], "--image_sizes", "1024", "--num_iters", "500", "--seed", "0",
                            "--no_hist_match", "--use_covariance", "--style_layers", "relu1,relu3,relu5,relu7,relu9,relu11",
                            "--content_layers", "relu8", "--output_dir", tmp])
    content, sty, init = synth.images(1024)
    with Timed() as t:
        t0 = time.perf_counter()
        out = optim.optimize(content, [sty], init, 500, args)
        wall = time.perf_counter() - t0
    return {"config": 5, "workload": "NIN + --use_covariance at 1024x1024, L-BFGS 500 iters", "calls": t.calls,
            "finite": bool(torch.isfinite(out).all()), **summarise(t.calls, wall)}


def config6(p, tmp, B=6, T=12, S=512, N=100):
    """SURVEY 8(f)-4 as a job: a T-frame pastiche at S x S optimised in sliding windows of B frames (optim.optimize with
    transfer_type img_vid): per-frame static Gram terms plus the (B C) x (B C) dynamic Gram term on the fused engine."""
    args = config.get_args(["--transfer_type", "img_vid", "--content", "c.png", "--style", "clip", "--model_file", p["vgg19"],
                            "--disable_check", "--scaling_args", p["scaling"], "--image_sizes", str(S), "--num_iters", str(N),
                            "--seed", "0", "--no_hist_match", "--output_dir", tmp])
    args.gram_frame_window = B
    g = torch.Generator().manual_seed(11)
    content = torch.rand(1, 3, S, S, generator=g) * 255 - 120
    clip = torch.rand(T, 3, S, S, generator=g) * 255 - 120
    init = torch.rand(T, 3, S, S, generator=g) * 255 - 120
    n_windows = len(optim.video_windows(init, [clip], B)[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = optim.optimize(content, [clip], init, N, args)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    fevals = n_windows * optim.lbfgs_moves(N)
    return {"config": 6, "workload": f"img_vid: {T}-frame pastiche at {S}x{S}, windows of B = {B} frames ({n_windows} windows), "
                                    f"{T}-frame style clip, avg_frame_window {args.avg_frame_window}, L-BFGS {N} iters per window",
            "finite": bool(torch.isfinite(out).all()), "wall_seconds": round(wall, 3), "window_fevals": fevals,
            "window_iterations_per_second_incl_setup": round(fevals / wall, 2),
            "frame_iterations_per_second_incl_setup": round(fevals * B / wall, 2)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--configs", default="2,3,4,5")
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    want = {int(c) for c in a.configs.split(",")}
    os.chdir(PKG)  # relative preset paths (config/...) as when style.py is run from its directory
    results = []
    with tempfile.TemporaryDirectory() as tmp:
        p = setup(tmp)
        torch.zeros(1, device="cuda")
        if 2 in want:
            results.append(config2(p, tmp))
        if 3 in want:
            results.append(config3(p, tmp, "scaling", "stock scaling table (L-BFGS to 1024, Adam at 2048)"))
            results.append(config3(p, tmp, "scaling_adam", "all-Adam table"))
        if 4 in want:
            results.append(config4(p, tmp, a.frames))
        if 5 in want:
            results.append(config5(p, tmp))
        if 6 in want:
            results.append(config6(p, tmp))
    for r in results:
        print(json.dumps(r), flush=True)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(os.path.join(ROOT, a.out))), exist_ok=True)
        with open(os.path.join(ROOT, a.out), "w") as f:
            json.dump({"device": torch.cuda.get_device_name(0), "results": results}, f, indent=1)


if __name__ == "__main__":
    main()
