"""Per-launch table of the 3x3 convolution launches of one VGG-19 iteration at a given image size: HIP events around every launch (the
engine's timer, eager launches), medians over the iterations, in launch order (forward conv1_1 ... conv5_1, then backward conv5_1 ... conv1_2; a
launch = the entry point's kernel plus its split-K finishing kernel where it has one; conv1_1's backward is the 64 -> 3 kernel, not listed).  python tools/layer_table.py SIZE [iterations]   """
import json, os, statistics, sys, tempfile
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "maua-style_amd"), os.path.join(REPO, "tests")]
import config, models, optim, synth  # noqa: E402

size = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 12
tmp = tempfile.mkdtemp(prefix="maua_lt_")
wfile = os.path.join(tmp, "vgg19_synth.pth"); torch.save(synth.vgg19_state_dict(), wfile)
scaling = os.path.join(tmp, "scaling.json"); json.dump({"100000": {"gpu": "0", "multidevice": False}}, open(scaling, "w"))
args = config.get_args(["--content", "c.png", "--style", "s.png", "--model_file", wfile, "--disable_check", "--scaling_args", scaling, "--optimizer", "lbfgs",
                        "--image_sizes", str(size), "--num_iters", "100", "--seed", "0", "--no_hist_match"])
args.hip_graph = False
optim.set_model_args(args, size)
net, losses = models.load_model(args)
content, style, init = synth.images(size)
optim.set_content_targets(net, content, args); optim.set_style_targets(net, [style], args)
for m in losses: m.mode = "loss"
opt = optim.PixelOptimizer(net, losses, init, args)
for _ in range(5): opt.step()
rows = None
for _ in range(iters):
    timer = []; opt.engine.timer = timer
    opt.step(); torch.cuda.synchronize()
    opt.engine.timer = None
    conv = [(tag, fl, nb, e0.elapsed_time(e1) * 1e3) for tag, fl, nb, e0, e1 in timer if tag.startswith("conv3x3_split")]
    if rows is None: rows = [[c] for c in conv]
    elif len(conv) == len(rows):
        for r, c in zip(rows, conv): r.append(c)
names_f = ["conv1_1", "conv1_2", "conv2_1", "conv2_2", "conv3_1", "conv3_2", "conv3_3", "conv3_4", "conv4_1", "conv4_2", "conv4_3", "conv4_4", "conv5_1"]
names = [n + " fwd" for n in names_f] + [n + " bwd" for n in reversed(names_f[1:])]
print(f"{size} x {size}: {len(rows)} launches per iteration; peak = 2500 / 3 = 833.3 TFLOP/s of fp32-equivalent work")
tot_fl = tot_us = 0.0
for i, r in enumerate(rows):
    tag, fl = r[0][0], r[0][1]
    us = statistics.median(c[3] for c in r)
    tot_fl += fl; tot_us += us
    nm = names[i] if len(rows) == len(names) else f"launch {i}"
    print(f"  {nm:13s} {tag:22s} {fl / 1e9:7.2f} GFLOP  {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  frac {fl / us / 1e6 / 833.3:5.3f}  MFMA floor {fl / 833.3e6:6.1f} us")
print(f"  sum {tot_fl / 1e9:.1f} GFLOP in {tot_us:.0f} us: frac {tot_fl / tot_us / 1e6 / 833.3:.3f}")
