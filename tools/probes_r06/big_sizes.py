"""Probe (round 6): the largest rows of the shipped and of the reference's scaling tables on one MI355X - NIN at 5312 / 6896 pixels
(reference config/scaling-img.json), VGG-19 at 8192 (the shipped table's last row): does the evaluation run, is it deterministic, is its
gradient the slope of its loss, how long does it take, how much memory does it hold.
    python tools/probes_r06/big_sizes.py [nin:5312 nin:6896 vgg19:8192 ...]"""
import os
import sys
import tempfile
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "maua-style_amd"), os.path.join(REPO, "tests")]
import synth  # noqa: E402
from conftest import product_args  # noqa: E402

d = tempfile.mkdtemp()
files = {"vgg19": os.path.join(d, "vgg19_synth.pth"), "nin": os.path.join(d, "nin_synth.pth")}
torch.save(synth.vgg19_state_dict(), files["vgg19"])
torch.save(synth.nin_state_dict(), files["nin"])
NIN = ["--style_layers", "relu1,relu3,relu5,relu7,relu9,relu11", "--content_layers", "relu8"]

for case in sys.argv[1:] or ["nin:5312", "nin:6896", "vgg19:8192"]:
    model, S = case.split(":")
    S = int(S)
    import engine
    import models
    import optim
    torch.cuda.reset_peak_memory_stats()
    try:
        args = product_args(files, ["--no_grad_norm"] + (NIN if model == "nin" else []), model=model, optimizer="adam", S=S, N=3)
        content, style, init = synth.images(S)
        optim.set_model_args(args, S)
        net, losses = models.load_model(args)
        optim.set_content_targets(net, content, args)
        optim.set_style_targets(net, [style], args)
        for m in losses:
            m.mode = "loss"
        eng = engine.StyleEngine(net, losses)
        x = init.cuda()
        s0, t0, g0 = [t.clone() for t in eng.feval(x)]
        s1, t1, g1 = eng.feval(x)
        torch.cuda.synchronize()
        same = torch.equal(g0, g1) and torch.equal(s0, s1)
        t = time.time()
        for _ in range(3):
            eng.feval(x)
        torch.cuda.synchronize()
        ms = (time.time() - t) / 3 * 1e3
        v = g0 / g0.norm()
        slope = float((g0.double() * v.double()).sum())
        eps = 4.0
        fd = (float(eng.feval(x + eps * v)[1]) - float(eng.feval(x - eps * v)[1])) / (2 * eps)
        bands = g0.abs().reshape(1, 3, 8, S // 8, S).amax(dim=(1, 3, 4)).flatten()
        out = optim.optimize(content, [style], init.clone(), 3, args, net, losses)
        after = float(eng.feval(out.cuda())[1])
        print(f"{case}: deterministic {same}  loss {float(t0):.6e} -> {after:.6e} after 4 Adam steps  slope {slope:.5e} fd {fd:.5e} ({abs(fd - slope) / abs(slope):.1e})  "
              f"{ms:.1f} ms / evaluation  peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB  dead bands {int((bands == 0).sum())}  finite {bool(torch.isfinite(g0).all())}", flush=True)
        del eng, net, losses, x, g0, g1, out
    except Exception as e:  # noqa: BLE001
        print(f"{case}: FAILED {type(e).__name__}: {str(e)[:300]}", flush=True)
    torch.cuda.empty_cache()
