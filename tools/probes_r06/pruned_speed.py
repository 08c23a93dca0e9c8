"""Probe (round 6): how fast the reference's other VGG stacks evaluate - VGG-16 and the channel-pruned VGG-16 of the reference's scaling table
(rows 3760 / 4096: "prune") - beside VGG-19 at the same size, and which kernel family every launch of the pruned stack takes.
    python tools/probes_r06/pruned_speed.py [sizes, default 1024 2048]"""
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
files = {"vgg19": os.path.join(d, "vgg19_synth.pth"), "nin": os.path.join(d, "nin_synth.pth"), "vgg16": os.path.join(d, "vgg16_synth.pth"),
         "prune": os.path.join(d, "vgg16-prune_synth.pth")}
torch.save(synth.vgg19_state_dict(), files["vgg19"])
torch.save(synth.nin_state_dict(), files["nin"])
torch.save(synth.vgg19_state_dict(channels=synth.VGG16_CHANNELS), files["vgg16"])
torch.save(synth.vgg19_state_dict(channels=synth.VGG16P_CHANNELS), files["prune"])
import engine  # noqa: E402
import models  # noqa: E402
import optim  # noqa: E402

for S in [int(a) for a in sys.argv[1:]] or [1024, 2048]:
    for model in ("vgg19", "vgg16", "prune"):
        args = product_args(files, [], model=model, optimizer="adam", S=S, N=3)
        content, style, init = synth.images(S)
        optim.set_model_args(args, S)
        net, losses = models.load_model(args)
        optim.set_content_targets(net, content, args)
        optim.set_style_targets(net, [style], args)
        for m in losses:
            m.mode = "loss"
        eng = engine.StyleEngine(net, losses)
        x = init.cuda()
        for _ in range(3):
            eng.feval(x)
        torch.cuda.synchronize()
        t = time.time()
        for _ in range(10):
            eng.feval(x)
        torch.cuda.synchronize()
        ms = (time.time() - t) / 10 * 1e3
        print(f"S={S} {model:6s} {ms:8.2f} ms / evaluation (eager launches)", flush=True)
        if model == "prune":
            for r in eng.describe_routes(x):
                print(f"      {r['pass']} {r['consumed']:4d} -> {r['produced']:4d} @ {r['plane'][0]:5d}  {r['kernel']}  ksplit {r['ksplit']}")
        del eng, net, losses
        torch.cuda.empty_cache()
