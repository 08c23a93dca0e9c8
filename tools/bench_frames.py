"""Frame-batch throughput (vid_img's independent frames): frame-iterations/s for several batch sizes, eager and graph replay.
    python tools/bench_frames.py [S] [ITERS] [B ...]"""
import json
import os
import sys
import tempfile
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import config  # noqa: E402
import models  # noqa: E402
import optim  # noqa: E402
import synth  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 50
batches = [int(v) for v in sys.argv[3:]] or [1, 4, 8, 16]
tmp = tempfile.mkdtemp()
wfile = os.path.join(tmp, "vgg19_synth.pth")
torch.save(synth.vgg19_state_dict(), wfile)
scaling = os.path.join(tmp, "scaling.json")
json.dump({"100000": {"gpu": "0", "multidevice": False}}, open(scaling, "w"))
args = config.get_args(["--content", "c.png", "--style", "s.png", "--model_file", wfile, "--disable_check", "--scaling_args", scaling,
                        "--image_sizes", str(S), "--num_iters", str(N), "--seed", "0", "--no_hist_match"])
optim.set_model_args(args, S)
net, losses = models.load_model(args)
style = synth.images(S)[1].cuda()
for B in batches:
    contents = torch.cat([synth.images(S, seed=50 + k)[0] for k in range(B)]).cuda()
    for graph in (False, True):
        args.hip_graph = graph
        if B == 1:
            run = lambda: optim.optimize(contents, [style], contents.clone(), N, args, net, losses, keep_on_device=True)
        else:
            run = lambda: optim.optimize_frames(contents, [style], contents.clone(), N, args, net, losses)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 2
        print(f"S={S} B={B:2d} {'graph' if graph else 'eager'}: {dt * 1e3:8.1f} ms per call of {N} iterations = {dt / N * 1e3:7.3f} ms / iteration = "
              f"{B * N / dt:7.1f} frame-iterations/s (incl. target capture)", flush=True)
