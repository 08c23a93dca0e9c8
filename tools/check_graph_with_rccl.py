"""hipGraph capture next to a live RCCL communicator (the multi-GPU bench / vid_img situation), on one GPU:
a 1-rank nccl process group with a real collective behind it, then PixelOptimizer's capture + replays."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "maua-style_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
import torch  # noqa: E402
import torch.distributed as td  # noqa: E402

td.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
x = torch.ones(1 << 20, device="cuda")
td.broadcast(x, src=0)
td.all_reduce(x)
torch.cuda.synchronize()

import tempfile  # noqa: E402
import json  # noqa: E402
import config  # noqa: E402
import models  # noqa: E402
import optim  # noqa: E402
import synth  # noqa: E402
import dist  # noqa: E402

tmp = tempfile.mkdtemp()
w = os.path.join(tmp, "vgg19_synth.pth")
torch.save(synth.vgg19_state_dict(), w)
sc = os.path.join(tmp, "s.json")
json.dump({"100000": {"gpu": "0", "multidevice": False}}, open(sc, "w"))
args = config.get_args(["--content", "c.png", "--style", "s.png", "--model_file", w, "--disable_check", "--scaling_args", sc,
                        "--image_sizes", "256", "--num_iters", "10", "--seed", "0", "--no_hist_match"])
optim.set_model_args(args, 256)
net, losses = models.load_model(args)
dist.broadcast_network(net, src=0)
content, style, init = synth.images(256)
optim.set_content_targets(net, content, args)
optim.set_style_targets(net, [style], args)
for m in losses:
    m.mode = "loss"
opt = optim.PixelOptimizer(net, losses, init, args)
for i in range(30):
    opt.step()
    if i % 10 == 0:
        td.all_reduce(x)  # collectives keep happening between replays (the watchdog thread stays busy)
torch.cuda.synchronize()
time.sleep(2.0)
for i in range(30):
    opt.step()
torch.cuda.synchronize()
print("graph active:", opt._graph is not None, " status:", opt.state.status())
td.destroy_process_group()
print("OK")
