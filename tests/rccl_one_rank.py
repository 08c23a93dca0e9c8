"""Helper of tests/test_cli_gpu.py::test_rccl_group_broadcast_and_graph_replay (run as its own process: it owns a process group).

The multi-GPU job's communication on the one GPU a test box has: a REAL `nccl` (= RCCL) process group of one rank, the
start-up broadcasts of dist.py over it (conv weights as one flat buffer, style targets with their shape exchange), the
barrier / max / gather helpers bench.py brackets its timed region with, then PixelOptimizer's hipGraph capture and replays
while the communicator is alive and collectives keep happening between replays, and the end-of-job barrier on the
long-timeout group."""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "maua-style_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
import torch  # noqa: E402
import torch.distributed as td  # noqa: E402

import config  # noqa: E402
import dist  # noqa: E402
import models  # noqa: E402
import optim  # noqa: E402
import synth  # noqa: E402

torch.cuda.set_device(0)
td.init_process_group("nccl", rank=0, world_size=1)          # dist.init() skips the group for one rank: formed here
dist._LONG_GROUP = td.new_group(ranks=[0], backend="nccl")
assert dist.backend_name() == "nccl" and dist.group_size() == 1
x = torch.ones(1 << 20, device="cuda")
td.broadcast(x, src=0)
td.all_reduce(x)
torch.cuda.synchronize()

tmp = tempfile.mkdtemp()
w = os.path.join(tmp, "vgg19_synth.pth")
torch.save(synth.vgg19_state_dict(), w)
sc = os.path.join(tmp, "s.json")
json.dump({"100000": {"gpu": "0", "multidevice": False}}, open(sc, "w"))
args = config.get_args(["--content", "c.png", "--style", "s.png", "--model_file", w, "--disable_check", "--scaling_args", sc,
                        "--image_sizes", "256", "--num_iters", "10", "--seed", "0", "--no_hist_match"])
optim.set_model_args(args, 256)
net, losses = models.load_model(args)
before = [p.detach().clone() for p in net.parameters()]
dist.broadcast_network(net, src=0)                            # ONE flat RCCL broadcast of the replica
torch.cuda.synchronize()
assert all(torch.equal(a, b.detach()) for a, b in zip(before, net.parameters()))
assert sum(p.numel() for p in net.parameters()) == 12944960   # conv weights + biases through conv5_1: the 51.8 MB of SURVEY section 5
content, style, init = synth.images(256)
optim.set_content_targets(net, content, args)
optim.set_style_targets(net, [style], args)
targets = [m.target.clone() for m in net.style_losses]
dist.broadcast_style_targets(net, src=0)                      # shapes, then one flat broadcast of the Gram targets
torch.cuda.synchronize()
assert all(torch.equal(a, m.target) for a, m in zip(targets, net.style_losses))
assert all(m.target.is_cuda for m in net.style_losses)
dist.barrier()
assert dist.max_over_ranks(3.5) == 3.5 and dist.gather_floats(1.25) == [1.25]
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
st = opt.state.status()
print("graph active:", opt._graph is not None, " status:", st)
assert opt._graph is not None and st["n_iter"] == 60
dist.end_of_job_barrier()
td.destroy_process_group()
print("OK")
