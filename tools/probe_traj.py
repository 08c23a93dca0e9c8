"""GPU probe: per-move deviation of the HIP L-BFGS trajectory from the fp64 oracle, next to the fp32 oracle's."""
import os, sys, tempfile
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd"), os.path.join(REPO, "tests")]
torch.set_num_threads(1)
import synth
from conftest import product_args, make_cfg, rel_l2
import models, optim
from oracle import optimize as oracle_optimize

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
m = int(sys.argv[3]) if len(sys.argv) > 3 else 3
d = tempfile.mkdtemp()
wf = {"vgg19": os.path.join(d, "vgg19_synth.pth"), "nin": os.path.join(d, "nin_synth.pth")}
sd = synth.vgg19_state_dict()
torch.save(sd, wf["vgg19"])
content, style, init = synth.images(S)
t64, t32 = [], []
oracle_optimize(content, [style], init, N, make_cfg(lbfgs_num_correction=m), sd, dtype=torch.float64, trace=t64)
oracle_optimize(content, [style], init, N, make_cfg(lbfgs_num_correction=m), sd, dtype=torch.float32, trace=t32)
args = product_args(wf, ["--lbfgs_num_correction", str(m)], S=S, N=N)
optim.set_model_args(args, S)
net, losses = models.load_model(args)
optim.set_content_targets(net, content, args); optim.set_style_targets(net, [style], args)
for mod in losses: mod.mode = "loss"
opt = optim.PixelOptimizer(net, losses, init, args)
for i in range(optim.lbfgs_moves(N)):
    opt.step(); torch.cuda.synchronize()
    x = opt.x.cpu().flatten()
    st = opt.state.status()
    print(f"move {i+1:2d}  hip-f64 {rel_l2(x, t64[i]):.3e}   cpu32-f64 {rel_l2(t32[i], t64[i]):.3e}   hip-cpu32 {rel_l2(x, t32[i]):.3e}  hist {st['history_len']} t {st['t']:.3e} gtd {st['gtd']:.4e}")
