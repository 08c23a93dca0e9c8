"""GPU probe: how well does y = g(x1) - g(x0) survive fp32 on the HIP path vs the CPU fp32 oracle (both vs fp64)?"""
import os, sys, tempfile
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd"), os.path.join(REPO, "tests")]
torch.set_num_threads(1)
import synth
from conftest import product_args, make_cfg, rel_l2
import models, optim, engine
from oracle import OracleNet, build_spec

S = int(sys.argv[1]) if len(sys.argv) > 1 else 32
d = tempfile.mkdtemp()
wf = {"vgg19": os.path.join(d, "vgg19_synth.pth"), "nin": os.path.join(d, "nin_synth.pth")}
sd = synth.vgg19_state_dict(); torch.save(sd, wf["vgg19"])
content, style, init = synth.images(S)
def oracle(dtype):
    net = OracleNet(build_spec(make_cfg()), sd, dtype); net.capture_content(content); net.capture_style([style], [1.0]); return net
o64, o32 = oracle(torch.float64), oracle(torch.float32)
_, _, g0_64 = o64.feval(init); _, _, g0_32 = o32.feval(init)
t = min(1.0, 1.0 / float(g0_32.abs().sum()))
x1 = (init + t * (-g0_32)).float()
_, _, g1_64 = o64.feval(x1); _, _, g1_32 = o32.feval(x1)
args = product_args(wf, S=S); optim.set_model_args(args, S)
net, losses = models.load_model(args)
optim.set_content_targets(net, content, args); optim.set_style_targets(net, [style], args)
for m in losses: m.mode = "loss"
eng = engine.StyleEngine(net, losses)
g0_h = eng.feval(init.cuda())[2].cpu().clone(); g1_h = eng.feval(x1.cuda())[2].cpu().clone()
y64 = (g1_64 - g0_64).flatten(); y32 = (g1_32 - g0_32).flatten().double(); yh = (g1_h - g0_h).flatten().double()
print("t", t, "|y|/|g|", float(y64.norm() / g0_64.norm()))
print("g0: cpu32-f64", rel_l2(g0_32, g0_64), " hip-f64", rel_l2(g0_h, g0_64))
print("y : cpu32-f64", rel_l2(y32, y64), " hip-f64", rel_l2(yh, y64))
s = (x1 - init).flatten().double()
for name, y in (("f64", y64), ("cpu32", y32), ("hip", yh)):
    print(name, "ys/yy", float(y.dot(s) / y.dot(y)))
# which part of the gradient carries the y error?  per-term decomposition is not available; look at spatial structure
e = (yh - y64).reshape(3, S, S); e32 = (y32 - y64).reshape(3, S, S)
print("hip y err: interior", float(e[:, 2:-2, 2:-2].norm()), "border", float((e.norm() ** 2 - e[:, 2:-2, 2:-2].norm() ** 2) ** 0.5))
print("cpu y err: interior", float(e32[:, 2:-2, 2:-2].norm()), "border", float((e32.norm() ** 2 - e32[:, 2:-2, 2:-2].norm() ** 2) ** 0.5))
