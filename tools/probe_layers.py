"""GPU probe: per-activation error of the engine's forward pass against the fp64 oracle (x6 on / off)."""
import os, sys, tempfile
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd"), os.path.join(REPO, "tests")]
torch.set_num_threads(4)
import synth
from conftest import product_args, make_cfg, rel_l2
import models, optim, engine
from oracle import OracleNet, build_spec
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
d = tempfile.mkdtemp(); wf = {"vgg19": os.path.join(d, "vgg19_synth.pth"), "nin": os.path.join(d, "n.pth")}
sd = synth.vgg19_state_dict(); torch.save(sd, wf["vgg19"])
content, style, init = synth.images(S)
o64 = OracleNet(build_spec(make_cfg()), sd, torch.float64); o64.capture_content(content); o64.capture_style([style], [1.0])
acts64, _ = o64._forward(init.double())
_, _, g64 = o64.feval(init)
args = product_args(wf, S=S); optim.set_model_args(args, S)
net, losses = models.load_model(args)
optim.set_content_targets(net, content, args); optim.set_style_targets(net, [style], args)
for m in losses: m.mode = "loss"
for x6 in (True, False):
    eng = engine.StyleEngine(net, losses); eng.use_x6 = x6
    _, _, g = eng.feval(init.cuda()); torch.cuda.synchronize()
    print("x6" if x6 else "fp32", "grad err", rel_l2(g.cpu(), g64))
    # engine activation k  <->  oracle spec index of the k-th conv/pool output
    k = 0
    for i, l in enumerate(o64.spec):
        if l.kind in ("relu", "pool"):
            k += 1
            print(f"   act {k:2d} {l.kind:5s} {l.name:8s} err {rel_l2(eng.act[k].cpu(), acts64[i]):.2e}  max {float(acts64[i].abs().max()):.1f}")

# where do the two HIP backward passes part ways?  (gbuf[k] = d loss / d act k, pre-masked)
engs = {}
for x6 in (True, False):
    e = engine.StyleEngine(net, losses); e.use_x6 = x6
    e.feval(init.cuda()); torch.cuda.synchronize()
    engs[x6] = e
for k in sorted(engs[True].gbuf, reverse=True):
    a, b = engs[True].gbuf[k].cpu(), engs[False].gbuf[k].cpu()
    print(f"   g[{k:2d}] x6-vs-fp32 {rel_l2(a, b):.2e}   |g| {float(b.norm()):.3e}  max {float(b.abs().max()):.3e}  nonzero {float((b != 0).float().mean()):.2f}")

print("signed bias of activations vs fp64 (mean of (hip - f64) / mean|f64|) and Gram error per style layer")
for x6 in (True, False):
    e = engs[x6]
    k = 0
    for i, l in enumerate(o64.spec):
        if l.kind in ("relu", "pool"):
            k += 1
            if l.kind == "relu" and l.name in ("relu1_1", "relu1_2", "relu2_1", "relu3_1", "relu4_1", "relu5_1"):
                a, r = e.act[k].cpu().double(), acts64[i]
                bias = float((a - r).mean() / r.abs().mean())
                G = (r.reshape(r.shape[1], -1) @ r.reshape(r.shape[1], -1).t())
                Gh = (a.reshape(r.shape[1], -1) @ a.reshape(r.shape[1], -1).t())
                print(f"   {'x6  ' if x6 else 'fp32'} {l.name}: bias {bias:+.2e}  rms err {rel_l2(a, r):.2e}  gram(fp64 of hip acts) err {rel_l2(Gh, G):.2e}")
