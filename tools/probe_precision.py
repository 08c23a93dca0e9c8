"""GPU probe: error of the HIP feval against the fp64 golden gradient (how much fp32 noise do our kernels add?)."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd"), os.path.join(REPO, "tests")]
import synth
from conftest import product_args, rel_l2
import tempfile, models, optim, engine

d = tempfile.mkdtemp()
wf = {"vgg19": os.path.join(d, "vgg19_synth.pth"), "nin": os.path.join(d, "nin_synth.pth")}
torch.save(synth.vgg19_state_dict(), wf["vgg19"])
g64 = np.load(os.path.join(REPO, "tests/golden/feval_vgg19_S32_default_f64.npz"))
g32 = np.load(os.path.join(REPO, "tests/golden/feval_vgg19_S32_default.npz"))
args = product_args(wf, S=32)
content, style, init = synth.images(32)
optim.set_model_args(args, 32)
net, losses = models.load_model(args)
optim.set_content_targets(net, content, args); optim.set_style_targets(net, [style], args)
for m in losses: m.mode = "loss"
eng = engine.StyleEngine(net, losses)
slots, total, grad = eng.feval(init.cuda()); torch.cuda.synchronize()
print("hip  vs f64 grad relL2:", rel_l2(grad.cpu(), g64["grad"]))
print("ref32 vs f64 grad relL2:", rel_l2(g32["grad"], g64["grad"]))
print("hip losses rel err vs f64:", (slots.cpu().double().numpy() - g64["loss_values"]) / np.maximum(g64["loss_values"], 1e-9))
print("ref losses rel err vs f64:", (g32["loss_values"] - g64["loss_values"]) / np.maximum(g64["loss_values"], 1e-9))
for k, m in enumerate(net.style_losses):
    print("target", k, rel_l2(m.target.cpu(), g32[f"style_target_{k}"]))
