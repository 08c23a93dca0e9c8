"""Randomised-shape check of the kernels added late in round 1 (fp16x3 1x1 product, fp16x3 5x5 convolution, LDS-tiled 3x3/2 max-pool
backward) against fp64 / ATen references: ragged channel counts, odd planes, batches, shifts, masks, accumulation.
    python tools/fuzz_new_kernels.py      (on an MI355X)"""
import sys, os, math, random, torch, importlib
import torch.nn.functional as F
sys.path.insert(0, os.getcwd())
hip = importlib.import_module("maua-style_amd.hip")
random.seed(5)
def rel(a, b): return float((a.double() - b).norm() / (b.norm() + 1e-300))
worst = 0
for it in range(40):
    cin, cout = random.randint(1, 300), random.randint(1, 300)
    hw = random.randint(1, 5000); n = random.randint(1, 3)
    x = torch.randn(n, cin, hw, device="cuda") * 10 ** random.uniform(-3, 3)
    w = torch.randn(cout, cin, device="cuda") / math.sqrt(cin)
    b = torch.randn(cout, device="cuda")
    sh = torch.randn(cin, device="cuda") if it % 2 else None
    base = torch.randn(n, cout, hw, device="cuda"); mask = torch.randn(n, cout, hw, device="cuda")
    xr = x.double() - (sh.double()[None, :, None] if sh is not None else 0)
    ref = torch.relu(torch.einsum("oc,ncp->nop", w.double(), xr) + b.double()[None, :, None] + base.double()) * (mask > 0)
    y = hip.conv1x1_x3(x, w, b, relu=True, out=base.clone(), accumulate=True, out_relu_mask=mask, x_shift=sh)
    e = rel(y, ref); worst = max(worst, e)
    assert e < 5e-6, ("1x1", cin, cout, hw, n, e)
print("conv1x1_x3 fuzz ok, worst", worst)
worst = 0
for it in range(30):
    cin, cout = random.randint(1, 130), random.randint(33, 200)
    H, W = random.randint(5, 70), random.randint(5, 70); pad = random.randint(0, 4); n = random.randint(1, 2)
    if H + 2 * pad < 5 or W + 2 * pad < 5: continue
    x = torch.randn(n, cin, H, W, device="cuda"); w = torch.randn(cout, cin, 5, 5, device="cuda") / math.sqrt(25 * cin)
    b = torch.randn(cout, device="cuda")
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=pad)
    bf, bb, ws = hip.conv_pack_filters_kxk_x3(w)
    y = hip.conv_kxk_x3(x, bf, ws, b, cout, 5, pad, False)
    e = rel(y, ref); worst = max(worst, e)
    assert e < 5e-6, ("5x5 fwd", cin, cout, H, W, pad, e)
    gy = torch.randn_like(y)
    refb = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=pad)
    gx = hip.conv_kxk_x3(gy, bb, ws, None, cin, 5, 4 - pad, False) if cin > 0 else None
    e = rel(gx, refb); worst = max(worst, e)
    assert e < 5e-6, ("5x5 bwd", cin, cout, H, W, pad, e)
print("conv_kxk_x3 fuzz ok, worst", worst)
worst = 0
for it in range(20):
    C = random.randint(1, 400); H, W = random.randint(3, 130), random.randint(3, 130)
    x = torch.relu(torch.randn(2, C, H, W, device="cuda")); 
    oh, ow = hip.pool_out_size(H, 3, 2, True), hip.pool_out_size(W, 3, 2, True)
    gy = torch.randn(2, C, oh, ow, device="cuda")
    xr = x.clone().requires_grad_(True)
    F.max_pool2d(xr.cpu(), 3, 2, 0, ceil_mode=True).backward(gy.cpu())
    gx = hip.pool2d_bwd(gy, x, 3, 2, True, "max")
    ref = None
    xr2 = x.cpu().clone().requires_grad_(True); F.max_pool2d(xr2, 3, 2, 0, ceil_mode=True).backward(gy.cpu())
    e = rel(gx.cpu(), xr2.grad.double()); worst = max(worst, e)
    assert e < 1e-6, ("pool", C, H, W, e)
print("pool3s2 bwd fuzz ok, worst", worst)
