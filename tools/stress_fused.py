"""Run-to-run stability of the fused convolution launches (Gram backward in the K loop, ReLU + pool in the epilogue, pool backward in the
staging, the image layer with its Gram slabs) and of the
L-BFGS kernels: the same launch repeated many times while other work keeps the chip busy must give the same bits every time.
    python tools/stress_fused.py [REPEATS=200]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = torch.Generator(device="cuda").manual_seed(5)
noise_a = torch.randn(4096, 4096, device="cuda", generator=g)
side = torch.cuda.Stream()
bad = 0
for (n, cin, c, H, W) in [(1, 64, 64, 256, 256), (1, 128, 128, 128, 160), (2, 128, 64, 72, 104), (1, 64, 128, 40, 520), (3, 64, 64, 64, 64)]:
    gy = torch.randn(n, cin, H, W, device="cuda", generator=g) * (torch.rand(n, cin, H, W, device="cuda", generator=g) > 0.5)
    w = torch.randn(cin, c, 3, 3, device="cuda", generator=g) * (2.0 / (9 * c)) ** 0.5
    f = torch.relu(torch.randn(n, c, H, W, device="cuda", generator=g))
    _, bb, wsc = hip.conv_pack_filters_x3w(w)
    banks, inv = hip.conv_x3w_dmat_bank(c, "cuda", n)
    for b in range(n):
        d = torch.randn(c, c, device="cuda", generator=g) * 1e-3
        hip.conv_pack_dmat_x3w((d + d.t()).contiguous(), banks[b], inv[b:b + 1])
    ws = torch.empty(max(hip.conv_x3w_workspace_bytes(n, cin, H, W, c, 1), 16), dtype=torch.uint8, device="cuda")
    first = hip.conv3x3_x3w_gram(gy, bb, wsc, f, banks, inv, c, 1, workspace=ws).clone()
    for r in range(reps):
        if r % 3 == 0:
            with torch.cuda.stream(side):
                noise_a @ noise_a  # something else on the chip
        out = hip.conv3x3_x3w_gram(gy, bb, wsc, f, banks, inv, c, 1, workspace=ws)
        if not torch.equal(out, first):
            bad += 1
    # forward + pool
    x = torch.relu(torch.randn(n, cin, H, W, device="cuda", generator=g))
    wf = torch.randn(c, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(c, device="cuda", generator=g) * 0.1
    bf, _, wsc2 = hip.conv_pack_filters_x3w(wf)
    p0 = torch.empty(n, c, H // 2, W // 2, device="cuda")
    c0 = torch.empty(n, c, H // 2, W // 2, dtype=torch.uint8, device="cuda")
    hip.conv3x3_x3w_relu_pool(x, bf, wsc2, bias, c, 1, p0, c0)
    p1, c1 = torch.empty_like(p0), torch.empty_like(c0)
    for r in range(reps):
        hip.conv3x3_x3w_relu_pool(x, bf, wsc2, bias, c, 1, p1, c1)
        if not (torch.equal(p0, p1) and torch.equal(c0, c1)):
            bad += 1
    # backward straight from the pooled gradient (split-K form where the workspace allows it), with the Gram backward along
    gp = torch.randn(n, cin, H // 2, W // 2, device="cuda", generator=g)
    cg = torch.empty(n, cin, H // 2, W // 2, dtype=torch.uint8, device="cuda")
    hip.pool2x2_fwd_codes(torch.relu(torch.randn(n, cin, H, W, device="cuda", generator=g)), torch.empty_like(gp), cg)
    u0 = hip.conv3x3_x3w_unpool(gp, cg, True, bb, wsc, c, 1, out_relu_mask=f, dmat_bank=banks, dmat_inv_scale=inv, workspace=ws).clone()
    for r in range(reps):
        if r % 3 == 0:
            with torch.cuda.stream(side):
                noise_a @ noise_a
        if not torch.equal(hip.conv3x3_x3w_unpool(gp, cg, True, bb, wsc, c, 1, out_relu_mask=f, dmat_bank=banks, dmat_inv_scale=inv, workspace=ws), u0):
            bad += 1
    torch.cuda.synchronize()
    print(f"n={n} {cin}->{c} @{H}x{W}: {reps} repeats each, mismatches so far {bad}", flush=True)
# the image layer with its Gram slabs: activation and slabs (what the finishing kernels read of them) the same every time
img = torch.randn(1, 3, 200, 328, device="cuda", generator=g) * 50
wi = torch.randn(64, 3, 3, 3, device="cuda", generator=g) * 0.3
bank_i = hip.conv_pack_filters_image(wi, torch.randn(64, device="cuda", generator=g))
ns = hip.conv_image_gram_slabs(200, 328, 1)
y0, s0 = torch.empty(1, 64, 200, 328, device="cuda"), torch.zeros(ns, 64, 64, device="cuda")
hip.conv3x3_image_gram(img, bank_i, 1, y0, s0)
y1, s1 = torch.empty_like(y0), torch.zeros_like(s0)
for r in range(reps):
    if r % 3 == 0:
        with torch.cuda.stream(side):
            noise_a @ noise_a
    hip.conv3x3_image_gram(img, bank_i, 1, y1, s1)
    if not (torch.equal(y0, y1) and torch.equal(s0[:, :32], s1[:, :32]) and torch.equal(s0[:, 32:, 32:], s1[:, 32:, 32:])):
        bad += 1
torch.cuda.synchronize()
print(f"image layer + Gram slabs: {reps} repeats, mismatches so far {bad}", flush=True)
# L-BFGS: two states fed the same gradients must stay identical
nvec = 3 * 256 * 256
A = 10.0 ** (torch.rand(nvec, device="cuda", generator=g) * 6 - 3)
x1 = torch.randn(nvec, device="cuda", generator=g)
x2 = x1.clone()
s1, s2 = hip.LbfgsState(nvec, 100, "cuda"), hip.LbfgsState(nvec, 100, "cuda")
for it in range(150):
    s1.iterate(x1, A * x1, 1.0, -1.0, -1.0, None)
    with torch.cuda.stream(side):
        noise_a @ noise_a
    s2.iterate(x2, A * x2, 1.0, -1.0, -1.0, None)
    if it % 10 == 9 and not torch.equal(x1, x2):
        bad += 1
torch.cuda.synchronize()
print("L-BFGS twin states identical:", torch.equal(x1, x2), s1.status())
print("TOTAL MISMATCHES", bad)
sys.exit(1 if bad else 0)
