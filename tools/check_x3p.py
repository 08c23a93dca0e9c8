"""conv_x3p.hip against conv_x3q.hip (same bits expected in one pass), conv_x3w.hip's Gram form (to rounding) and fp64, then timing on
the VGG-19 layer shapes of an S x S image (one process, interleaved rounds).   python tools/check_x3p.py [S] [rounds] [reps] [--notime]"""
import math
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip  # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
S = int(args[0]) if len(args) > 0 else 1024
rounds = int(args[1]) if len(args) > 1 else 5
reps = int(args[2]) if len(args) > 2 else 10
g = torch.Generator(device="cuda").manual_seed(1)
small = torch.empty(16, dtype=torch.uint8, device="cuda")


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def check(cin, cout, H, W, n=1):
    x = torch.relu(torch.randn(n, cin, H, W, device="cuda", generator=g))
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    fq, bq, wsc = hip.conv_pack_filters_x3q(w)
    out = []
    yq = hip.conv3x3_x3q(x, fq, wsc, b, cout, 1, True, workspace=small)
    yp = hip.conv3x3_x3p(x, fq, wsc, b, cout, 1, True, workspace=small)
    ys = hip.conv3x3_x3p(x, fq, wsc, b, cout, 1, True)  # (may split)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    out.append(f"plain eq {bool(torch.equal(yq, yp))} ({rel(yp, yq):.1e}) fp64 {rel(yp, ref):.1e} split[{hip.conv_x3p_split(n, cin, H, W, cout, 1)}] {rel(ys, ref):.1e}")
    # masked backward geometry
    gy = torch.randn(n, cout, H, W, device="cuda", generator=g) * (yq > 0)
    if cout % 32 == 0 and cin % 64 == 0:
        gq = hip.conv3x3_x3q(gy, bq, wsc, None, cin, 1, False, out_relu_mask=x, workspace=small)
        gp = hip.conv3x3_x3p(gy, bq, wsc, None, cin, 1, False, out_relu_mask=x, workspace=small)
        out.append(f"masked eq {bool(torch.equal(gq, gp))} ({rel(gp, gq):.1e})")
    # pool
    if H >= 2 and W >= 2:
        pq = torch.full((n, cout, H // 2, W // 2), float("nan"), device="cuda")
        cq = torch.full((n, cout, H // 2, W // 2), 255, dtype=torch.uint8, device="cuda")
        pp, cp = pq.clone(), cq.clone()
        hip.conv3x3_x3q_relu_pool(x, fq, wsc, b, cout, 1, pq, cq, workspace=small)
        hip.conv3x3_x3p(x, fq, wsc, b, cout, 1, True, out=pp, pool_codes=cp, workspace=small)
        out.append(f"pool eq {bool(torch.equal(pq, pp))} codes {bool(torch.equal(cq, cp))}")
    # unpool: gradient of the pooled map of a cout-channel layer output, produces cin
    if cout % 32 == 0 and cin % 64 == 0 and H % 2 == 0 and W % 2 == 0:
        gpool = torch.randn(n, cout, H // 2, W // 2, device="cuda", generator=g)
        codes = torch.randint(0, 8, (n, cout, H // 2, W // 2), dtype=torch.uint8, device="cuda", generator=g)
        for mask in (None, x):
            uq = hip.conv3x3_x3q_unpool(gpool, codes, True, bq, wsc, cin, 1, out=torch.empty_like(x), out_relu_mask=mask, workspace=small)
            up = hip.conv3x3_x3p(gpool, bq, wsc, None, cin, 1, False, out=torch.empty_like(x), out_relu_mask=mask, workspace=small, in_codes=codes)
            out.append(f"unpool{'+mask' if mask is not None else ''} eq {bool(torch.equal(uq, up))} ({rel(up, uq):.1e})")
    # gram: out = [F > 0] (bwd conv + D F), F = x (cin channels)
    if cout % 32 == 0 and cin % 64 == 0:
        D = torch.randn(cin, cin, device="cuda", generator=g)
        D = (D + D.t()) * 1e-3
        bank = hip.conv_x3w_dmat_bank(cin, "cuda", n)
        for f in range(n):
            hip.conv_pack_dmat_x3w(D, bank[0][f], bank[1][f:f + 1])
        gp = hip.conv3x3_x3p(gy, bq, wsc, None, cin, 1, False, out_relu_mask=x, dmat_bank=bank[0], dmat_inv_scale=bank[1], workspace=small)
        refg = (torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1) + torch.einsum("ij,njhw->nihw", D.double(), x.double())) * (x > 0)
        out.append(f"gram fp64 {rel(gp, refg):.1e}")
    torch.cuda.synchronize()
    print(f"{cin}->{cout} @{H}x{W} n{n}: " + " | ".join(out), flush=True)


for shape in [] if "--nocheck" in sys.argv else [(64, 64, 64, 64, 1), (64, 128, 67, 100, 1), (128, 128, 64, 96, 2), (64, 64, 130, 97, 1), (256, 256, 45, 91, 1), (512, 512, 64, 64, 1),
              (512, 64, 33, 70, 2), (128, 256, 181, 181, 1), (64, 64, 256, 256, 1)]:
    check(*shape)
if "--notime" in sys.argv:
    sys.exit(0)

layers = [("conv1_2", 64, 64, 1), ("conv2_1", 64, 128, 2), ("conv2_2", 128, 128, 2), ("conv3_1", 128, 256, 4),
          ("conv3_2", 256, 256, 4), ("conv4_1", 256, 512, 8), ("conv4_2", 512, 512, 8), ("conv5_1", 512, 512, 16)]
if os.environ.get("LAYERS"):
    layers = [l for l in layers if l[0] in os.environ["LAYERS"].split(",")]
for name, cin, cout, div in layers:
    H = S // div
    x = torch.relu(torch.randn(1, cin, H, H, device="cuda", generator=g))
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    fw, bw, wsw = hip.conv_pack_filters_x3w(w)
    fq, bq, wsq = hip.conv_pack_filters_x3q(w)
    wsz = max(hip.conv_x3q_workspace_bytes(1, cin, H, H, cout, 1), hip.conv_x3w_workspace_bytes(1, cin, H, H, cout, 1), hip.conv_x3p_workspace_bytes(1, cin, H, H, cout, 1), 256)
    wsp = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    y = torch.empty(1, cout, H, H, device="cuda")
    pooled = torch.empty(1, cout, H // 2, H // 2, device="cuda")
    codes = torch.empty(1, cout, H // 2, H // 2, dtype=torch.uint8, device="cuda")
    gy = torch.randn(1, cout, H, H, device="cuda", generator=g)
    gx = torch.empty(1, cin, H, H, device="cuda")
    forms = {
        "plain": {"x3w": lambda: hip.conv3x3_x3w(x, fw, wsw, b, cout, 1, True, out=y, workspace=wsp),
                  "x3q": lambda: hip.conv3x3_x3q(x, fq, wsq, b, cout, 1, True, out=y, workspace=wsp),
                  "x3p": lambda: hip.conv3x3_x3p(x, fq, wsq, b, cout, 1, True, out=y, workspace=wsp)},
        "pool": {"x3w": lambda: hip.conv3x3_x3w_relu_pool(x, fw, wsw, b, cout, 1, pooled, codes, workspace=wsp),
                 "x3q": lambda: hip.conv3x3_x3q_relu_pool(x, fq, wsq, b, cout, 1, pooled, codes, workspace=wsp),
                 "x3p": lambda: hip.conv3x3_x3p(x, fq, wsq, b, cout, 1, True, out=pooled, pool_codes=codes, workspace=wsp)},
        "masked bwd": {"x3w": lambda: hip.conv3x3_x3w(gy, bw, wsw, None, cin, 1, False, out=gx, out_relu_mask=x, workspace=wsp),
                       "x3q": lambda: hip.conv3x3_x3q(gy, bq, wsq, None, cin, 1, False, out=gx, out_relu_mask=x, workspace=wsp),
                       "x3p": lambda: hip.conv3x3_x3p(gy, bq, wsq, None, cin, 1, False, out=gx, out_relu_mask=x, workspace=wsp)},
    }
    if cin <= 256:
        D = torch.randn(cin, cin, device="cuda", generator=g) * 1e-3
        bank = hip.conv_x3w_dmat_bank(cin, "cuda", 1)
        hip.conv_pack_dmat_x3w(D, bank[0][0], bank[1])
        forms["gram bwd"] = {"x3w": lambda: hip.conv3x3_x3w_gram(gy, bw, wsw, x, bank[0], bank[1], cin, 1, out=gx, workspace=wsp),
                             "x3p": lambda: hip.conv3x3_x3p(gy, bq, wsq, None, cin, 1, False, out=gx, out_relu_mask=x, dmat_bank=bank[0], dmat_inv_scale=bank[1], workspace=wsp)}
    fl = 2.0 * 9 * cin * cout * H * H
    for form, runs in forms.items():
        if os.environ.get("FORMS") and form not in os.environ["FORMS"].split(","):
            continue
        times = {k: [] for k in runs}
        for fn in runs.values():
            fn()
        torch.cuda.synchronize()
        for _ in range(rounds):
            for tag, fn in runs.items():
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[tag].append(e0.elapsed_time(e1) * 1e3 / reps)
        med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        print(f"{os.environ.get('TAG', '')}{name} {cin:4d}->{cout:4d} @{H:4d} {form:10s}: " + "  ".join(f"{k} {v:7.1f} us ({fl / v / 1e6:5.0f} TF)" for k, v in med.items()), flush=True)
