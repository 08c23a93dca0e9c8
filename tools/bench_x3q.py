"""A/B of conv_x3w.hip (16-channel chunks, 32x32x16) and conv_x3q.hip (32-channel chunks, 16x16x32) on the VGG-19 layer shapes of an
S x S image: results against each other and against fp64 on a crop, then interleaved timing rounds in ONE process (post-ReLU-like
inputs: half the values are zero, as in the network).     python tools/bench_x3q.py [S] [rounds] [reps]"""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
layers = [("conv1_2", 64, 64, 1), ("conv2_1", 64, 128, 2), ("conv2_2", 128, 128, 2), ("conv3_1", 128, 256, 4),
          ("conv3_2", 256, 256, 4), ("conv4_1", 256, 512, 8), ("conv4_2", 512, 512, 8), ("conv5_1", 512, 512, 16)]
only = os.environ.get("LAYERS")
if only:
    layers = [l for l in layers if l[0] in only.split(",")]
mult = {"conv3_2": 3, "conv4_2": 3}
tot = {"x3w": 0.0, "x3q": 0.0}
g = torch.Generator(device="cuda").manual_seed(1)
for name, cin, cout, div in layers:
    H = S // div
    x = torch.relu(torch.randn(1, cin, H, H, device="cuda", generator=g))
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    fw, bw, wsw = hip.conv_pack_filters_x3w(w)
    fq, bq, wsq = hip.conv_pack_filters_x3q(w)
    assert wsq == wsw
    wsz = max(hip.conv_x3q_workspace_bytes(1, cin, H, H, cout, 1), hip.conv_x3w_workspace_bytes(1, cin, H, H, cout, 1), 256)
    wsp = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    yw = torch.empty(1, cout, H, H, device="cuda")
    yq = torch.empty(1, cout, H, H, device="cuda")
    runw = lambda: hip.conv3x3_x3w(x, fw, wsw, b, cout, 1, True, out=yw, workspace=wsp)
    runq = lambda: hip.conv3x3_x3q(x, fq, wsq, b, cout, 1, True, out=yq, workspace=wsp)
    runw()
    runq()
    torch.cuda.synchronize()
    rel = float((yq.double() - yw.double()).norm() / yw.double().norm())
    c = min(48, H - 1)  # (the crop reads one row and column beyond itself)
    win = torch.zeros(1, cin, c + 2, c + 2, dtype=torch.float64)
    win[:, :, 1:, 1:] = x[:, :, :c + 1, :c + 1].cpu().double()
    ref = torch.relu(F.conv2d(win, w.cpu().double(), b.cpu().double()))
    ref32 = torch.relu(F.conv2d(win.float(), w.cpu(), b.cpu())).double()
    e32 = float((ref32 - ref).norm() / ref.norm())
    ew = float((yw[:, :, :c, :c].cpu().double() - ref).norm() / ref.norm())
    eq = float((yq[:, :, :c, :c].cpu().double() - ref).norm() / ref.norm())
    # backward-data geometry too (roles swapped): gradient-like input, masked by the ReLU pattern
    gy = torch.randn(1, cout, H, H, device="cuda", generator=g) * (yw > 0)
    gxw = hip.conv3x3_x3w(gy, bw, wsw, None, cin, 1, False, out_relu_mask=x, workspace=wsp)
    gxq = hip.conv3x3_x3q(gy, bq, wsq, None, cin, 1, False, out_relu_mask=x, workspace=wsp)
    torch.cuda.synchronize()
    relb = float((gxq.double() - gxw.double()).norm() / gxw.double().norm())
    times = {"x3w": [], "x3q": []}
    for _ in range(rounds):
        for tag, fn in (("x3w", runw), ("x3q", runq)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            times[tag].append(e0.elapsed_time(e1) * 1e3 / reps)
    fl = 2.0 * 9 * cin * cout * H * H
    med = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
    for k in tot:
        tot[k] += med[k] * mult.get(name, 1)
    print(f"{os.environ.get('TAG', '')}{name} {cin:4d}->{cout:4d} @{H:4d}: x3w {med['x3w']:7.1f} us ({fl / med['x3w'] / 1e6:6.1f} TF)  x3q {med['x3q']:7.1f} us "
          f"({fl / med['x3q'] / 1e6:6.1f} TF)  ratio {med['x3w'] / med['x3q']:.3f}  split {hip.conv_x3q_split(1, cin, H, H, cout, 1)} | "
          f"x3q vs x3w fwd {rel:.1e} bwd {relb:.1e} | vs fp64 crop: fp32-CPU {e32:.1e} x3w {ew:.1e} x3q {eq:.1e}", flush=True)
print(f"sum over the 12 3x3 layers (fwd geometry): x3w {tot['x3w'] / 1e3:.3f} ms  x3q {tot['x3q'] / 1e3:.3f} ms  ratio {tot['x3w'] / tot['x3q']:.3f}")
