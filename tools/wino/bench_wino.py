"""A/B of conv_x3w.hip (direct, fp16x3) and conv_wino.hip (Winograd F(2x2,3x3), fp16x3) on the VGG-19 layer shapes of an S x S
image: errors against fp64 on a crop (next to the fp32 CPU convolution's), then interleaved timing rounds in ONE process
(post-ReLU-like inputs: half the values are zero, as in the network).     python tools/bench_wino.py [S] [rounds] [reps]"""
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wino  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
only = sys.argv[4].split(",") if len(sys.argv) > 4 else None
layers = [("conv1_2", 64, 64, 1), ("conv2_1", 64, 128, 2), ("conv2_2", 128, 128, 2), ("conv3_1", 128, 256, 4),
          ("conv3_2", 256, 256, 4), ("conv4_1", 256, 512, 8), ("conv4_2", 512, 512, 8), ("conv5_1", 512, 512, 16)]
mult = {"conv3_2": 3, "conv4_2": 3}
tot = {"x3w": 0.0, "wino": 0.0}
g = torch.Generator(device="cuda").manual_seed(1)
for name, cin, cout, div in layers:
    if only and name not in only:
        continue
    H = S // div
    x = torch.relu(torch.randn(1, cin, H, H, device="cuda", generator=g))
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    fw, bw, wsw = hip.conv_pack_filters_x3w(w)
    fn_, bn_ = wino.conv_pack_filters_wino(w)
    wsz = max(hip.conv_x3w_workspace_bytes(1, cin, H, H, cout, 1), 256)
    wsp = torch.empty(wsz, dtype=torch.uint8, device="cuda")
    yw = torch.empty(1, cout, H, H, device="cuda")
    yn = torch.empty(1, cout, H, H, device="cuda")
    runw = lambda: hip.conv3x3_x3w(x, fw, wsw, b, cout, 1, True, out=yw, workspace=wsp)
    runn = lambda: wino.conv3x3_wino(x, fn_, b, cout, 1, True, out=yn)
    runw()
    runn()
    torch.cuda.synchronize()
    c = min(48, H)
    win = torch.zeros(1, cin, c + 2, c + 2, dtype=torch.float64)
    win[:, :, 1:, 1:] = x[:, :, :c + 1, :c + 1].cpu().double()
    ref = torch.relu(F.conv2d(win, w.cpu().double(), b.cpu().double()))
    r32 = torch.relu(F.conv2d(win.float(), w.cpu(), b.cpu()))
    e32 = float((r32.double() - ref).norm() / ref.norm())
    ew = float((yw[:, :, :c, :c].cpu().double() - ref).norm() / ref.norm())
    en = float((yn[:, :, :c, :c].cpu().double() - ref).norm() / ref.norm())
    gy = torch.randn(1, cout, H, H, device="cuda", generator=g) * (yw > 0)
    gxw = hip.conv3x3_x3w(gy, bw, wsw, None, cin, 1, False, workspace=wsp)
    gxn = wino.conv3x3_wino(gy, bn_, None, cin, 1, False)
    torch.cuda.synchronize()
    relb = float((gxw.double() - gxn.double()).norm() / gxw.double().norm())
    times = {"x3w": [], "wino": []}
    for _ in range(rounds):
        for tag, fn in (("x3w", runw), ("wino", runn)):
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
    print(f"{name} {cin:4d}->{cout:4d} @{H:4d}: x3w {med['x3w']:7.1f} us ({fl / med['x3w'] / 1e6:6.1f} TF)  wino {med['wino']:7.1f} us "
          f"({fl / med['wino'] / 1e6:6.1f} TF)  ratio {med['x3w'] / med['wino']:.3f} | bwd wino vs x3w {relb:.1e} | "
          f"vs fp64 crop: fp32-CPU {e32:.1e} x3w {ew:.1e} wino {en:.1e}", flush=True)
print(f"sum over the layers (fwd geometry): x3w {tot['x3w'] / 1e3:.3f} ms  wino {tot['wino'] / 1e3:.3f} ms  ratio {tot['x3w'] / tot['wino']:.3f}")
