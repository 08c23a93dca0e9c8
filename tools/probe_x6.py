"""GPU probe: accuracy (vs fp64) and speed of the bf16x6 3x3 convolution next to the fp32-MFMA kernel."""
import os, sys, math
import torch, torch.nn.functional as F
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip

def rel(a, b):
    return float((a.double() - b).norm() / b.norm())

torch.manual_seed(0)
for cin, cout, H, W in ((3, 64, 37, 45), (64, 64, 64, 64), (128, 256, 32, 32), (512, 512, 16, 16), (20, 40, 13, 70)):
    x = torch.randn(1, cin, H, W); w = torch.randn(cout, cin, 3, 3) * math.sqrt(2 / (9 * cin)); b = torch.randn(cout) * 0.1
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    wf, wb = hip.conv_pack_filters(w.cuda()); f6, b6 = hip.conv_pack_filters_x6(w.cuda())
    y32 = hip.conv2d_fwd(x.cuda(), wf, b.cuda(), 3, 1, 1, False)
    y6 = hip.conv3x3_x6(x.cuda(), f6, b.cuda(), cout, 1, False)
    ycpu = F.conv2d(x, w, b, padding=1)
    gy = torch.randn(1, cout, H, W)
    refb = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=1)
    g32 = hip.conv2d_bwd_data(gy.cuda(), None, wb, w.cuda(), x.shape, 3, 1, 1)
    g6 = hip.conv3x3_x6(gy.cuda(), b6, None, cin, 1, False)
    torch.cuda.synchronize()
    print(f"{cin:4d}->{cout:4d} @{H}x{W}: fwd err  mfma32 {rel(y32.cpu(), ref):.2e}  x6 {rel(y6.cpu(), ref):.2e}  cpu32 {rel(ycpu, ref):.2e} | "
          f"bwd err mfma32 {rel(g32.cpu(), refb):.2e}  x6 {rel(g6.cpu(), refb):.2e}")

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
layers = [("conv1_2", 64, 64, 1), ("conv2_1", 64, 128, 2), ("conv2_2", 128, 128, 2), ("conv3_1", 128, 256, 4),
          ("conv3_2", 256, 256, 4), ("conv4_1", 256, 512, 8), ("conv4_2", 512, 512, 8), ("conv5_1", 512, 512, 16)]
mult = {"conv3_2": 3, "conv4_2": 3}
tot = [0.0, 0.0]
for name, cin, cout, div in layers:
    H = S // div
    x = torch.randn(1, cin, H, H, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda"); f6, b6 = hip.conv_pack_filters_x6(w); wf, wb = hip.conv_pack_filters(w)
    y = torch.empty(1, cout, H, H, device="cuda")
    flops = 2 * 9 * cin * cout * H * H
    res = []
    for fn in (lambda: hip.conv2d_fwd(x, wf, b, 3, 1, 1, True, out=y), lambda: hip.conv3x3_x6(x, f6, b, cout, 1, True, out=y)):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): fn()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5)
    tot[0] += res[0] * mult.get(name, 1); tot[1] += res[1] * mult.get(name, 1)
    print(f"{name} {cin:4d}->{cout:4d} @{H:4d}  mfma32 {res[0]*1e3:8.1f} us ({flops/res[0]/1e9:6.1f} TF)   x6 {res[1]*1e3:8.1f} us ({flops/res[1]/1e9:6.1f} TF-equivalent)")
print(f"sum fwd (12 big convs): mfma32 {tot[0]:.3f} ms   x6 {tot[1]:.3f} ms")
