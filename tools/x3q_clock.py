"""Where a chunk of conv_x3q spends its cycles: in-kernel shader-clock stamps (diagnostic build, -DXQ_STAMP) at the phase
boundaries of every chunk of every wave.
    tools/build_stamp_libs.sh      (builds tools/_build/libmaua_qstamp.so)
    python tools/x3q_clock.py CIN COUT SIDE"""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
os.environ.setdefault("MAUA_HIP_LIB", os.path.join(REPO, "tools", "_build", "libmaua_qstamp.so"))
import hip  # noqa: E402

cin, cout, H = (int(v) for v in sys.argv[1:4])
L = hip.lib()
L.maua_xq_set_stamp_buffer.argtypes = [ctypes.c_void_p]
L.maua_xq_set_stamp_buffer.restype = None
x = torch.relu(torch.randn(1, cin, H, H, device="cuda"))
w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
fq, bq, wsc = hip.conv_pack_filters_x3q(w)
y = torch.empty(1, cout, H, H, device="cuda")
tiles = ((H + 31) // 32) * ((H + 15) // 16)
gx = ((tiles + 7) // 8) * 8
ncot = (cout + 63) // 64
nch = cin // 32
stamps = torch.zeros(ncot * gx * 8 * 64 * 8, dtype=torch.float32, device="cuda")
for _ in range(20):  # warm: clocks settle under load
    hip.conv3x3_x3q(x, fq, wsc, None, cout, 1, True, out=y)
torch.cuda.synchronize()
L.maua_xq_set_stamp_buffer(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
hip.conv3x3_x3q(x, fq, wsc, None, cout, 1, True, out=y)
e1.record()
torch.cuda.synchronize()
L.maua_xq_set_stamp_buffer(None)
print(f"{cin}->{cout} @{H}: stamped launch {e0.elapsed_time(e1) * 1e3:.1f} us, {nch} chunks per workgroup")
raw = stamps.view(torch.int32).view(ncot * gx, 8, 64, 8).long() & 0xFFFFFFFF
t = raw[:, :, :nch]
valid = t[:, 0, 0, 0] != 0
t = t[valid]
names = ["taps 0-3 (192 MFMA per wave)", "tap 4 + vmcnt + max", "wait + XM", "scale, taps 5-8, split, DMA", "lgkm + X1", "patch store + wait + X2"]
seg = []
for k in range(6):
    seg.append(((t[..., k + 1] - t[..., k]) & 0xFFFFFFFF)[:, :, :nch - 1].float())
top = ((t[:, :, 1:, 0] - t[:, :, :-1, 6]) & 0xFFFFFFFF).float()
tot = sum(s.mean() for s in seg) + top.mean()
for nme, s in zip(names + ["loop back-edge"], seg + [top]):
    print(f"  {nme:30s} mean {float(s.mean()):8.0f}  p10 {float(s.flatten().kthvalue(max(1, s.numel() // 10)).values):8.0f}  "
          f"p90 {float(s.flatten().kthvalue(max(1, s.numel() * 9 // 10)).values):8.0f}   {float(s.mean() / tot) * 100:5.1f} %")
print(f"  cycles per chunk {float(tot):.0f}  (2 waves x 432 MFMAs per SIMD: {864 * 16} cycles at 16 per instruction)")
span = ((t[:, :, nch - 1, 6] - t[:, :, 0, 0]) & 0xFFFFFFFF).float()
print(f"  K loop per wave: mean {float(span.mean()):.0f} cycles, min {float(span.min()):.0f}, max {float(span.max()):.0f}; "
      f"workgroups stamped {int(valid.sum())}")
clk = raw[:, :, 63, 1:5][valid].float()
dc, dr = (clk[..., 2] - clk[..., 0]) % 2**32, (clk[..., 3] - clk[..., 1]) % 2**32
print(f"  in-kernel clock over the K loop: median {float((dc / dr * 0.1).median()):.3f} GHz (shader cycles per 100 MHz tick)")
print(f"  loop entry -> exit per wave: median {float(dc.median()):.0f} cycles = {float((dr * 10).median()) / 1e3:.1f} us")
