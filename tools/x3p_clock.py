"""Where a chunk of conv_x3p spends its cycles: in-kernel shader-clock stamps (diagnostic build, -DXP_STAMP) at the phase boundaries of
every chunk of every wave, first chunks of items (they carry the previous item's epilogue) apart from the others.
    tools/build_stamp_libs.sh      (builds tools/_build/libmaua_pstamp.so)
    python tools/x3p_clock.py CIN COUT SIDE [plain|pool|masked|gram]"""
import ctypes
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
os.environ.setdefault("MAUA_HIP_LIB", os.path.join(REPO, "tools", "_build", "libmaua_pstamp.so"))
import hip  # noqa: E402

cin, cout, H = (int(v) for v in sys.argv[1:4])
form = sys.argv[4] if len(sys.argv) > 4 else "plain"
L = hip.lib()
L.maua_xp_set_stamp_buffer.argtypes = [ctypes.c_void_p]
L.maua_xp_set_stamp_buffer.restype = None
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.relu(torch.randn(1, cin, H, H, device="cuda", generator=g))
w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * math.sqrt(2.0 / (9 * cin))
b = torch.randn(cout, device="cuda", generator=g) * 0.1
fq, bq, wsc = hip.conv_pack_filters_x3q(w)
small = torch.empty(16, dtype=torch.uint8, device="cuda")
y = torch.empty(1, cout, H, H, device="cuda")
fm = torch.relu(torch.randn(1, cout, H, H, device="cuda", generator=g))
if form == "pool":
    pooled = torch.empty(1, cout, H // 2, H // 2, device="cuda")
    codes = torch.empty(1, cout, H // 2, H // 2, dtype=torch.uint8, device="cuda")
    run = lambda: hip.conv3x3_x3p(x, fq, wsc, b, cout, 1, True, out=pooled, pool_codes=codes, workspace=small)
elif form == "masked":
    run = lambda: hip.conv3x3_x3p(x, fq, wsc, None, cout, 1, False, out=y, out_relu_mask=fm, workspace=small)
elif form == "gram":
    D = torch.randn(cout, cout, device="cuda", generator=g) * 1e-3
    bank = hip.conv_x3w_dmat_bank(cout, "cuda", 1)
    hip.conv_pack_dmat_x3w(D, bank[0][0], bank[1])
    run = lambda: hip.conv3x3_x3p(x, fq, wsc, None, cout, 1, False, out=y, out_relu_mask=fm, dmat_bank=bank[0], dmat_inv_scale=bank[1], workspace=small)
else:
    run = lambda: hip.conv3x3_x3p(x, fq, wsc, b, cout, 1, True, out=y, workspace=small)
nch = cin // 32
groups = 256
stamps = torch.zeros(groups * 8 * 64 * 8, dtype=torch.float32, device="cuda")
for _ in range(30):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 100
L.maua_xp_set_stamp_buffer(stamps.data_ptr())
e0.record()
run()
e1.record()
torch.cuda.synchronize()
L.maua_xp_set_stamp_buffer(None)
fl = 2.0 * 9 * cin * cout * H * H
print(f"{form} {cin}->{cout} @{H}: {us:.1f} us unstamped ({fl / us / 1e6:.0f} TF), stamped launch {e0.elapsed_time(e1) * 1e3:.1f} us, {nch} chunks per item")
raw = stamps.view(torch.int32).view(groups, 8, 64, 8).long() & 0xFFFFFFFF
nstamped = int((raw[0, 0, :, 0] != 0).sum())
t = raw[:, :, :min(nstamped, 60)]
names = ["taps 0-3", "tap 4 (vmcnt + max)", "wait + XM", "taps 5-8 (scale, split, DMA)", "lgkm + X1", "patch store + wait + X2"]
M = 1 << 32
for label, sel in (("first chunk of an item (epilogue of the previous one)", [c for c in range(1, t.shape[2] - 1) if c % nch == 0]),
                   ("other chunks", [c for c in range(1, t.shape[2] - 1) if c % nch != 0])):
    if not sel:
        continue
    tt = t[:, :, sel]
    seg = [((tt[..., k + 1] - tt[..., k]) % M).float() for k in range(6)]
    nxt = t[:, :, [c + 1 for c in sel]]
    gap = ((nxt[..., 0] - tt[..., 6]) % M).float()
    tot = sum(float(s.mean()) for s in seg) + float(gap.mean())
    print(f" {label}: {len(sel)} chunks per wave")
    for nme, s in zip(names + ["to the next chunk's first stamp"], seg + [gap]):
        v = s.flatten()
        print(f"  {nme:32s} mean {float(v.mean()):8.0f}  p10 {float(v.kthvalue(max(1, v.numel() // 10)).values):8.0f}  "
              f"p90 {float(v.kthvalue(max(1, v.numel() * 9 // 10)).values):8.0f}   {float(v.mean()) / tot * 100:5.1f} %")
    print(f"  cycles per chunk {tot:.0f}  (2 waves x 432 MFMAs per SIMD: {864 * 16} cycles at 16 per instruction)")
