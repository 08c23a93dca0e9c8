"""Where a chunk of conv_x3w spends its cycles: in-kernel shader-clock stamps (diagnostic build, -DXW_STAMP) at the phase
boundaries of every chunk of every wave.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DXW_STAMP -o tools/_build/libmaua_stamp.so maua-style_amd/csrc/*.hip
    python tools/x3w_clock.py CIN COUT SIDE"""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
os.environ.setdefault("MAUA_HIP_LIB", os.path.join(REPO, "tools", "_build", "libmaua_stamp.so"))
import hip  # noqa: E402

cin, cout, H = (int(v) for v in sys.argv[1:4])
L = hip.lib()
L.maua_xw_set_stamp_buffer.argtypes = [ctypes.c_void_p]
L.maua_xw_set_stamp_buffer.restype = None
x = torch.relu(torch.randn(1, cin, H, H, device="cuda"))
w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
fw, bw, wsc = hip.conv_pack_filters_x3w(w)
y = torch.empty(1, cout, H, H, device="cuda")
tiles = ((H + 31) // 32) * ((H + 7) // 8)
gx = ((tiles + 7) // 8) * 8
ncot = (cout + 63) // 64
nch = cin // 16
stamps = torch.zeros(ncot * gx * 4 * 64 * 8, dtype=torch.float32, device="cuda")
for _ in range(20):  # warm: clocks settle under load
    hip.conv3x3_x3w(x, fw, wsc, None, cout, 1, True, out=y)
torch.cuda.synchronize()
L.maua_xw_set_stamp_buffer(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
hip.conv3x3_x3w(x, fw, wsc, None, cout, 1, True, out=y)
e1.record()
torch.cuda.synchronize()
L.maua_xw_set_stamp_buffer(None)
print(f"{cin}->{cout} @{H}: stamped launch {e0.elapsed_time(e1) * 1e3:.1f} us, {nch} chunks per workgroup")
raw = stamps.view(torch.int32).view(ncot * gx, 4, 64, 8).long() & 0xFFFFFFFF
t = raw[:, :, :nch]
valid = t[:, 0, 0, 0] != 0
hwid = raw[:, :, 63, 0][valid]
t = t[valid]
names = ["taps 0-4 (60 MFMA)", "max + lgkm wait", "XM wait", "DMA a, taps 5-8, split", "lgkm + X1 wait", "DMA b, write, fold", "vm/lgkm wait", "X2 wait"]
seg = []
for k in range(7):
    seg.append(((t[..., k + 1] - t[..., k]) & 0xFFFFFFFF)[:, :, :nch - 1].float())
nxt = ((t[:, :, 1:, 0] - t[:, :, :-1, 7]) & 0xFFFFFFFF).float()
seg.append(nxt)
tot = sum(s.mean() for s in seg)
for nme, s in zip(names, seg):
    print(f"  {nme:22s} mean {float(s.mean()):8.0f}  p10 {float(s.flatten().kthvalue(max(1, s.numel() // 10)).values):8.0f}  "
          f"p90 {float(s.flatten().kthvalue(max(1, s.numel() * 9 // 10)).values):8.0f}   {float(s.mean() / tot) * 100:5.1f} %")
print(f"  cycles per chunk {float(tot):.0f}  (MFMA issue alone: {108 * 32} per wave, two waves per SIMD)")
span = ((t[:, :, nch - 1, 6] - t[:, :, 0, 0]) & 0xFFFFFFFF).float()
print(f"  K loop per wave: mean {float(span.mean()):.0f} cycles, min {float(span.min()):.0f}, max {float(span.max()):.0f}; "
      f"workgroups stamped {int(valid.sum())}")

clk = raw[:, :, 63, 1:5][valid].float()
dc, dr = (clk[..., 2] - clk[..., 0]) % 2**32, (clk[..., 3] - clk[..., 1]) % 2**32
print(f"  in-kernel clock over the K loop: median {float((dc / dr * 0.1).median()):.3f} GHz (shader cycles per 100 MHz tick)")
# waves that share a SIMD: how much of one wave's matrix segments [0,1] and [3,4] overlaps its partner's
import collections
xcc_unknown = 0
groups = collections.defaultdict(list)
tc = t.cpu()
hw = hwid.cpu()
for wg in range(tc.shape[0]):
    for wv in range(4):
        h = int(hw[wg, wv])
        key = (h >> 4) & 0x3, (h >> 8) & 0xf, (h >> 12) & 0x1, (h >> 13) & 0x7  # SIMD, CU, SH, SE (XCD unknown: clocks tell them apart)
        groups[key].append((wg, wv))
def segs(wg, wv):
    out = []
    for c in range(nch):
        out.append((int(tc[wg, wv, c, 0]), int(tc[wg, wv, c, 1])))
        out.append((int(tc[wg, wv, c, 3]), int(tc[wg, wv, c, 4])))
    return out
tot_m = tot_ov = 0
for key, members in groups.items():
    # members of one (SIMD, CU, SH, SE) over all 8 XCDs and all rounds: pair those whose lifetimes overlap
    for i in range(len(members)):
        a = segs(*members[i])
        a0, a1 = a[0][0], a[-1][1]
        for j in range(len(members)):
            if i == j:
                continue
            b = segs(*members[j])
            if b[0][0] > a1 or b[-1][1] < a0 or abs(b[0][0] - a0) > 2_000_000:
                continue
            for s0, s1 in a:
                tot_m += (s1 - s0)
                for u0, u1 in b:
                    lo, hi = max(s0, u0), min(s1, u1)
                    if hi > lo:
                        tot_ov += hi - lo
print(f"  matrix-segment cycles overlapped by a same-SIMD partner's matrix segments: {100.0 * tot_ov / max(tot_m, 1):.1f} %")
