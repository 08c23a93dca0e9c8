"""Where a chunk of conv_wino spends its cycles: in-kernel shader-clock stamps (diagnostic build, -DWN_STAMP) at the phase
boundaries of every chunk of every wave.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DWN_STAMP -o tools/_build/libmaua_wnstamp.so maua-style_amd/csrc/*.hip
    python tools/wino_clock.py CIN COUT SIDE"""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
os.environ.setdefault("MAUA_HIP_LIB", os.path.join(REPO, "tools", "_build", "libmaua_wnstamp.so"))
import hip  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import wino  # noqa: E402

cin, cout, H = (int(v) for v in sys.argv[1:4])
L = hip.lib()
L.maua_wn_set_stamp_buffer.argtypes = [ctypes.c_void_p]
L.maua_wn_set_stamp_buffer.restype = None
x = torch.relu(torch.randn(1, cin, H, H, device="cuda"))
w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
fw, bw = wino.conv_pack_filters_wino(w)
y = torch.empty(1, cout, H, H, device="cuda")
tiles = ((H + 31) // 32) * ((H + 7) // 8)
ncot = (cout + 63) // 64
nch = cin // 16
blocks = ncot * tiles if ncot % 8 == 0 else (8 * ((tiles + 8 // ncot - 1) // (8 // ncot)) if ncot in (1, 2, 4) else ncot * tiles)
stamps = torch.zeros(blocks * 8 * 64 * 8, dtype=torch.float32, device="cuda")
for _ in range(20):  # warm: clocks settle under load
    wino.conv3x3_wino(x, fw, None, cout, 1, True, out=y)
torch.cuda.synchronize()
L.maua_wn_set_stamp_buffer(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
wino.conv3x3_wino(x, fw, None, cout, 1, True, out=y)
e1.record()
torch.cuda.synchronize()
L.maua_wn_set_stamp_buffer(None)
print(f"{cin}->{cout} @{H}: stamped launch {e0.elapsed_time(e1) * 1e3:.1f} us, {nch} chunks per workgroup, {blocks} workgroups")
raw = stamps.view(torch.int32).view(blocks, 8, 64, 8).long() & 0xFFFFFFFF
n = min(nch, 62)
t = raw[:, :, :n]
valid = t[:, 0, 0, 0] != 0
t = t[valid]
names = ["barrier wait", "operand of tile block 0", "12 MFMA + 4 folds", "20 LDS-DMA rows (raw patch, chunk + 2)", "operand of tile block 1",
         "12 MFMA + 4 folds", "filter fragments of the next chunk"]
for grp, sel in (("waves 0-3", slice(0, 4)), ("waves 4-7", slice(4, 8))):
    tt = t[:, sel]
    seg = [((tt[..., k + 1] - tt[..., k]) & 0xFFFFFFFF)[:, :, 1:n - 1].float() for k in range(7)]
    tot = sum(s.mean() for s in seg)
    print(f" {grp}:")
    for nme, s in zip(names, seg):
        print(f"  {nme:40s} mean {float(s.mean()):8.0f}  p10 {float(s.flatten().kthvalue(max(1, s.numel() // 10)).values):8.0f}  "
              f"p90 {float(s.flatten().kthvalue(max(1, s.numel() * 9 // 10)).values):8.0f}   {float(s.mean() / tot) * 100:5.1f} %")
    print(f"  cycles per chunk {float(tot):.0f}  (MFMA issue alone: {24 * 32} per wave, two waves per SIMD)")
clk = raw[:, :, 63, 1:5][valid].float()
dc, dr = (clk[..., 2] - clk[..., 0]) % 2**32, (clk[..., 3] - clk[..., 1]) % 2**32
print(f"  in-kernel clock over the K loop: median {float((dc / dr * 0.1).median()):.3f} GHz; K loop {float(dc.median()):.0f} cycles = "
      f"{float((dr * 0.01).median()):.1f} us per workgroup")
