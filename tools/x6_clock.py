"""In-kernel shader clock of the bf16x6 convolution's main loop (MI355X_MICROARCH.md, DVFS item 6): a diagnostic build of
conv_x6.hip with s_memtime / s_memrealtime stamps around the K loop, run back to back for >= 2 s on random data.

    python tools/x6_clock.py CIN COUT SIDE          (builds tools/_build/libx6_stamp.so with -DMAUA_X6_STAMP)
Prints the median clock, the loop's cycles per chunk and the MFMA pipe occupancy those cycles imply."""
import ctypes
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "maua-style_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "_build")
os.makedirs(OUT, exist_ok=True)
so = os.path.join(OUT, "libx6_stamp.so")
srcs = [os.path.join(CSRC, f) for f in ("conv_x6.hip", "common.hip")] if os.path.exists(os.path.join(CSRC, "common.hip")) else None
if srcs is None:
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".hip")]
if not os.path.exists(so) or "--rebuild" in sys.argv:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DMAUA_X6_STAMP",
                           "-I", os.path.join(ROOT, "include"), "-o", so] + srcs)
if "--build-only" in sys.argv:
    sys.exit(0)
L = ctypes.CDLL(so)
cin, cout, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(1, cin, H, H, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
L.maua_conv_x6_bank_bytes.restype = ctypes.c_size_t
bank = torch.empty(L.maua_conv_x6_bank_bytes(cout, cin), dtype=torch.uint8, device="cuda")
vp = ctypes.c_void_p
assert L.maua_conv_pack_filters_x6(vp(w.data_ptr()), vp(bank.data_ptr()), None, cout, cin, None) == 0
y = torch.empty(1, cout, H, H, device="cuda")
tiles = ((H + 31) // 32) * ((H + 3) // 4)
cot = (cout + 63) // 64
per_xcd = (tiles + 7) // 8
g8 = min(per_xcd, max(1, -(-1024 // (cot * 8))))  # the launcher's persistent grid (conv_x6_launch)
if os.environ.get("MAUA_X6_PERSIST", "0") == "0":
    g8 = per_xcd
nwg = g8 * 8 * cot
stamps = torch.zeros(nwg * 4 * 2 * 10, dtype=torch.int64, device="cuda")


def launch():
    rc = L.maua_conv3x3_x6(vp(x.data_ptr()), vp(bank.data_ptr()), None, None, vp(y.data_ptr()), 1, cin, H, H, cout, 1, 1, 0,
                           vp(stamps.data_ptr()), ctypes.c_size_t(stamps.numel() * 8), None)
    assert rc == 0


launch()
torch.cuda.synchronize()
t0 = time.time()
n = 0
while time.time() - t0 < 2.5:
    for _ in range(50):
        launch()
    torch.cuda.synchronize()
    n += 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    launch()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 20
allp = stamps.view(10, -1, 2).cpu()  # planes of nwg*4 (wave) entries; the kernel indexes them with ITS grid size
s = allp[0]
live = s[:, 1] > 0
r0, r1 = allp[1][:, 0][live].double(), allp[2][:, 0][live].double()
s = s[live]
clk = (s[:, 0].double() / s[:, 1].double() * 0.1)  # GHz
cyc = s[:, 0].double()
nchunks = int(allp[3][:, 1][live].double().median())  # chunks this wave went through (all its tiles)
mfma_cycles = nchunks * 60 * 32
print(f"x6 {cin}->{cout} @{H}: {us:.1f} us/launch after {n} launches; waves stamped {len(s)}")
print(f"  shader clock (median over waves) {clk.median():.3f} GHz  [p10 {clk.quantile(0.1):.3f}, p90 {clk.quantile(0.9):.3f}]")
print(f"  K-loop cycles per wave: median {cyc.median():.0f}  = {cyc.median() / nchunks:.0f} per 8-channel chunk; "
      f"own MFMA issue cycles {mfma_cycles} -> x4 waves/SIMD = {4 * mfma_cycles / cyc.median() * 100:.1f} % of the loop")
print(f"  loop wall time {cyc.median() / clk.median() / 1e3:.1f} us of the {us:.1f} us launch")
print(f"  K-loop cycles per wave: p1 {cyc.quantile(0.01):.0f} p50 {cyc.median():.0f} p99 {cyc.quantile(0.99):.0f} max {cyc.max():.0f}")
t_first, t_last = r0.min(), r1.max()
print(f"  last launch, 100 MHz clock: loop starts spread over {(r0.max() - t_first) / 100:.1f} us, loop ends spread over {(t_last - r1.min()) / 100:.1f} us, "
      f"first start -> last end {(t_last - t_first) / 100:.1f} us; median start {(r0.median() - t_first) / 100:.1f} us, median end {(r1.median() - t_first) / 100:.1f} us")
enter, issued, done = allp[1][:, 1][live].double(), allp[2][:, 1][live].double(), allp[3][:, 0][live].double()
print(f"  per wave (median, us): entry -> loop start {((r0 - enter).median()) / 100:.2f}; loop {((r1 - r0).median()) / 100:.2f}; "
      f"loop end -> stores issued {((issued - r1).median()) / 100:.2f}; stores issued -> landed {((done - issued).median()) / 100:.2f}; "
      f"residency {((done - enter).median()) / 100:.2f}")
names = ["wait B landed (vmcnt) before X1", "barrier X1", "lgkmcnt + barrier X2", "store_patch + DMA issue", "wait A landed (vmcnt)", "barrier X3"]
tot = cyc.median()
acc = 0.0
for i, nm in enumerate(names):
    v = allp[4 + i][:, 0][live].double().median()
    acc += v
    print(f"  in-loop {nm}: median {v:.0f} cycles = {v / tot * 100:.1f} % of the loop ({v / nchunks:.0f} per chunk)")
print(f"  sum of the stamped waits {acc / tot * 100:.1f} % of the loop")
