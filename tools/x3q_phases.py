"""Where a WORKGROUP of conv_x3q spends its time on a layer shape: prologue (entry -> K loop), K loop, epilogue (loop exit -> stores
done), and what its CU does between two workgroups (diagnostic build -DXQ_STAMP: tools/build_stamp_libs.sh).
    python tools/x3q_phases.py CIN COUT SIDE [plain|pool|unpool|masked]"""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
os.environ.setdefault("MAUA_HIP_LIB", os.path.join(REPO, "tools", "_build", "libmaua_qstamp.so"))
import hip  # noqa: E402

cin, cout, H = (int(v) for v in sys.argv[1:4])
form = sys.argv[4] if len(sys.argv) > 4 else "plain"
L = hip.lib()
L.maua_xq_set_stamp_buffer.argtypes = [ctypes.c_void_p]
L.maua_xq_set_stamp_buffer.restype = None
g = torch.Generator(device="cuda").manual_seed(3)
x = torch.relu(torch.randn(1, cin, H, H, device="cuda", generator=g))
w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
b = torch.randn(cout, device="cuda", generator=g) * 0.1
fq, bq, wsc = hip.conv_pack_filters_x3q(w)
y = torch.empty(1, cout, H, H, device="cuda")
if form == "pool":
    pooled = torch.empty(1, cout, H // 2, H // 2, device="cuda")
    codes = torch.empty(1, cout, H // 2, H // 2, dtype=torch.uint8, device="cuda")
    run = lambda: hip.conv3x3_x3q_relu_pool(x, fq, wsc, b, cout, 1, pooled, codes)
elif form == "unpool":  # backward of a cout -> cin layer's conv + ReLU + pool group: consumes cout pooled channels of H/2, produces cin @ H
    gp = torch.randn(1, cin, H // 2, H // 2, device="cuda", generator=g)
    codes = torch.randint(0, 8, (1, cin, H // 2, H // 2), dtype=torch.uint8, device="cuda", generator=g)
    fm = torch.relu(torch.randn(1, cout, H, H, device="cuda", generator=g))
    run = lambda: hip.conv3x3_x3q_unpool(gp, codes, True, fq, wsc, cout, 1, out=y, out_relu_mask=fm)
elif form == "masked":
    fm = torch.relu(torch.randn(1, cout, H, H, device="cuda", generator=g))
    run = lambda: hip.conv3x3_x3q(x, fq, wsc, None, cout, 1, False, out=y, out_relu_mask=fm)
else:
    run = lambda: hip.conv3x3_x3q(x, fq, wsc, b, cout, 1, True, out=y)
tiles = ((H + 31) // 32) * ((H + 15) // 16)
gx = ((tiles + 7) // 8) * 8
ncot = (cout + 63) // 64
nch = cin // 32
stamps = torch.zeros(ncot * gx * 8 * 64 * 8, dtype=torch.float32, device="cuda")
for _ in range(30):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    run()
e1.record()
torch.cuda.synchronize()
plain_us = e0.elapsed_time(e1) * 100
L.maua_xq_set_stamp_buffer(stamps.data_ptr())
e0.record()
run()
e1.record()
torch.cuda.synchronize()
L.maua_xq_set_stamp_buffer(None)
raw = stamps.view(torch.int32).view(ncot * gx, 8, 64, 8).long() & 0xFFFFFFFF
s63, s62 = raw[:, :, 63], raw[:, :, 62]
valid = s63[:, 0, 1] != 0
s63, s62 = s63[valid], s62[valid]
M = 1 << 32
t_entry, t_loop0, t_loop1, t_exit = s63[..., 5], s63[..., 1], s63[..., 3], s63[..., 6]
r_entry, r_loop0, r_loop1 = s63[..., 7], s63[..., 2], s63[..., 4]
pro, loop, epi = (t_loop0 - t_entry) % M, (t_loop1 - t_loop0) % M, (t_exit - t_loop1) % M
clk = (loop.float() / ((r_loop1 - r_loop0) % M).float() * 0.1)  # GHz
fl = 2.0 * 9 * cin * cout * H * H
print(f"{form} {cin}->{cout} @{H}: {plain_us:.1f} us unstamped ({fl / plain_us / 1e6:.0f} TF), stamped launch {e0.elapsed_time(e1) * 1e3:.1f} us; "
      f"{int(valid.sum())} workgroups of {nch} chunks, in-loop clock median {float(clk.median()):.3f} GHz")
def stat(name, v):
    v = v.float().flatten()
    q = lambda f: float(v.kthvalue(max(1, int(v.numel() * f))).values)
    print(f"  {name:34s} mean {float(v.mean()):8.0f}  p10 {q(0.1):8.0f}  p50 {q(0.5):8.0f}  p90 {q(0.9):8.0f} cycles")
    return float(v.mean())
# per workgroup: the LAST wave's view (max over waves of the exit, min of the entry)
a = stat("prologue (entry -> K loop), per wave", pro)
b_ = stat("K loop, per wave", loop)
c = stat("epilogue (loop exit -> stores done)", epi)
print(f"  shares of a wave's life: prologue {a / (a + b_ + c) * 100:.1f} %  loop {b_ / (a + b_ + c) * 100:.1f} %  epilogue {c / (a + b_ + c) * 100:.1f} %")
# CU timeline: workgroups keyed by (XCC, SE, SH, CU); realtime of entry and (estimated) exit
hw = s63[:, 0, 0]
xcc = s62[:, 0, 0] & 0xF
cu = (xcc << 16) | (hw & 0xFF00)  # CU_ID [11:8], SH_ID [12], SE_ID [15:13]
r_in = r_entry.min(dim=1).values
exit_r = (r_loop1 + ((t_exit - t_loop1) % M).float() / (clk * 10.0)).max(dim=1).values  # 100 MHz ticks
gaps, lives = [], []
for key in cu.unique():
    idx = (cu == key).nonzero().flatten()
    order = idx[r_in[idx].argsort()]
    for i0, i1 in zip(order[:-1], order[1:]):
        gaps.append(float((r_in[i1] - exit_r[i0]) % M) * 10.0)  # ns
    lives += [float((exit_r[i] - r_in[i]) % M) * 10.0 for i in order]
if gaps:
    gt = torch.tensor(gaps)
    gt = torch.where(gt > 2e9, gt - M * 10.0, gt)
    print(f"  {int(cu.unique().numel())} CUs seen; workgroup life mean {sum(lives) / len(lives) / 1e3:.2f} us; gap between a workgroup's exit and the next one's entry on its CU: "
          f"mean {float(gt.mean()) / 1e3:.2f} us, p10 {float(gt.kthvalue(max(1, len(gaps) // 10)).values) / 1e3:.2f}, p90 {float(gt.kthvalue(max(1, len(gaps) * 9 // 10)).values) / 1e3:.2f}")
span = float(((exit_r.max() - r_in.min()) % M)) * 10.0
print(f"  first entry -> last exit: {span / 1e3:.1f} us")
