"""One bf16x6 convolution layer in isolation (for rocprofv3 --pmc passes and quick timing).
    python tools/bench_x6_one.py CIN COUT SIDE [REPS]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip  # noqa: E402

cin, cout, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
x = torch.randn(1, cin, H, H, device="cuda")
w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
if os.environ.get("X6_ZERO") == "1":  # DVFS probe: same instruction stream on all-zero operands (MI355X_MICROARCH.md, DVFS item 1)
    x.zero_()
    w.zero_()
if os.environ.get("X6_ZERO") == "x":
    x.zero_()
use_x3 = os.environ.get("X3") == "1"  # the fp16 three-product kernel instead of bf16x6
if use_x3:
    f3, b3, wsc = hip.conv_pack_filters_x3(w)
    run = lambda: hip.conv3x3_x3(x, f3, wsc, None, cout, 1, True, out=y)
else:
    f6, b6 = hip.conv_pack_filters_x6(w)
    run = lambda: hip.conv3x3_x6(x, f6, None, cout, 1, True, out=y)
y = torch.empty(1, cout, H, H, device="cuda")
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / reps
fl = 2.0 * 9 * cin * cout * H * H
print(f"{'x3' if use_x3 else 'x6'} {cin}->{cout} @{H}: {us:.1f} us  {fl / us / 1e6:.1f} TF algorithmic")
