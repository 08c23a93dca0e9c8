"""Known-byte-count launches for calibrating FETCH_SIZE / WRITE_SIZE under `rocprofv3 --pmc` (MI355X_MICROARCH.md, HBM:
"calibrate on a known byte count in your own access pattern").  Prints the algorithmic bytes of every launch; compare
with tools/pmc_summary.py's per-kernel figures of the same run."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "maua-style_amd"))
import torch  # noqa: E402

import hip  # noqa: E402

dev = "cuda"
n = 64 * 1024 * 1024
x, y = torch.randn(n, device=dev), torch.randn(n, device=dev)
img = torch.randn(1, 3, 4096, 4096, device=dev)
g = torch.zeros_like(img)
loss = torch.zeros(1, device=dev)
feat = torch.randn(1, 64, 1024, 1024, device=dev)
w = torch.randn(64, 3, 3, 3, device=dev) * 0.05
_, wb = hip.conv_pack_filters(w)
out = {}
for rep in range(3):
    hip.axpy_(y, x, 0.5)                      # y += a x : reads 2n, writes n floats
    hip.fill_(y, 0.0)                         # writes n
    hip.relu_(x)                              # reads n, writes n
    hip.tv_fwd_bwd(img, g, 1e-3, False, loss)  # reads img (+ neighbours from cache), writes g
    pooled = hip.pool2d_fwd(feat, 2, 2, False, "max")   # reads 64 Mi floats, writes 16 Mi
    hip.conv2d_bwd_data(feat, None, wb, w, (1, 3, 1024, 1024), 3, 1, 1)  # few-output-channel kernel: reads 64 Mi floats, writes 3 Mi
torch.cuda.synchronize()
out = {"axpy_kernel": {"read": 2 * n * 4, "write": n * 4}, "fill_kernel": {"read": 0, "write": n * 4},
       "relu_fwd_kernel": {"read": n * 4, "write": n * 4},
       "tv_kernel": {"read": img.numel() * 4, "write": img.numel() * 4},
       "pool2x2_fwd_kernel": {"read": feat.numel() * 4, "write": feat.numel()},
       "conv3x3_few_out_kernel<3, false>": {"read": feat.numel() * 4, "write": 3 * 1024 * 1024 * 4}}
print(json.dumps(out))
