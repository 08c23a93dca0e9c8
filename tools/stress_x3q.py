"""Run-to-run stability of conv_x3q.hip: the same launch repeated under changing conditions (alone, right behind another kernel that
leaves other bytes in LDS, with a GEMM running beside it on a second stream) must give the same bits every time.
    python tools/stress_x3q.py [repeats]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator(device="cuda").manual_seed(3)
side = torch.cuda.Stream()
a = torch.randn(4096, 4096, device="cuda", generator=g)
bad = 0
for name, cin, cout, H, W in [("conv4_2@128", 512, 512, 128, 128), ("conv4_2@64 (split-K)", 512, 512, 64, 64), ("conv3_2@128", 256, 256, 128, 128),
                              ("ragged", 256, 200, 90, 91)]:
    x = torch.relu(torch.randn(1, cin, H, W, device="cuda", generator=g))
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device="cuda", generator=g) * 0.1
    fq, bq, ws = hip.conv_pack_filters_x3q(w)
    fw, bw, _ = hip.conv_pack_filters_x3w(w)
    wsp = torch.empty(max(hip.conv_x3q_workspace_bytes(1, cin, H, W, cout, 1), hip.conv_x3w_workspace_bytes(1, cin, H, W, cout, 1), 256), dtype=torch.uint8, device="cuda")
    mask = torch.randn(1, cout, H, W, device="cuda", generator=g)
    ev = H % 2 == 0 and W % 2 == 0
    if ev:
        pooled0 = torch.empty(1, cout, H // 2, W // 2, device="cuda")
        codes0 = torch.empty(1, cout, H // 2, W // 2, dtype=torch.uint8, device="cuda")
        gp = torch.randn(1, cout, H // 2, W // 2, device="cuda", generator=g)
    ref = {}
    for r in range(reps):
        mode = r % 3
        if mode == 1:   # another kernel's bytes in LDS right before
            hip.conv3x3_x3w(x, fw, ws, b, cout, 1, True, workspace=wsp)
        if mode == 2:   # a GEMM beside it
            with torch.cuda.stream(side):
                a @ a
        out = {"fwd": hip.conv3x3_x3q(x, fq, ws, b, cout, 1, True, workspace=wsp).clone(),
               "masked": hip.conv3x3_x3q(x, fq, ws, None, cout, 1, False, out_relu_mask=mask, workspace=wsp).clone()}
        if ev:
            hip.conv3x3_x3q_relu_pool(x, fq, ws, b, cout, 1, pooled0, codes0, workspace=wsp)
            out["pool"], out["codes"] = pooled0.clone(), codes0.clone()
            if cout % 32 == 0:
                out["unpool"] = hip.conv3x3_x3q_unpool(gp, codes0, True, bq, ws, cin, 1, workspace=wsp).clone()
        torch.cuda.synchronize()
        for k, v in out.items():
            if k not in ref:
                ref[k] = v
            elif not torch.equal(ref[k], v):
                bad += 1
                d = (ref[k].float() - v.float()).abs()
                print(f"{name} {k} repeat {r} (mode {mode}): {int((d > 0).sum())} differing values, max {float(d.max()):.3e}", flush=True)
    print(name, "done", flush=True)
print("differing results:", bad)
sys.exit(1 if bad else 0)
