"""Backward-data pass + Gram backward: two launches (conv_x3w, conv1x1_x3 read-modify-write) against the fused one
(conv3x3_x3w_gram) on the VGG-19 style layers of an S x S image.      python tools/bench_fused_gram.py [S] [reps]"""
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
g = torch.Generator(device="cuda").manual_seed(3)


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for name, c, div in (("relu1_1 / conv1_2", 64, 1), ("relu2_1 / conv2_2", 128, 2), ("relu3_1 / conv3_2", 256, 4), ("relu4_1 / conv4_2", 512, 8)):
    H = S // div
    gy = torch.randn(1, c, H, H, device="cuda", generator=g) * (torch.rand(1, c, H, H, device="cuda", generator=g) > 0.5)
    w = torch.randn(c, c, 3, 3, device="cuda", generator=g) * (2.0 / (9 * c)) ** 0.5
    f = torch.relu(torch.randn(1, c, H, H, device="cuda", generator=g))
    d = torch.randn(c, c, device="cuda", generator=g) * 1e-3
    d = (d + d.t()).contiguous()
    _, bb, wsc = hip.conv_pack_filters_x3w(w)
    dbank, dinv = hip.conv_x3w_dmat_bank(c, "cuda")
    hip.conv_pack_dmat_x3w(d, dbank, dinv)
    out = torch.empty(1, c, H, H, device="cuda")
    ws = torch.empty(max(hip.conv_x3w_workspace_bytes(1, c, H, H, c, 1), hip.gram_workspace_bytes(c, H * H), 4096), dtype=torch.uint8, device="cuda")
    t_conv = timed(lambda: hip.conv3x3_x3w(gy, bb, wsc, None, c, 1, False, out=out, workspace=ws))
    t_gram = timed(lambda: hip.gram_bwd(d, f, None, out, True, workspace=ws, relu_mask=f))
    t_fused = timed(lambda: hip.conv3x3_x3w_gram(gy, bb, wsc, f, dbank, dinv, c, 1, out=out, workspace=ws))
    t_pack = timed(lambda: hip.conv_pack_dmat_x3w(d, dbank, dinv))
    t_mask = timed(lambda: hip.conv3x3_x3w(gy, bb, wsc, None, c, 1, False, out=out, out_relu_mask=f, workspace=ws))
    print(f"{name} C = {c:3d} @ {H:4d}: conv {t_conv:6.1f} us + Gram backward {t_gram:6.1f} us = {t_conv + t_gram:6.1f} | fused {t_fused:6.1f} us "
          f"(+ pack {t_pack:4.1f} us) | saves {t_conv + t_gram - t_fused - t_pack:6.1f} us | conv with the mask in its epilogue {t_mask:6.1f} us: "
          f"the chunks of F cost {t_fused - t_mask:5.1f} us")
