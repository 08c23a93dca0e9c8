"""Soak test for silent wrong results at the 1e-6-per-workgroup level (the rate of the first 128 x 128 Gram form's fault, probes_r05.md section 4):
every matrix kernel family that runs two MFMA waves per SIMD is launched thousands of times on one fixed input and every result compared bit for
bit with the first one ON THE DEVICE (no host copies: ~1e7 workgroup launches per family in seconds).
    python tools/soak_kernels.py [launches per family]"""
import math
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g = torch.Generator(device="cuda").manual_seed(7)
small = torch.empty(16, dtype=torch.uint8, device="cuda")


def soak(name, fn, wgs):
    ref = fn().clone()
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    for _ in range(N):
        out = fn()
        bad += (out.view(torch.int32) != ref.view(torch.int32)).any().long()
    torch.cuda.synchronize()
    print(f"{name:44s}: {int(bad)} of {N} launches differ from the first ({N * wgs / 1e6:.1f} M workgroup launches)", flush=True)
    return int(bad)


cin = cout = 512
H = 128
x = torch.relu(torch.randn(1, cin, H, H, device="cuda", generator=g))
w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * math.sqrt(2.0 / (9 * cin))
fw, bw, wsw = hip.conv_pack_filters_x3w(w)
fq, bq, wsq = hip.conv_pack_filters_x3q(w)
y = torch.empty(1, cout, H, H, device="cuda")
total = 0
total += soak("conv_x3w 512->512 @128 (32x32x16, 2 WG/CU)", lambda: hip.conv3x3_x3w(x, fw, wsw, None, cout, 1, True, out=y, workspace=small), 64 * 8 * 8)
total += soak("conv_x3q 512->512 @128 (16x16x32, 8 waves)", lambda: hip.conv3x3_x3q(x, fq, wsq, None, cout, 1, True, out=y, workspace=small), 256)
x2 = torch.relu(torch.randn(1, 128, 512, 512, device="cuda", generator=g))
w2 = torch.randn(128, 128, 3, 3, device="cuda", generator=g) * math.sqrt(2.0 / (9 * 128))
f2w, b2w, ws2 = hip.conv_pack_filters_x3w(w2)
f2q, _, _ = hip.conv_pack_filters_x3q(w2)
y2 = torch.empty(1, 128, 512, 512, device="cuda")
total += soak("conv_x3w 128->128 @512", lambda: hip.conv3x3_x3w(x2, f2w, ws2, None, 128, 1, True, out=y2, workspace=small), 8192)
total += soak("conv_x3p 128->128 @512 (persistent)", lambda: hip.conv3x3_x3p(x2, f2q, ws2, None, 128, 1, True, out=y2, workspace=small), 256)
D = torch.randn(128, 128, device="cuda", generator=g) * 1e-3
bank = hip.conv_x3w_dmat_bank(128, "cuda", 1)
hip.conv_pack_dmat_x3w(D + D.t(), bank[0][0], bank[1])
gy = torch.randn(1, 128, 512, 512, device="cuda", generator=g)
total += soak("conv_x3w + Gram backward 128 @512", lambda: hip.conv3x3_x3w_gram(gy, b2w, ws2, x2, bank[0], bank[1], 128, 1, out=y2, workspace=small), 8192)
f = torch.relu(torch.randn(1, 512, 128, 128, device="cuda", generator=g))
total += soak("gram_fwd 512 x 16384 (planner default)", lambda: hip.gram_fwd(f, 1.0 / f.numel(), False)[0], 136 * 8)
f3 = torch.relu(torch.randn(1, 256, 256, 256, device="cuda", generator=g))
total += soak("gram_fwd 256 x 65536", lambda: hip.gram_fwd(f3, 1.0 / f3.numel(), False)[0], 512)
gf = torch.zeros(512, 128 * 128, device="cuda")
Ds = torch.randn(512, 512, device="cuda", generator=g) * 1e-3
total += soak("gram_bwd 512 x 16384 (conv1x1_x3)", lambda: hip.gram_bwd(Ds, f, None, gf, False, relu_mask=f), 2048)
# round 5: the other families that run two (or more) MFMA-issuing waves per SIMD
gy64 = torch.randn(1, 64, 512, 512, device="cuda", generator=g)
w11 = torch.randn(64, 3, 3, 3, device="cuda", generator=g) * 0.1
fm_bank = hip.conv_pack_filters_few_mfma(w11)
gx3 = torch.empty(1, 3, 512, 512, device="cuda")
total += soak("conv_few_mfma 64->3 @512 (bf16x6, 2 WG/CU)", lambda: hip.conv3x3_few_mfma(gy64, fm_bank, 3, out=gx3), 9 * 128)
f64 = torch.relu(torch.randn(1, 96, 254, 254, device="cuda", generator=g))
total += soak("gram_fwd 96 x 64516 (64 x 64 blocks, 2 WG/CU)", lambda: hip.gram_fwd(f64, 1.0 / f64.numel(), True)[0], 759)
f5 = torch.relu(torch.randn(1, 512, 16, 16, device="cuda", generator=g))
total += soak("gram_fwd 512 x 256 (64 x 64 blocks)", lambda: hip.gram_fwd(f5, 1.0 / f5.numel(), False)[0], 136 * 4)
xs = torch.relu(torch.randn(1, 512, 32, 32, device="cuda", generator=g))
fx3, bx3, wsx3 = hip.conv_pack_filters_x3(w)
ys = torch.empty(1, 512, 32, 32, device="cuda")
wss = torch.empty(hip.conv_x3_workspace_bytes(1, 512, 32, 32, 512, 1), dtype=torch.uint8, device="cuda")
total += soak("conv_x3 512->512 @32 (split-K, 4 WG/CU)", lambda: hip.conv3x3_x3(xs, fx3, wsx3, None, 512, 1, True, out=ys, workspace=wss), 1024)
x5 = torch.relu(torch.randn(1, 96, 127, 127, device="cuda", generator=g))
w5 = torch.randn(256, 96, 5, 5, device="cuda", generator=g) * 0.02
bk5 = hip.conv_pack_filters_kxk_x3(w5)
y5 = torch.empty(1, 256, 127, 127, device="cuda")
ws5 = torch.empty(max(16, hip.conv_kxk_x3_workspace_bytes(1, 96, 127, 127, 256, 5, 2)), dtype=torch.uint8, device="cuda")
total += soak("conv_kxk_x3 5x5 96->256 @127 (2 WG/CU)", lambda: hip.conv_kxk_x3(x5, bk5[0], bk5[2], None, 256, 5, 2, True, out=y5, workspace=ws5), 128 * 4)
ximg = torch.rand(1, 3, 512, 512, device="cuda", generator=g) * 255 - 120
ibank = hip.conv_pack_filters_image(w11, torch.zeros(64, device="cuda"))
yimg = torch.empty(1, 64, 512, 512, device="cuda")
total += soak("conv_image 3->64 @512 (bf16x6, 2 WG/CU)", lambda: hip.conv3x3_image(ximg, ibank, 64, 1, True, out=yimg), 512)
f6, b6 = hip.conv_pack_filters_x6(w2)
ws6 = torch.empty(max(16, hip.conv_x6_workspace_bytes(1, 128, 512, 512, 128, 1)), dtype=torch.uint8, device="cuda")
total += soak("conv_x6 128->128 @512 (bf16x6, 4 WG/CU)", lambda: hip.conv3x3_x6(x2, f6, None, 128, 1, True, out=y2, workspace=ws6), 16384)
print("differing launches in all:", total)
sys.exit(1 if total else 0)
