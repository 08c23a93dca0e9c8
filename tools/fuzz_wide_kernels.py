"""Fuzzing of the wide 3x3 kernels on planes and channel counts the hand-written lists do not hold (tests/test_conv_x3p_gpu.py and friends
fix 14 + 7 + 8 shapes; tests/test_random_shapes_gpu.py draws planes up to 150 pixels): random channel counts 32 ... 512, planes 17 ... 420
pixels a side, one to three images, both paddings, random workgroup limits of the persistent kernel, with and without split workspaces -
and per case the identities the kernels promise:

  * conv_x3p with a random workgroup limit == conv_x3p with all 256 workgroups bit for bit, plain / bias + ReLU / masked, and within 1e-6
    of conv_x3q (its first chunk of an item folds once where conv_x3q folds twice: not the same bits);
  * the pooling form of conv_x3p and of conv_x3q == the plain form + maua_pool2x2_fwd_codes bit for bit (pooled map and decision bytes);
  * the unpooling form of both == maua_pool2x2_bwd_codes + the plain form (of the same kernel) bit for bit;
  * conv_x3q against fp64 (F.conv2d on the CPU) on two random 48 x 48 output crops: 2e-6 rel-L2;  conv_x3w against conv_x3q: 1e-6;
  * (half of the cases) the Gram-carrying backward form [F > 0] (backward-data + D . F) of conv_x3w against fp64 crops and rerun bit for
    bit, conv_x3p's against conv_x3w's: 1e-6.

    python tools/fuzz_wide_kernels.py [cases, default 150] [seed base, default 0]          (prints one line per failing case, then a summary)"""
import math
import os
import random
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "maua-style_amd")]
import hip  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 150
BASE = int(sys.argv[2]) if len(sys.argv) > 2 else 0
hip.lib()


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-300))


def crop_ref(x, w, b, y0, x0, size, pad):
    n, c, h, wd = x.shape
    ys, xs = y0 - pad, x0 - pad
    win = torch.zeros(n, c, size + 2, size + 2, dtype=torch.float64)
    sy0, sx0, sy1, sx1 = max(ys, 0), max(xs, 0), min(ys + size + 2, h), min(xs + size + 2, wd)
    win[:, :, sy0 - ys:sy1 - ys, sx0 - xs:sx1 - xs] = x[:, :, sy0:sy1, sx0:sx1].cpu().double()
    return F.conv2d(win, w.cpu().double(), None if b is None else b.cpu().double())


fails, done = [], 0
for case in range(N):
    r = random.Random(777000 + 1000003 * BASE + case)
    cin = 32 * r.randint(1, 16)
    cout = 64 * r.randint(1, 8)
    H, W = r.randint(17, 420), r.randint(33, 420)
    n = r.choice([1, 1, 1, 2, 3])
    if n * (cin + cout) * H * W * 4 > 3 << 30:
        n = 1
    pad = r.choice([1, 1, 1, 0])
    groups = r.choice([0, 8, 16, 64, 256])
    split = r.random() < 0.5
    tag = f"case {case}: {cin}->{cout} {H}x{W} n={n} pad={pad} groups={groups} split_ws={split}"
    try:
        g = torch.Generator(device="cuda").manual_seed(case + 31 * BASE)
        x = torch.relu(torch.randn(n, cin, H, W, generator=g, device="cuda")) * float(10.0 ** r.uniform(-2, 2))
        w = torch.randn(cout, cin, 3, 3, generator=g, device="cuda") * math.sqrt(2.0 / (9 * cin))
        b = torch.randn(cout, generator=g, device="cuda") * 0.1
        bank_f, bank_b, wsc = hip.conv_pack_filters_x3q(w)
        hip.conv_x3p_set_max_groups(groups)

        def ws_for(fn_bytes, *a):
            return torch.empty(max(int(fn_bytes(*a)), 16), dtype=torch.uint8, device="cuda") if split else torch.empty(16, dtype=torch.uint8, device="cuda")

        wq = ws_for(hip.conv_x3q_workspace_bytes, n, cin, H, W, cout, pad)
        wp = ws_for(hip.conv_x3p_workspace_bytes, n, cin, H, W, cout, pad)
        ok_p = hip.conv_x3p_supported(cin, H, W, cout, pad)
        OH, OW = H + 2 * pad - 2, W + 2 * pad - 2
        mask = torch.randn(n, cout, OH, OW, generator=g, device="cuda")
        for relu, bias, m in ((False, None, None), (True, b, None), (False, None, mask)):
            yq = hip.conv3x3_x3q(x, bank_f, wsc, bias, cout, pad, relu, out_relu_mask=m, workspace=wq)
            if ok_p:
                yp = hip.conv3x3_x3p(x, bank_f, wsc, bias, cout, pad, relu, out_relu_mask=m, workspace=wp)
                hip.conv_x3p_set_max_groups(0)
                yp_all = hip.conv3x3_x3p(x, bank_f, wsc, bias, cout, pad, relu, out_relu_mask=m, workspace=wp)
                hip.conv_x3p_set_max_groups(groups)
                torch.cuda.synchronize()
                if not torch.equal(yp, yp_all):
                    fails.append(tag + f" x3p with {groups} workgroups != x3p with all (relu={relu}, bias={bias is not None}, masked={m is not None})")
                if not rel_l2(yp, yq) <= 1e-6:
                    fails.append(tag + f" x3p vs x3q (relu={relu}, bias={bias is not None}, masked={m is not None}): {rel_l2(yp, yq):.2e}")
        yq = hip.conv3x3_x3q(x, bank_f, wsc, b, cout, pad, False, workspace=wq)
        torch.cuda.synchronize()
        for _ in range(2):
            size = min(48, OH, OW)
            y0, x0 = r.randint(0, OH - size), r.randint(0, OW - size)
            ref = crop_ref(x, w, b, y0, x0, size, pad)
            e = rel_l2(yq[:, :, y0:y0 + size, x0:x0 + size].cpu(), ref)
            if not e <= 2e-6:
                fails.append(tag + f" x3q vs fp64 at ({y0},{x0}): {e:.2e}")
        if hip.conv_x3w_supported(cin, H, W, pad):
            bw_f, _, wscw = hip.conv_pack_filters_x3w(w)
            yw = hip.conv3x3_x3w(x, bw_f, wscw, b, cout, pad, False)
            torch.cuda.synchronize()
            if not rel_l2(yw, yq) <= 1e-6:
                fails.append(tag + f" x3w vs x3q: {rel_l2(yw, yq):.2e}")
        # pooling form (padding 1 layers, planes of 2 x 2 and more, cout % 8 == 0)
        if pad == 1 and H >= 2 and W >= 2:
            for name, plain, fn in (("x3q", lambda: hip.conv3x3_x3q(x, bank_f, wsc, b, cout, 1, True, workspace=wq),
                                     lambda po, co: hip.conv3x3_x3q_relu_pool(x, bank_f, wsc, b, cout, 1, po, co, workspace=wq)),
                                    ("x3p", lambda: hip.conv3x3_x3p(x, bank_f, wsc, b, cout, 1, True, workspace=wp),
                                     (lambda po, co: hip.conv3x3_x3p(x, bank_f, wsc, b, cout, 1, True, out=po, pool_codes=co, workspace=wp)) if ok_p else None)):
                if fn is None:
                    continue
                act = plain()
                pooled2 = torch.empty(n, cout, H // 2, W // 2, device="cuda")
                codes2 = torch.empty(n, cout, H // 2, W // 2, dtype=torch.uint8, device="cuda")
                hip.pool2x2_fwd_codes(act, pooled2, codes2)
                pooled = torch.full_like(pooled2, float("nan"))
                codes = torch.full_like(codes2, 255)
                fn(pooled, codes)
                torch.cuda.synchronize()
                if not (torch.equal(pooled, pooled2) and torch.equal(codes, codes2)):
                    fails.append(tag + f" {name} pooling form != its plain form + pool")
            # unpooling form: the backward-data pass (cout -> cin channels) staged from the pooled gradient and the decision bytes
            gp = torch.randn(n, cout, H // 2, W // 2, generator=g, device="cuda")
            honour = r.random() < 0.5
            full = hip.pool2x2_bwd_codes(gp, codes2, torch.empty(n, cout, H, W, device="cuda"), honour)
            wq2 = ws_for(hip.conv_x3q_workspace_bytes, n, cout, H, W, cin, 1)
            wp2 = ws_for(hip.conv_x3p_workspace_bytes, n, cout, H, W, cin, 1)
            fm = torch.randn(n, cin, H, W, generator=g, device="cuda") if r.random() < 0.5 else None
            if cout % 32 == 0:
                two = hip.conv3x3_x3q(full, bank_b, wsc, None, cin, 1, False, out_relu_mask=fm, workspace=wq2)
                one = hip.conv3x3_x3q_unpool(gp, codes2, honour, bank_b, wsc, cin, 1, out=torch.full((n, cin, H, W), float("nan"), device="cuda"),
                                             out_relu_mask=fm, workspace=wq2)
                torch.cuda.synchronize()
                if not torch.equal(one, two):
                    fails.append(tag + f" x3q unpooling form != pool backward + plain (honour={honour}, masked={fm is not None})")
                if hip.conv_x3p_supported(cout, H, W, cin, 1):
                    onep = hip.conv3x3_x3p(gp, bank_b, wsc, None, cin, 1, False, out=torch.full((n, cin, H, W), float("nan"), device="cuda"),
                                           out_relu_mask=fm, workspace=wp2, in_codes=codes2, honour_relu_bit=honour)
                    twop = hip.conv3x3_x3p(full, bank_b, wsc, None, cin, 1, False, out_relu_mask=fm, workspace=wp2)
                    torch.cuda.synchronize()
                    if not torch.equal(onep, twop):
                        fails.append(tag + f" x3p unpooling form != pool backward + its plain form (honour={honour}, masked={fm is not None})")
        # Gram-carrying backward form (the style layers sit on 64 ... 512-channel maps): out = [F > 0] (backward-data + D . F) in one launch of
        # conv_x3w / conv_x3p against fp64 on two crops, the two kernels against each other, a rerun bit for bit
        if pad == 1 and cin % 16 == 0 and cin <= 512 and r.random() < 0.5:
            c = cin                                              # channels of F = channels the backward pass produces
            gy = torch.randn(n, cout, H, W, generator=g, device="cuda") * (torch.rand(n, cout, H, W, generator=g, device="cuda") > 0.5)
            fmap = torch.relu(torch.randn(n, c, H, W, generator=g, device="cuda"))
            D = torch.randn(c, c, generator=g, device="cuda") * 1e-3
            D = D + D.t()
            bank = hip.conv_x3w_dmat_bank(c, "cuda", n)
            for f in range(n):
                hip.conv_pack_dmat_x3w(D, bank[0][f], bank[1][f:f + 1])
            _, bbw, wscw = hip.conv_pack_filters_x3w(w)
            wsw = ws_for(hip.conv_x3w_workspace_bytes, n, cout, H, W, c, 1)
            yw = hip.conv3x3_x3w_gram(gy, bbw, wscw, fmap, bank[0], bank[1], c, 1, workspace=wsw)
            yw2 = hip.conv3x3_x3w_gram(gy, bbw, wscw, fmap, bank[0], bank[1], c, 1, workspace=wsw)
            torch.cuda.synchronize()
            if not torch.equal(yw, yw2):
                fails.append(tag + " x3w Gram form: rerun differs")
            wb = w.flip(2, 3).transpose(0, 1).contiguous()
            for _ in range(2):
                size = min(48, H, W)
                y0, x0 = r.randint(0, H - size), r.randint(0, W - size)
                fc = fmap[:, :, y0:y0 + size, x0:x0 + size].cpu().double()
                ref = (crop_ref(gy, wb, None, y0, x0, size, 1) + torch.einsum("ij,njhw->nihw", D.cpu().double(), fc)) * (fc > 0)
                e = rel_l2(yw[:, :, y0:y0 + size, x0:x0 + size].cpu(), ref)
                if not e <= 2e-6:
                    fails.append(tag + f" x3w Gram form vs fp64 at ({y0},{x0}): {e:.2e}")
            if cout % 32 == 0 and c % 64 == 0 and hip.conv_x3p_supported(cout, H, W, c, 1):
                yp = hip.conv3x3_x3p(gy, bank_b, wsc, None, c, 1, False, out=torch.full((n, c, H, W), float("nan"), device="cuda"), out_relu_mask=fmap,
                                     dmat_bank=bank[0], dmat_inv_scale=bank[1], workspace=ws_for(hip.conv_x3p_workspace_bytes, n, cout, H, W, c, 1))
                torch.cuda.synchronize()
                if not rel_l2(yp, yw) <= 1e-6:
                    fails.append(tag + f" x3p Gram form vs x3w's: {rel_l2(yp, yw):.2e}")
        done += 1
    except Exception as e:  # noqa: BLE001
        fails.append(tag + f" raised {type(e).__name__}: {str(e)[:200]}")
    finally:
        hip.conv_x3p_set_max_groups(0)
for f in fails:
    print("FAIL", f)
print(f"fuzz_wide_kernels: base {BASE}, {done} of {N} cases ran to the end, {len(fails)} failing checks")
