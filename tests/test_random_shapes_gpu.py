"""Seeded random geometries for the four fp16x3 3x3 kernel families against fp64 (round 6).

The shape lists of tests/test_conv_x3{w,q,p}_gpu.py were written by hand around the cases each kernel's author worried about; this sweeps
what nobody thought of: per seed a random family, channel counts, plane (3 ... 150 pixels a side, odd and even, non-square), padding 0 / 1 /
2, batch 1 - 3, a random subset of {bias, ReLU, output mask, accumulation}, with the split-K workspace or forced into one pass, and - for
the kernels that have it - armed for the in-launch finish.  Every result against `F.conv2d` in fp64 (bar 2e-6, the families' own bar), and
the one-pass and split forms against each other to 1e-6 (they differ in summation order only).  Same arithmetic as the layer the reference
runs: `nn.Conv2d(cin, cout, 3, padding=p)` + `nn.ReLU`, `/root/reference/models.py:129-130`; backward-data = the gradient autograd derives."""
import math
import random

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu
BAR = 2e-6


@pytest.fixture(scope="module")
def hip():
    import hip as h
    h.lib()
    yield h
    h.conv_arm_workspace(None)


def draw(seed):
    r = random.Random(1000 + seed)
    family = r.choice(["x3w", "x3q", "x3p", "x3"])
    step = {"x3w": 16, "x3q": 32, "x3p": 32, "x3": 8}[family]
    cin = step * r.randint(1, {"x3w": 12, "x3q": 8, "x3p": 8, "x3": 10}[family])
    cout = 64 * r.randint(1, 4) if family == "x3p" else r.choice([8, 24, 64, 72, 128, 200, 256])
    pad = r.choice([0, 1, 1, 1, 2])
    H, W = r.randint(3, 150), r.randint(3, 150)
    if H + 2 * pad < 3 or W + 2 * pad < 3:
        H, W = H + 3, W + 3
    n = r.choice([1, 1, 2, 3])
    flags = dict(bias=r.random() < 0.6, relu=r.random() < 0.5, mask=r.random() < 0.4,
                 accumulate=family != "x3p" and r.random() < 0.25, armed=family in ("x3w", "x3q") and r.random() < 0.5)
    return family, cin, cout, H, W, n, pad, flags


@pytest.mark.parametrize("seed", range(160))
def test_random_geometry_against_fp64(hip, seed):
    family, cin, cout, H, W, n, pad, fl = draw(seed)
    pack = {"x3w": hip.conv_pack_filters_x3w, "x3q": hip.conv_pack_filters_x3q, "x3p": hip.conv_pack_filters_x3q, "x3": hip.conv_pack_filters_x3}[family]
    conv = {"x3w": hip.conv3x3_x3w, "x3q": hip.conv3x3_x3q, "x3p": hip.conv3x3_x3p, "x3": hip.conv3x3_x3}[family]
    wsb = {"x3w": hip.conv_x3w_workspace_bytes, "x3q": hip.conv_x3q_workspace_bytes, "x3p": hip.conv_x3p_workspace_bytes,
           "x3": hip.conv_x3_workspace_bytes}[family]
    supported = {"x3w": lambda: hip.conv_x3w_supported(cin, H, W, pad), "x3q": lambda: hip.conv_x3q_supported(cin, H, W, pad),
                 "x3p": lambda: hip.conv_x3p_supported(cin, H, W, cout, pad), "x3": lambda: True}[family]()
    if not supported:
        pytest.skip(f"{family} does not take {cin} -> {cout} on {H} x {W}, padding {pad}")
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n, cin, H, W, generator=g)) * float(10.0 ** random.Random(seed).uniform(-3, 3))
    w = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=g) * 0.1 if fl["bias"] else None
    OH, OW = H + 2 * pad - 2, W + 2 * pad - 2
    mask = torch.relu(torch.randn(n, cout, OH, OW, generator=g)) if fl["mask"] else None
    prev = torch.randn(n, cout, OH, OW, generator=g) * float(x.abs().max()) if fl["accumulate"] else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=pad)
    if prev is not None:
        ref = ref + prev.double()
    if fl["relu"]:
        ref = torch.relu(ref)
    if mask is not None:
        ref = ref * (mask > 0)
    bank, _, wsc = pack(w.cuda())
    xd, bd, md = x.cuda(), None if b is None else b.cuda(), None if mask is None else mask.cuda()
    kw = dict(out_relu_mask=md)
    if family != "x3p":
        kw["accumulate"] = fl["accumulate"]
    need = wsb(n, cin, H, W, cout, pad)
    ws = torch.empty(max(need, 16), dtype=torch.uint8, device="cuda")
    ws[:ws.numel() // 4 * 4].view(torch.float32)[:] = float("nan")
    outs = {}
    for form in ("planned", "one_pass"):
        out = prev.cuda().clone() if prev is not None else torch.full((n, cout, OH, OW), float("nan"), device="cuda")
        hip.conv_arm_workspace(ws if (fl["armed"] and form == "planned") else None)
        conv(xd, bank, wsc, bd, cout, pad, fl["relu"], out=out, workspace=ws if form == "planned" else torch.empty(16, dtype=torch.uint8, device="cuda"), **kw)
        torch.cuda.synchronize()
        outs[form] = out.cpu()
    hip.conv_arm_workspace(None)
    scale = float(ref.abs().max()) or 1.0
    for form, out in outs.items():
        assert torch.isfinite(out).all(), (family, form)
        err = float((out.double() - ref).norm() / (ref.norm() if float(ref.norm()) > 0 else 1.0))
        assert err <= BAR or float((out.double() - ref).abs().max()) <= 1e-6 * scale, (family, cin, cout, H, W, n, pad, fl, form, err)
    assert rel_l2(outs["planned"], outs["one_pass"].double()) <= 1e-6 or float((outs["planned"] - outs["one_pass"]).abs().max()) <= 1e-6 * scale


def draw_pool(seed):
    r = random.Random(5000 + seed)
    family = r.choice(["x3w", "x3q", "x3p"])
    step = 16 if family == "x3w" else 32
    cin = step * r.randint(1, 8)
    cout = 64 * r.randint(1, 4) if family == "x3p" else 8 * r.choice([1, 3, 8, 9, 16, 25, 32])
    H, W = r.randint(2, 140), r.randint(2, 140)
    n = r.choice([1, 1, 2])
    return family, cin, cout, H, W, n, r.random() < 0.5, r.random() < 0.5, r.random() < 0.5


@pytest.mark.parametrize("seed", range(96))
def test_random_geometry_of_the_pooling_and_unpooling_forms(hip, seed):
    """conv + ReLU + 2x2 max pool in the epilogue (floor mode: odd planes lose their last row / column) and the backward pass staged from the
    pooled map's gradient and the decision bytes, on random planes of 2 ... 140 pixels a side: the pooled map against
    `max_pool2d(relu(conv))` in fp64, the decision bytes by rebuilding the full gradient from them (`maua_pool2x2_bwd_codes`, exact
    routing) and comparing the fused backward launch with the fp64 convolution of that gradient.  Reference: `nn.MaxPool2d(2, 2)`,
    `/root/reference/models.py:120`, behind `models.py:129-130`."""
    family, cin, cout, H, W, n, honour, masked, armed = draw_pool(seed)
    pack = hip.conv_pack_filters_x3w if family == "x3w" else hip.conv_pack_filters_x3q
    wsb = {"x3w": hip.conv_x3w_workspace_bytes, "x3q": hip.conv_x3q_workspace_bytes, "x3p": hip.conv_x3p_workspace_bytes}[family]
    ok = {"x3w": lambda c, h, w_: hip.conv_x3w_supported(c, h, w_, 1), "x3q": lambda c, h, w_: hip.conv_x3q_supported(c, h, w_, 1),
          "x3p": lambda c, h, w_: True}[family]
    if not ok(cin, H, W) or (family == "x3p" and not hip.conv_x3p_supported(cin, H, W, cout, 1)):
        pytest.skip("unsupported geometry")
    g = torch.Generator().manual_seed(7000 + seed)
    x = torch.relu(torch.randn(n, cin, H, W, generator=g))
    w = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=g) * 0.1
    bf, bb, wsc = pack(w.cuda())
    PH, PW = H // 2, W // 2
    ws = torch.empty(max(wsb(n, cin, H, W, cout, 1), wsb(n, cout, H, W, cin, 1), 16), dtype=torch.uint8, device="cuda")
    ws[:ws.numel() // 4 * 4].view(torch.float32)[:] = float("nan")
    if armed and family != "x3p":
        hip.conv_arm_workspace(ws)
    pooled = torch.full((n, cout, PH, PW), float("nan"), device="cuda")
    codes = torch.full((n * cout * PH * PW,), 255, dtype=torch.uint8, device="cuda")
    if family == "x3w":
        hip.conv3x3_x3w_relu_pool(x.cuda(), bf, wsc, b.cuda(), cout, 1, pooled, codes, workspace=ws)
    elif family == "x3q":
        hip.conv3x3_x3q_relu_pool(x.cuda(), bf, wsc, b.cuda(), cout, 1, pooled, codes, workspace=ws)
    else:
        hip.conv3x3_x3p(x.cuda(), bf, wsc, b.cuda(), cout, 1, True, out=pooled, pool_codes=codes, workspace=ws)
    torch.cuda.synchronize()
    ref = F.max_pool2d(torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)), 2, 2)
    assert torch.isfinite(pooled).all()
    assert rel_l2(pooled.cpu(), ref) <= BAR, (family, cin, cout, H, W, n)
    assert int(codes.max()) <= 7
    # backward: the fused launch against the fp64 convolution of the gradient the decision bytes route
    supported_b = {"x3w": lambda: hip.conv_x3w_supported(cout, H, W, 1), "x3q": lambda: hip.conv_x3q_supported(cout, H, W, 1),
                   "x3p": lambda: cout % 32 == 0 and hip.conv_x3p_supported(cout, H, W, cin, 1)}[family]()
    if not supported_b or cin % 64 and family == "x3p":
        hip.conv_arm_workspace(None)
        return
    gp = torch.randn(n, cout, PH, PW, generator=g).cuda()
    full = hip.pool2x2_bwd_codes(gp, codes, torch.empty(n, cout, H, W, device="cuda"), honour)
    gx = torch.full((n, cin, H, W), float("nan"), device="cuda")
    m = x.cuda() if masked else None
    if family == "x3w":
        hip.conv3x3_x3w_unpool(gp, codes, honour, bb, wsc, cin, 1, out=gx, out_relu_mask=m, workspace=ws)
    elif family == "x3q":
        hip.conv3x3_x3q_unpool(gp, codes, honour, bb, wsc, cin, 1, out=gx, out_relu_mask=m, workspace=ws)
    else:
        hip.conv3x3_x3p(gp, bb, wsc, None, cin, 1, False, out=gx, out_relu_mask=m, in_codes=codes, honour_relu_bit=honour, workspace=ws)
    torch.cuda.synchronize()
    hip.conv_arm_workspace(None)
    refb = torch.nn.grad.conv2d_input(x.shape, w.double(), full.cpu().double(), padding=1)
    if masked:
        refb = refb * (x > 0)
    assert torch.isfinite(gx).all()
    assert rel_l2(gx.cpu(), refb) <= BAR or float((gx.cpu().double() - refb).abs().max()) <= 1e-6 * float(refb.abs().max() or 1.0), (family, cin, cout, H, W, n, honour, masked)
