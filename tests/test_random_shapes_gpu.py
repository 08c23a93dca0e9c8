"""Seeded random geometries for the four fp16x3 3x3 kernel families against fp64 (round 6).

The shape lists of tests/test_conv_x3{w,q,p}_gpu.py were written by hand around the cases each kernel's author worried about; this sweeps
what nobody thought of: per seed a random family, channel counts, plane (3 ... 150 pixels a side, odd and even, non-square), padding 0 / 1 /
2, batch 1 - 3, a random subset of {bias, ReLU, output mask, accumulation}, with the split-K workspace or forced into one pass, and - for
the kernels that have it - armed for the in-launch finish.  Every result against `F.conv2d` in fp64 (bar 2e-6, the families' own bar), and
the one-pass and split forms against each other to 1e-6 (they differ in summation order only).  Same arithmetic as the layer the reference
runs: `nn.Conv2d(cin, cout, 3, padding=p)` + `nn.ReLU`, `/root/reference/models.py:129-130`; backward-data = the gradient autograd derives."""
import math
import os
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


# A fuzzing campaign draws other cases from the same generators: RANDOM_SHAPES_BASE=k python -m pytest tests/test_random_shapes_gpu.py -m gpu
# (k = 0: the committed cases; profiles/fuzz_r06_random_shapes.txt: k = 1 ... 6 on the final tree of round 6).
BASE = 1000003 * int(os.environ.get("RANDOM_SHAPES_BASE", "0"))
# ... and RANDOM_CONFIG_SCALE=k multiplies the image side of the random CONFIGURATIONS (k = 3: VGG-19 at 120 ... 600 pixels, where the wide
# kernels, the pooling epilogues and the fused Gram backward are routed; NIN at 450 ... 1260)
SCALE = float(os.environ.get("RANDOM_CONFIG_SCALE", "1"))


def draw(seed):
    r = random.Random(1000 + seed + BASE)
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
    g = torch.Generator().manual_seed(seed + BASE)
    x = torch.relu(torch.randn(n, cin, H, W, generator=g)) * float(10.0 ** random.Random(seed + BASE).uniform(-3, 3))
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
    r = random.Random(5000 + seed + BASE)
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
    g = torch.Generator().manual_seed(7000 + seed + BASE)
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


def draw_config(seed):
    r = random.Random(9000 + seed + BASE)
    nin = seed >= 32   # (the last sixteen configurations: the alternative backbone, reference models.py:74-113)
    S = r.randint(150, 420) if nin else r.randint(40, 200)   # (NIN's stem divides the image by four; three ceil-mode pools follow)
    S = int(S * SCALE)
    relus = [f"relu{i}" for i in range(1, 13)] if nin else \
        ["relu1_1", "relu1_2", "relu2_1", "relu2_2", "relu3_1", "relu3_2", "relu3_3", "relu3_4", "relu4_1", "relu4_2", "relu4_3", "relu4_4", "relu5_1"]
    style = sorted(r.sample(relus, r.randint(1, 5)), key=relus.index)
    content = sorted(r.sample(relus, r.choice([1, 1, 1, 2])), key=relus.index)
    extra = ["--style_layers", ",".join(style), "--content_layers", ",".join(content),
             "--style_weight", str(r.choice([1e2, 1e3, 5e1])), "--content_weight", str(r.choice([5.0, 1.0, 20.0])),
             "--tv_weight", str(r.choice([0.0, 1e-3, 1e-1]))]
    if r.random() < 0.3:
        extra += ["--pooling", "avg"]
    if r.random() < 0.4:
        extra += ["--no_grad_norm"]
    if r.random() < 0.25:
        # (with the default temporal module present the reference divides by the size of its EMPTY target: optim.py:176-178 raises
        #  ZeroDivisionError, and so does the oracle - the flag is only usable without a temporal loss)
        extra += ["--normalize_weights", "--temporal_weight", "0"]
    if r.random() < 0.25:
        extra += ["--use_covariance"]
    two_styles = r.random() < 0.3
    if two_styles:
        extra += ["--style_blend_weights", "0.3,0.7"]
    return S, extra, two_styles, nin


@pytest.mark.parametrize("seed", range(48))
def test_random_configurations_against_the_cpu_oracle(weight_files, seed):
    """The whole function evaluation - forward, per-module losses, hand-derived backward pass to the pixels - on RANDOM configurations
    against the CPU oracle in fp64 (oracle/style_oracle.py: the restatement of reference optim.py:201-238 pinned to the reference's own
    fixtures): image sides 40 ... 200 (odd planes and floor-mode pools included), one to five style layers and one or two content layers
    anywhere between relu1_1 and relu5_1 (fused and unfused Gram backward, content losses on and off the fused-pool groups, networks cut at
    any depth), max / avg pooling, --no_grad_norm, --normalize_weights, --use_covariance, two style images with blend weights, TV on and
    off.  The goldens cover twelve configurations the reference was run on; this covers what nobody listed.  Bars: every module's loss
    1e-4, pixel gradient 1e-4 rel-L2 (the bars of test_engine_feval_matches_reference)."""
    import models
    import optim
    import synth
    from conftest import product_args
    from oracle import OracleNet, build_spec
    from oracle.style_oracle import loss_order
    S, extra, two_styles, nin = draw_config(seed)
    styles = ("s.png", "t.png") if two_styles else ("s.png",)
    args = product_args(weight_files, extra, model="nin" if nin else "vgg19", S=S, N=3, styles=styles)
    content, style, init = synth.images(S)
    style_images = [style] + ([synth.images(S, seed=77)[1]] if two_styles else [])
    optim.set_model_args(args, S)
    net, losses = models.load_model(args)
    optim.set_content_targets(net, content, args)
    optim.set_style_targets(net, style_images, args)
    if args.normalize_weights:
        for mod in net.content_losses + net.style_losses:
            mod.strength = mod.strength / max(mod.target.size())
    for m in losses:
        m.mode = "loss"
    opt = optim.PixelOptimizer(net, losses, init, args)
    slots, total, grad = opt.feval()
    torch.cuda.synchronize()
    slots, grad = slots.clone().cpu(), grad.clone().cpu()
    sd = synth.nin_state_dict() if nin else synth.vgg19_state_dict()
    onet = OracleNet(build_spec(args), sd, dtype=torch.float64)
    onet.capture_content(content)
    onet.capture_style(style_images, args.style_blend_weights)
    if args.normalize_weights:
        onet.normalize_weights()
    ototal, olosses, ograd = onet.feval(init)
    order = loss_order(onet.spec)
    got = slots.tolist()
    for k, i in enumerate(order):
        want = float(olosses.get(i, 0.0))
        assert abs(got[k] - want) <= 1e-4 * max(abs(want), 1e-9), (seed, S, extra, onet.spec[i].name, got[k], want)
    diff = grad.double() - ograd.double()
    rel = float(diff.norm() / ograd.double().norm())
    # A ReLU decision within rounding of its boundary can fall on the other side than in fp64 ("decision-bound calls":
    # at S = 151 with avg pooling ONE relu2_1 activation of 720 000 is 0 here and 2.7e-5 - of typically 60 - in fp64, and its gradient of
    # typical size then moves the image gradient by 1.6e-3 rel-L2, all of it inside the activation's receptive field).  Such a flip is not
    # an error of the arithmetic: up to three 40 x 40 windows around the largest differences are set aside, the rest must meet the bar.
    flips = 0
    if rel > 1e-4 and opt.engine is not None:
        # ... and such flips can be SEEN: the engine's saved activations against the oracle's (they agree to 1e-7 .. 4e-7 everywhere; an
        # element that is positive on one side only is a flipped decision).  A deep flip's receptive field is a tenth of the image (NIN,
        # 322 px: two flips, image gradient 1.5e-3 off with 79 % of the squared error in 1 % of the pixels): no window rule holds
        # there, so with flips in sight the bar is a sanity bound and the strict one applies to what decisions cannot explain.
        acts, _ = onet._forward(init.double())
        for k, v in opt.engine.act.items():
            if k == 0 or v is None or v.is_meta:
                continue
            ev = v.cpu().double()
            cands = [oa for oa in acts if tuple(oa.shape) == tuple(ev.shape)]
            oa = min(cands, key=lambda t: float((ev - t).norm()))
            assert float((ev - oa).norm() / oa.norm()) <= 2e-6, (seed, k)
            flips += int(((ev > 0) ^ (oa > 0)).sum())
        # ... and so can the other kind of decision, the arg-max of a max-pooling window whose two largest entries lie within rounding of
        # each other (found by the fuzzing campaign of round 6, RANDOM_SHAPES_BASE = 1 ... 6: six of 288 configurations, five of them NIN
        # with its overlapping 3 x 3 windows, 1.6e-4 ... 9.2e-4 off with every activation equal to 1e-7): the engine's decision - from the
        # activation it pooled, or from the decision bytes where the convolution's epilogue pooled - against the oracle's indices, counted
        # where the window's maximum is positive (a maximum of zero passes no gradient).
        acts, aux = onet._forward(init.double())
        epools = [st for st in opt.engine.steps if st.kind == "pool"]
        opools = [i for i, l in enumerate(onet.spec) if l.kind == "pool"]
        assert len(epools) == len(opools)
        for st, i in zip(epools, opools):
            l = onet.spec[i]
            if l.pool_mode != "max":
                continue
            src = opt.engine.act[st.src]
            if not src.is_meta:
                eidx = F.max_pool2d(src.cpu(), l.k, l.stride, 0, ceil_mode=l.ceil, return_indices=True)[1]
            else:   # fused into the producing launch: byte = corner (2 dy + dx) | 4 where the maximum is not positive
                from conftest import planar_codes
                codes = planar_codes(opt.engine.pool_codes[id(st)].cpu()).long()   # ([image][c / 8][pooled pixel][c % 8] in memory)
                hp, wp = codes.shape[2:]
                w_in = acts[i - 1].shape[3]
                py = torch.arange(hp).view(1, 1, hp, 1)
                px = torch.arange(wp).view(1, 1, 1, wp)
                eidx = (2 * py + ((codes & 3) >> 1)) * w_in + 2 * px + (codes & 1)
            flips += int(((eidx != aux[i]) & (acts[i] > 0)).sum())
    if flips:
        assert rel <= 2e-2 * flips, (seed, S, extra, rel, flips)
        return
    windows = 0
    while rel > 1e-4 and windows < 3:
        amp = diff.abs().sum(dim=(0, 1))
        y, x = divmod(int(amp.argmax()), amp.shape[1])
        diff[:, :, max(0, y - 20):y + 20, max(0, x - 20):x + 20] = 0
        rel = float(diff.norm() / ograd.double().norm())
        windows += 1
    assert rel <= 1e-4, (seed, S, extra, rel, windows, opt.engine is not None)


@pytest.mark.parametrize("seed", range(20))
def test_random_lbfgs_problems_against_the_oracle(hip, seed):
    """The device-resident L-BFGS (lbfgs.hip: coefficient-space two-loop, ring of `history` pairs, the y.s > 1e-10 rule, H_diag, the first
    step min(1, 1 / |g|_1)) on random vector lengths (ragged: not multiples of the kernels' block sizes), histories 1 ... 60 (wrapping several
    times within the run) and iteration counts, against the oracle's restatement of torch.optim.LBFGS in fp64 (reference optim.py:180-191)."""
    from oracle import lbfgs_run
    r = random.Random(3000 + seed + BASE)
    n = r.choice([r.randint(50, 5000), r.randint(5000, 60000), 3 * r.randint(20, 120) ** 2])
    history = r.choice([1, 2, 3, 5, 8, 17, 33, 60])
    iters = r.randint(6, 45)
    gg = torch.Generator().manual_seed(seed + BASE)
    a = torch.rand(n, generator=gg, dtype=torch.float64) * 3 + 0.5
    q = 0.1 * n * n
    sigma = 0.1

    def fg(x):
        aa = a.to(x.device, x.dtype)
        rr = torch.roll(x, 1)
        loss = sigma * ((aa * x * x).sum() + q * (x ** 4).sum() + 0.5 * (x * rr).sum())
        grad = sigma * (2 * aa * x + 4 * q * x ** 3 + 0.5 * (rr + torch.roll(x, -1)))
        return float(loss), grad
    x0 = (torch.randn(n, generator=gg, dtype=torch.float64) * (1.25 / n)).float().double()
    trace, stats = [], {}
    ref, _ = lbfgs_run(fg, x0, iters, history=history, trace=trace, stats=stats)
    x = x0.float().cuda()
    st = hip.LbfgsState(n, history, x.device)
    for it in range(iters):
        st.iterate(x, fg(x)[1].contiguous())
        if it in (0, 1, 2, 4):
            torch.cuda.synchronize()
            assert rel_l2(x.cpu(), trace[it]) <= 2e-4, (n, history, iters, it)
    torch.cuda.synchronize()
    s = st.status()
    assert s["n_iter"] == iters and not s["stopped"] and s["history_len"] <= history and abs(s["history_len"] - stats["history_len"]) <= 2
    assert float((x.cpu().double() - ref).norm() / x0.norm()) <= 1e-4, (n, history, iters)


@pytest.mark.parametrize("seed", range(40))
def test_random_gram_shapes_against_fp64(hip, seed):
    """GramMatrix forward (`torch.mm(x, x.t())`, reference loss.py:67-91; covariance form :87-89) and its backward D (F - mean) on random
    channel counts (8 ... 600, ragged against the 64- and 128-channel blocks) and plane sizes (ragged against the 64-pixel stages), with and
    without centring, masked and accumulating: both against fp64."""
    r = random.Random(4000 + seed + BASE)
    c = r.choice([8, 24, 64, 96, 128, 200, 256, 384, 512, 600])
    h, w = r.randint(3, 120), r.randint(3, 120)
    center = r.random() < 0.4
    g = torch.Generator().manual_seed(seed + BASE)
    f = torch.relu(torch.randn(1, c, h, w, generator=g)) * float(10.0 ** r.uniform(-2, 2))
    fd = f.cuda()
    n = f.numel()
    gram, mean = hip.gram_fwd(fd, 1.0 / n, center)
    torch.cuda.synchronize()
    F2 = f.double().reshape(c, -1)
    Fc = F2 - F2.mean(dim=1, keepdim=True) if center else F2
    ref = Fc @ Fc.t() / n
    assert rel_l2(gram.cpu(), ref) <= 2e-6, (c, h, w, center)
    assert torch.equal(gram, gram.t())
    if center:
        assert rel_l2(mean.cpu(), F2.mean(dim=1)) <= 1e-6
    d = torch.randn(c, c, generator=g) * 1e-3
    d = (d + d.t()).cuda()
    masked, acc = r.random() < 0.5, r.random() < 0.5
    prev = torch.randn(c, h * w, generator=g).cuda() * float(f.abs().max()) * 1e-3
    gf = prev.clone()
    hip.gram_bwd(d, fd, mean if center else None, gf, acc, relu_mask=fd if masked else None)
    torch.cuda.synchronize()
    want = d.cpu().double() @ Fc
    if acc:
        want = want + prev.cpu().double()
    if masked:
        want = want * (F2 > 0)
    assert rel_l2(gf.cpu(), want) <= 2e-6 or float((gf.cpu().double() - want).abs().max()) <= 1e-6 * float(want.abs().max() or 1.0), (c, h, w, center, masked, acc)


@pytest.mark.parametrize("seed", range(40))
def test_random_pool_geometries_against_torch(hip, seed):
    """`nn.MaxPool2d` / `nn.AvgPool2d` (reference models.py:119-123: 2x2 stride 2; NIN's 3x3 stride 2 ceil mode, :77-80) forward and backward
    on random planes: values bit for bit against ATen on the CPU (selections and averages of the same fp32 values), gradients routed to
    ATen's receivers; with the ReLU mask of the pooled map's source folded in."""
    r = random.Random(6000 + seed + BASE)
    k, stride, ceil = r.choice([(2, 2, False), (2, 2, False), (3, 2, True), (3, 2, False), (2, 2, True), (3, 3, False)])
    mode = r.choice(["max", "max", "avg"])
    n, c = r.choice([1, 2]), r.choice([3, 8, 24, 64])
    h, w = r.randint(k, 90), r.randint(k, 90)
    g = torch.Generator().manual_seed(seed + BASE)
    x = torch.relu(torch.randn(n, c, h, w, generator=g))
    x[x > 0] = torch.round(x[x > 0] * 8) / 8 + 0.125     # many exact ties: the first maximum in scan order must win, as in ATen
    xd = x.cuda()
    y = hip.pool2d_fwd(xd, k, stride, ceil, mode)
    xr = x.clone().requires_grad_(True)
    yr = (F.max_pool2d(xr, k, stride, 0, ceil_mode=ceil) if mode == "max" else F.avg_pool2d(xr, k, stride, 0, ceil_mode=ceil))
    torch.cuda.synchronize()
    assert y.shape == yr.shape and torch.equal(y.cpu(), yr.detach()), (k, stride, ceil, mode, n, c, h, w)
    gy = torch.randn(*yr.shape, generator=g)
    yr.backward(gy)
    mask = r.random() < 0.5
    gx = hip.pool2d_bwd(gy.cuda(), xd, k, stride, ceil, mode, relu_mask_by_x=mask)
    torch.cuda.synchronize()
    want = xr.grad * (x > 0) if mask else xr.grad
    if mode == "max":
        assert torch.equal(gx.cpu() != 0, want != 0) or k == 3, (k, stride, ceil, mode, h, w)
    assert rel_l2(gx.cpu(), want.double()) <= 1e-6, (k, stride, ceil, mode, n, c, h, w, mask)


@pytest.mark.parametrize("seed", range(40))
def test_random_1x1_5x5_and_image_layer_shapes_against_fp64(hip, seed):
    """The other convolution kernels on random geometries against fp64: the fp16x3 1x1 product (NIN's cccp layers and the Gram backward,
    reference models.py:84-110), the fp16x3 5x5 layer (NIN's conv2, :86), the image layer forward (bf16x6, `nn.Conv2d(3, 64, 3)`, :129) and
    its backward-data pass on the matrix cores (conv_few_mfma.hip)."""
    r = random.Random(8000 + seed + BASE)
    kind = ["1x1", "5x5", "image", "few"][seed % 4]
    g = torch.Generator().manual_seed(seed + BASE)
    n = r.choice([1, 1, 2])
    h, w = r.randint(6, 130), r.randint(6, 130)
    if kind == "1x1":
        cin, cout = r.choice([16, 96, 200, 256, 384, 1024]), r.choice([64, 96, 200, 256, 1024])
        x = torch.relu(torch.randn(n, cin, h, w, generator=g))
        wt = torch.randn(cout, cin, generator=g) * math.sqrt(2.0 / cin)
        b = torch.randn(cout, generator=g) * 0.1
        relu, masked = r.random() < 0.5, r.random() < 0.4
        mask = torch.relu(torch.randn(n, cout, h, w, generator=g)) if masked else None
        y = hip.conv1x1_x3(x.cuda(), wt.cuda(), b.cuda(), relu, out_relu_mask=None if mask is None else mask.cuda())
        ref = F.conv2d(x.double(), wt.double()[:, :, None, None], b.double())
        ref = torch.relu(ref) if relu else ref
        ref = ref * (mask > 0) if masked else ref
    elif kind == "5x5":
        cin, cout, pad = r.choice([8, 96, 128]), r.choice([64, 256, 200]), r.choice([2, 2, 0, 1])
        if h + 2 * pad < 5 or w + 2 * pad < 5:
            h, w = h + 5, w + 5
        x = torch.relu(torch.randn(n, cin, h, w, generator=g))
        wt = torch.randn(cout, cin, 5, 5, generator=g) * math.sqrt(2.0 / (25 * cin))
        b = torch.randn(cout, generator=g) * 0.1
        bf, _, wsc = hip.conv_pack_filters_kxk_x3(wt.cuda())
        y = hip.conv_kxk_x3(x.cuda(), bf, wsc, b.cuda(), cout, 5, pad, True)
        ref = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=pad))
    elif kind == "image":
        cin, cout, pad = r.choice([1, 3, 3]), 64, r.choice([1, 1, 0])
        x = torch.rand(1, cin, h, w, generator=g) * 255 - 120
        wt = torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
        b = torch.randn(cout, generator=g) * 0.1
        bank = hip.conv_pack_filters_image(wt.cuda(), b.cuda())
        y = hip.conv3x3_image(x.cuda(), bank, cout, pad, True)
        ref = torch.relu(F.conv2d(x.double(), wt.double(), b.double(), padding=pad))
    else:
        cin = r.choice([1, 3, 3])
        if not hip.conv_few_mfma_supported(n, cin, h, w, 64, 1):
            pytest.skip("geometry outside conv_few_mfma")
        wt = torch.randn(64, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
        gy = torch.randn(n, 64, h, w, generator=g) * (torch.rand(n, 64, h, w, generator=g) > 0.5)
        bank = hip.conv_pack_filters_few_mfma(wt.cuda())
        y = hip.conv3x3_few_mfma(gy.cuda(), bank, cin, tile=r.choice([0, 1, 2, 3]))
        ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), gy.double(), padding=1)
    torch.cuda.synchronize()
    assert y.shape == ref.shape and torch.isfinite(y).all()
    assert rel_l2(y.cpu(), ref) <= BAR, (kind, n, h, w)


@pytest.mark.parametrize("seed", range(24))
def test_random_bilinear_resizes_against_aten(hip, seed):
    """`F.interpolate(x, ..., mode="bilinear", align_corners=False)` between two scales (reference style.py:38-66) on random sizes, in both
    calling forms (size / scale_factor): ATen's source-index arithmetic, to 1e-6 of the value range."""
    r = random.Random(9500 + seed + BASE)
    c, h, w = r.choice([1, 3]), r.randint(8, 300), r.randint(8, 300)
    x = torch.rand(1, c, h, w, generator=torch.Generator().manual_seed(seed + BASE)) * 255 - 120
    if r.random() < 0.5:
        size = (r.randint(8, 400), r.randint(8, 400))
        y = hip.resize_bilinear(x.cuda(), size=size)
        ref = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    else:
        sf = r.choice([0.5, 2.0, 1.4142135, 0.70710678, 1.3, 3.0])
        y = hip.resize_bilinear(x.cuda(), scale_factor=sf)
        ref = F.interpolate(x, scale_factor=sf, mode="bilinear", align_corners=False)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert float((y.cpu() - ref).abs().max()) <= 1e-6 * 255 * 4, (c, h, w)


@pytest.mark.parametrize("seed", range(10))
def test_random_frame_batches_are_bit_identical_to_single_frames(weight_files, seed):
    """optim.optimize_frames on random batch sizes, image sides (odd ones too) and flags: every frame of the batch carries the bits of the
    same frame optimised alone under the same plan (the split-K policy follows the PLANNED frames, per-frame kernels share nothing):
    frames are independent B = 1 problems (reference style.py:192-290)."""
    import models
    import optim
    import synth
    from conftest import product_args
    r = random.Random(12000 + seed + BASE)
    B, N, S = r.randint(2, 5), r.randint(2, 6), r.randint(48, 140)
    opt = r.choice(["lbfgs", "lbfgs", "adam"])
    extra = []
    if r.random() < 0.3:
        extra += ["--pooling", "avg"]
    if r.random() < 0.3:
        extra += ["--no_grad_norm"]
    if r.random() < 0.3:
        extra += ["--use_covariance"]
    if r.random() < 0.4:
        extra += ["--style_layers", r.choice(["relu1_1,relu3_1", "relu2_1,relu4_1,relu5_1", "relu1_2"]), "--content_layers", r.choice(["relu3_2", "relu4_2", "relu2_2"])]
    style = synth.images(S)[1]
    contents = torch.cat([synth.images(S, seed=50 + k)[0] for k in range(B)])
    inits = torch.cat([synth.images(S, seed=60 + k)[2] for k in range(B)])
    args = product_args(weight_files, extra, optimizer=opt, S=S, N=N)
    optim.set_model_args(args, S)
    net, losses = models.load_model(args)
    together = optim.optimize_frames(contents.cuda(), [style], inits.cuda(), N, args, net, losses).cpu()
    for k in sorted(r.sample(range(B), 2)):
        single = optim.optimize_frames(contents[k:k + 1].cuda(), [style], inits[k:k + 1].cuda(), N, args, net, losses, planned_frames=B).cpu()
        assert torch.equal(together[k], single[0]), (B, N, S, opt, extra, k, rel_l2(together[k], single[0].double()))
    assert torch.isfinite(together).all() and not torch.equal(together[0], together[1])
