"""The largest planes the shipped scaling table names (config/scaling-img.json: a 4096-pixel row, Adam; the reference's max-sizes.py
probes for the largest size a device holds - on 288 GB that is far beyond 4096): a 4096 x 4096 image is 2^24 pixels per plane, 64 channels
of it are exactly 2^32 bytes - the sizes at which 32-bit byte offsets and buffer-resource ranges wrap.  The oracle cannot run here, so the
checks are the full-size ones of test_fullsize_gpu.py moved to this size: every convolution layer of the REAL forward pass against fp64 on
crops (first rows, interior, last rows - the last channels' last pixels sit just below 2^32 bytes), the four 3x3 kernel families standing
alone in both directions, Gram / pooling / TV / update kernels on 2^24-pixel operands, determinism and the directional derivative of the
whole network."""
import math

import pytest
import torch
import torch.nn.functional as F

import synth
from conftest import product_args, rel_l2
from test_fullsize_gpu import _VGG_CONVS, _crop_reference, _fp16x3_bar, _loss_f64

pytestmark = pytest.mark.gpu
S = 4096


def dev(t):
    return t.cuda().contiguous()


# 4096 x 4096: planes of exactly 2^24 pixels.  4400 x 4600: 20.2 M pixels - past what conv_x3q / conv_x3p / conv_img take (2^24), inside
# conv_x3w's 2^25: the full-resolution layers change kernel family there, the pooled ones do not.
# 5800 x 5800: 33.6 M pixels, past conv_x3w too: conv_x3.hip / conv_x6.hip (64-bit addressing throughout) carry the full-resolution layers, a
# 64-channel map is 8.6 GB and has more than 2^31 values.
@pytest.fixture(scope="module", params=[(S, S), (4400, 4600), (5800, 5800)], ids=["4096x4096", "4400x4600", "5800x5800"])
def big(request, weight_files):
    import engine
    import models
    import optim
    import gc
    H, W = request.param
    gc.collect()
    torch.cuda.empty_cache()   # (earlier modules' cached blocks: a 5800 x 5800 evaluation holds ~160 GB)
    need = 5.0e3 * H * W       # bytes: 4.95 KB per pixel measured at 6896 x 6896 (profiles/probe_r06_big_sizes.txt)
    free = torch.cuda.mem_get_info()[0]
    if free < 1.15 * need:
        pytest.skip(f"{H} x {W} needs ~{need / 2**30:.0f} GiB of device memory, {free / 2**30:.0f} GiB are free")
    args = product_args(weight_files, ["--no_grad_norm"], optimizer="adam", S=max(H, W), N=4)
    content, style, init = synth.images(S, H=H, W=W)
    optim.set_model_args(args, S)
    net, losses = models.load_model(args)
    optim.set_content_targets(net, content, args)
    optim.set_style_targets(net, [style], args)
    for m in losses:
        m.mode = "loss"
    eng = engine.StyleEngine(net, losses)
    yield args, net, losses, eng, init.cuda()
    del eng
    torch.cuda.empty_cache()


def _crops(side, size):
    return sorted({(0, 0), (side // 2 - size // 2, min(side // 2 - size // 4, side - size)), (side - size, side - size), (side - size, 0)})


@pytest.mark.parametrize("layer", _VGG_CONVS)
def test_every_layer_of_the_4096_forward_pass_against_fp64(big, layer):
    """relu(conv(a_in)) (and its 2x2 max pool where the launch pools in its epilogue) of the real 4096 x 4096 forward pass, from the engine's
    own buffers, against fp64 on crops at the four places where an offset that wrapped would show (reference models.py:120,129-130)."""
    _, _, _, eng, x = big
    eng.feval(x)
    torch.cuda.synchronize()
    convs = [s for s in eng.steps if s.kind == "conv"]
    step = convs[_VGG_CONVS.index(layer)]
    mod = step.mod
    a_in = eng.act[step.src]
    if a_in.is_meta:
        pytest.skip("the input of this layer exists as a shape only at this size")
    pooled = int(id(step) in eng.fused_pool and eng.act[step.dst].is_meta)
    out = eng.act[eng.fused_pool[id(step)].dst] if pooled else eng.act[step.dst]
    assert float(out.abs().max()) > 0
    size = 64
    hh, ww = a_in.shape[2:]
    for y0, x0 in sorted({(0, 0), ((hh // 2 - 32) & ~1, (ww // 2 - 16) & ~1), ((hh - size) & ~pooled, (ww - size) & ~pooled), ((hh - size) & ~pooled, 0)}):
        res = {}
        for dt in (torch.float64, torch.float32):
            r = _crop_reference(a_in, mod.weight.detach(), mod.bias.detach(), y0, x0, size, 1, dt)
            r = torch.relu(r) if step.relu else r
            res[dt] = F.max_pool2d(r, 2, 2) if pooled else r
        mine = (out[:, :, y0 // 2:(y0 + size) // 2, x0 // 2:(x0 + size) // 2] if pooled else out[:, :, y0:y0 + size, x0:x0 + size]).cpu()
        floor = rel_l2(res[torch.float32], res[torch.float64])
        err = rel_l2(mine, res[torch.float64])
        assert err <= max(_fp16x3_bar("x3p", mod.in_channels) * floor, 1e-7), (layer, (y0, x0), err, floor)


def test_kernel_families_at_the_limits_of_their_offsets(big):
    """Which family carries the two full-resolution layers: the image kernel and the persistent kernel stop below 2^24-pixel planes (conv_x6 /
    conv_x3w take over at 4096 x 4096), conv_x3q at 2^24, conv_x3w at 2^25 (conv_x3) - each by its `_supported` entry point, never
    by a wrapped offset."""
    from test_fullsize_gpu import _routes_by_layer
    _, _, _, eng, x = big
    by = _routes_by_layer(eng, x)
    px = x.shape[2] * x.shape[3]
    assert by["conv1_1"][0]["kernel"] == "conv_x6", by["conv1_1"][0]
    # (64-channel layers outside conv_x3p's reach are conv_x3w's by the planner's table - conv_x3q starts at 256 channels)
    assert by["conv1_2"][0]["kernel"] == ("conv_x3w" if px <= 1 << 25 else "conv_x3"), by["conv1_2"][0]
    for name, (f, b, step) in by.items():
        for r in (f, b):
            if r["kernel"] == "conv_x3p":
                assert r["produced"] * r["plane"][0] * r["plane"][1] * 4 < 1 << 31, (name, r)


def test_4096_evaluation_is_deterministic_and_its_gradient_is_the_slope_of_its_loss(big):
    _, _, _, eng, x = big
    s0, t0, g0 = eng.feval(x)
    s0, t0, g0 = s0.clone(), t0.clone(), g0.clone()
    s1, t1, g1 = eng.feval(x)
    torch.cuda.synchronize()
    assert torch.equal(g0, g1) and torch.equal(s0, s1) and torch.equal(t0, t1)
    assert torch.isfinite(g0).all() and float(t0) > 0
    # every 512-row band of the gradient is alive (a launch whose stores fell outside a wrapped range leaves bands of zeros behind)
    bands = g0.abs().reshape(1, 3, 8, g0.shape[2] // 8, g0.shape[3]).amax(dim=(1, 3, 4)).flatten()
    assert float(bands.min()) > 0, bands
    v = g0 / g0.norm()
    slope = float((g0.double() * v.double()).sum())
    fd = {}
    # (steps of 2 and 4 along the unit gradient of 50-60 M pixels move a pixel by 3-6e-4: the difference of two losses of ~1e9 that a step
    #  of 1 leaves - 44 - is within reach of the evaluations' own fp32 noise: measured 1.6e-3 off at 4400 x 4600, 4e-5 with a step of 2)
    for eps in (2.0, 4.0):
        fd[eps] = (_loss_f64(eng, x + eps * v) - _loss_f64(eng, x - eps * v)) / (2 * eps)
    rich = (4.0 * fd[2.0] - fd[4.0]) / 3.0
    assert abs(rich - slope) <= 2e-3 * abs(slope), (fd, rich, slope)
    assert abs(fd[2.0] - slope) <= 5e-3 * abs(slope), (fd, slope)


@pytest.mark.parametrize("kernel", ["x3q", "x3p", "x3w", "x3"])
@pytest.mark.parametrize("cin,cout,side", [(64, 64, 4096), (64, 128, 4096), (128, 64, 2896)])
def test_3x3_kernels_alone_on_planes_of_2_to_the_24_pixels(cin, cout, side, kernel):
    """Each split-precision 3x3 family on its own, forward bank and backward-data bank, against fp64 crops; 64 -> 128 channels at 4096 x 4096
    writes 8 GiB - twice what a 32-bit byte offset spans."""
    import hip
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.randn(1, cin, side, side, generator=g, device="cuda")
    w = torch.randn(cout, cin, 3, 3, generator=g, device="cuda") * math.sqrt(2.0 / (9 * cin))
    b = torch.randn(cout, generator=g, device="cuda")
    pack, conv = {"x3w": (hip.conv_pack_filters_x3w, hip.conv3x3_x3w), "x3": (hip.conv_pack_filters_x3, hip.conv3x3_x3),
                  "x3q": (hip.conv_pack_filters_x3q, hip.conv3x3_x3q), "x3p": (hip.conv_pack_filters_x3q, hip.conv3x3_x3p)}[kernel]
    bank_f, bank_b, wsc = pack(w)
    if kernel == "x3p" and not hip.conv_x3p_supported(cin, side, side, cout, 1):
        # (its output offsets are 31-bit: 64 channels of 2^24 pixels are past them, and the entry point says so instead of wrapping)
        with pytest.raises(hip.HipError):
            conv(x, bank_f, wsc, b, cout, 1, False)
        return
    y = conv(x, bank_f, wsc, b, cout, 1, False)
    torch.cuda.synchronize()
    for y0, x0 in _crops(side, 64):
        r64 = _crop_reference(x, w, b, y0, x0, 64, 1, torch.float64)
        r32 = _crop_reference(x, w, b, y0, x0, 64, 1, torch.float32)
        err, floor = rel_l2(y[:, :, y0:y0 + 64, x0:x0 + 64].cpu(), r64), rel_l2(r32, r64)
        assert err <= max(_fp16x3_bar(kernel, cin) * floor, 1e-7), ("fwd", (y0, x0), err, floor)
    if kernel == "x3p" and not hip.conv_x3p_supported(cout, side, side, cin, 1):
        return
    gx = conv(y, bank_b, wsc, None, cin, 1, False)
    torch.cuda.synchronize()
    wb = w.flip(2, 3).transpose(0, 1).contiguous()
    for y0, x0 in _crops(side, 64):
        r64 = _crop_reference(y, wb, None, y0, x0, 64, 1, torch.float64)
        r32 = _crop_reference(y, wb, None, y0, x0, 64, 1, torch.float32)
        err, floor = rel_l2(gx[:, :, y0:y0 + 64, x0:x0 + 64].cpu(), r64), rel_l2(r32, r64)
        assert err <= max(_fp16x3_bar(kernel, cout) * floor, 1e-7), ("bwd", (y0, x0), err, floor)


def test_gram_of_64_planes_of_2_to_the_24_pixels():
    import hip
    f = torch.relu(torch.randn(1, 64, S * S, 1, generator=torch.Generator(device="cuda").manual_seed(6), device="cuda"))
    gram, _ = hip.gram_fwd(f, 1.0, False)
    torch.cuda.synchronize()
    assert torch.equal(gram, gram.t())
    want = (f[0, :, :, 0].double() @ f[0, :, :, 0].double().t()).cpu()
    assert rel_l2(gram.cpu(), want) <= 2e-6


def test_pooling_of_2_to_the_24_pixel_planes_routes_every_gradient_once():
    import hip
    g = torch.Generator(device="cuda").manual_seed(8)
    x = torch.randn(1, 64, S, S, generator=g, device="cuda")
    assert hip.pool2x2_codes_supported(1, 64, S, S)
    y = hip.pool2x2_fwd_codes(x, torch.empty(1, 64, S // 2, S // 2, device="cuda"), codes := torch.empty(1, 64, S // 2, S // 2, dtype=torch.uint8, device="cuda"))
    torch.cuda.synchronize()
    for y0, x0 in ((0, 0), (S // 2 - 64, S // 2 - 64), (S // 2 - 64, 0)):
        assert torch.equal(y[:, :, y0:y0 + 64, x0:x0 + 64], F.max_pool2d(x[:, :, 2 * y0:2 * y0 + 128, 2 * x0:2 * x0 + 128], 2, 2))
    gy = torch.randn(y.shape, generator=g, device="cuda")
    gx = hip.pool2x2_bwd_codes(gy, codes, torch.empty_like(x), True)   # (the ReLU bit: only windows whose maximum is positive receive)
    torch.cuda.synchronize()
    win = gx.view(1, 64, S // 2, 2, S // 2, 2)
    assert int((win != 0).sum(dim=(3, 5)).max()) <= 1                  # at most one receiver per window
    assert torch.equal(win.sum(dim=(3, 5)), gy * (y > 0))              # ... and it receives the window's gradient, unchanged
    # the plain pooling pair (no decision bytes) on the same planes
    xr = torch.relu(x)
    y2 = hip.pool2d_fwd(xr, 2, 2, False, "max")
    gx2 = hip.pool2d_bwd(gy, xr, 2, 2, False, "max")
    torch.cuda.synchronize()
    assert torch.equal(y2, torch.relu(y))
    assert int((gx2 != 0).view(1, 64, S // 2, 2, S // 2, 2).sum(dim=(3, 5)).max()) <= 1
    assert abs(float(gx2.double().sum()) - float(gy.double().sum())) <= 1e-6 * float(gy.double().abs().sum())


def test_both_optimisers_on_50_million_pixels_descend(big):
    """Adam through optim.optimize (N + 1 steps, graph path == eager path bit for bit) and twelve L-BFGS iterations (history sweeps over
    vectors of 3 x 2^24 elements) from the same start: finite losses, descent."""
    import optim
    args, net, losses, eng, x = big
    before = float(eng.feval(x)[1])
    content, style, init = synth.images(S, H=x.shape[2], W=x.shape[3])
    outs = []
    for flag in (True, False):
        args.hip_graph = flag
        outs.append(optim.optimize(content, [style], init.clone(), 3, args, net, losses))
    assert torch.equal(outs[0], outs[1])
    after = float(eng.feval(outs[0].cuda())[1])
    assert math.isfinite(after) and after < before, (before, after)
    args.optimizer = "lbfgs"
    try:
        opt = optim.PixelOptimizer(net, losses, init, args)
        totals = [float(opt.step()[1]) for _ in range(12)]
        torch.cuda.synchronize()
        st = opt.state.status()
        assert all(math.isfinite(t) for t in totals) and st["n_iter"] == 12 and not st["stopped"], (totals, st)
        assert torch.isfinite(opt.x).all()
    finally:
        args.optimizer = "adam"
