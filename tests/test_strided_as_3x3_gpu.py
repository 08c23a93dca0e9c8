"""The strided, unpadded layer of NIN (`nn.Conv2d(3, 96, (11, 11), (4, 4))`, /root/reference/models.py:161) as a stride-1 3x3 convolution
over SITES (round 5): the forward pass = space to depth (maua_space_to_depth: the image's 4 x 4 pixel phases become 48 channels over
ceil(H / 4) x ceil(W / 4) sites) + an unpadded 3x3 convolution 48 -> 96 with the 11 x 11 filter regrouped and zero-padded to 12 x 12
(fp16x3 products, conv_x3w.hip); the backward-data pass = a 3x3 convolution 96 -> 48 (pad 2) over the output's sites with the flipped
regrouped filter (fp16x3 products, conv_x3w.hip) + depth to space (maua_depth_to_space).

Against fp64 `F.conv2d(..., stride=4)` / `torch.nn.grad.conv2d_input` (what autograd runs for the reference): <= 2e-6 rel-L2 (measured
1.5e-7); the regrouping kernels bit-exact against a torch restatement; ragged planes (H, W not multiples of 4, sites beyond the image),
batch 2, and the eligibility rules.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu

BAR = 2e-6


@pytest.fixture(scope="module")
def env():
    import hip as h
    import models as m
    h.lib()
    return h, m


def _stem(m, seed, cin=3, cout=96, k=11, s=4):
    mod = m.Conv2d(cin, cout, (k, k), (s, s), (0, 0)).cuda()
    g = torch.Generator(device="cuda").manual_seed(seed)
    mod.weight.data = torch.randn(cout, cin, k, k, device="cuda", generator=g) * 0.05
    mod.bias.data = torch.randn(cout, device="cuda", generator=g)
    return mod, g


@pytest.mark.parametrize("n,c,r,h,w", [(1, 3, 4, 64, 64), (2, 3, 4, 99, 131), (1, 5, 2, 17, 300), (3, 1, 3, 10, 7)])
def test_space_to_depth_and_back_are_exact(env, n, c, r, h, w):
    hip, _ = env
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.randn(n, c, h, w, device="cuda", generator=g)
    qh, qw = -(-h // r), -(-w // r)
    sites = torch.full((n, r * r * c, qh, qw), float("nan"), device="cuda")
    hip.space_to_depth(x, r, sites)
    xp = F.pad(x, (0, qw * r - w, 0, qh * r - h))
    want = xp.view(n, c, qh, r, qw, r).permute(0, 3, 5, 1, 2, 4).reshape(n, r * r * c, qh, qw)
    assert torch.equal(sites, want)
    fewer = torch.full((n, r * r * c, qh - 1, qw + 2), float("nan"), device="cuda")  # any site grid: truncated rows, columns of zeros
    hip.space_to_depth(x, r, fewer)
    assert torch.equal(fewer[:, :, :, :qw], want[:, :, :qh - 1]) and not fewer[:, :, :, qw:].any()
    # depth to space: the channel order of the backward form is (ry, rx, c) too; plain and accumulating
    back = torch.full((n, c, h, w), float("nan"), device="cuda")
    hip.depth_to_space(sites, r, back)
    assert torch.equal(back, x)
    hip.depth_to_space(sites, r, back, accumulate=True)
    assert torch.equal(back, x + x)


@pytest.mark.parametrize("n,h,w", [(1, 256, 256), (1, 1024, 1024), (2, 261, 334), (2, 130, 203), (1, 43, 97), (1, 11, 11)])
@pytest.mark.parametrize("relu", [True, False])
def test_stem_forward_as_3x3_against_fp64(env, n, h, w, relu):
    hip, m = env
    mod, g = _stem(m, 3)
    x = (torch.rand(n, 3, h, w, device="cuda", generator=g) * 255 - 120)  # the preprocessed image's range (mean-subtracted BGR 0-255)
    oh, ow = (h - 11) // 4 + 1, (w - 11) // 4 + 1
    assert m.conv_strided_fwd_is_3x3(mod, h, w) == ((oh + 2) * (ow + 2) >= 4096)  # planner field x3w_min_pixels; the form is exact below it too
    sites = torch.full((n, 48, oh + 2, ow + 2), float("nan"), device="cuda")
    out = torch.full((n, 96, oh, ow), float("nan"), device="cuda")
    m.conv_strided_fwd_as_3x3(x, mod, out, relu, sites)
    ref = F.conv2d(x.double(), mod.weight.double(), mod.bias.double(), stride=4)
    if relu:
        ref = ref.relu()
    assert rel_l2(out, ref) <= BAR
    # and no further from fp64 than the direct kernel it replaces (fp32 FMA chain over 363 taps)
    wf, _ = mod.banks()
    direct = hip.conv2d_fwd(x, wf, mod.bias_device(), 11, 4, 0, relu)
    assert rel_l2(out, ref) <= 2 * rel_l2(direct, ref) + 1e-8


@pytest.mark.parametrize("n,h,w", [(1, 1024, 1024), (2, 256, 300), (1, 99, 131), (2, 130, 67)])
def test_stem_backward_as_3x3_against_fp64(env, n, h, w):
    hip, m = env
    mod, g = _stem(m, 2)
    oh, ow = (h - 11) // 4 + 1, (w - 11) // 4 + 1
    gy = torch.randn(n, 96, oh, ow, device="cuda", generator=g) * (torch.rand(n, 96, oh, ow, device="cuda", generator=g) > 0.5)
    out = torch.full((n, 3, h, w), float("nan"), device="cuda")
    m.conv_strided_bwd_as_3x3(gy, mod, out)
    ref = torch.nn.grad.conv2d_input((n, 3, h, w), mod.weight.double(), gy.double(), stride=4)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) <= BAR


def test_eligibility_of_the_3x3_forms(env):
    hip, m = env
    import plan
    stem, _ = _stem(m, 1)
    assert m.conv_strided_fwd_is_3x3(stem, 1024, 1024) and m.conv_strided_bwd_is_3x3(stem, 254, 254)
    assert not m.conv_strided_fwd_is_3x3(stem, 10, 64)  # smaller than the filter
    wide, _ = _stem(m, 1, k=13)  # 13 > 3 x 4 taps
    assert not m.conv_strided_fwd_is_3x3(wide, 1024, 1024) and not m.conv_strided_bwd_is_3x3(wide, 253, 253)
    padded = m.Conv2d(3, 96, (11, 11), (4, 4), (2, 2)).cuda()
    assert not m.conv_strided_fwd_is_3x3(padded, 1024, 1024) and not m.conv_strided_bwd_is_3x3(padded, 255, 255)
    plan.OVERRIDES["strided_fwd_3x3"] = "0"
    plan.OVERRIDES["strided_bwd_3x3"] = "0"
    try:
        assert not m.conv_strided_fwd_is_3x3(stem, 1024, 1024) and not m.conv_strided_bwd_is_3x3(stem, 254, 254)
    finally:
        del plan.OVERRIDES["strided_fwd_3x3"], plan.OVERRIDES["strided_bwd_3x3"]


@pytest.mark.parametrize("S", [256, 301])
def test_nin_engine_with_and_without_the_3x3_stem_against_the_fp64_oracle(weight_files, S):
    """Config 5's network end to end at sizes where the stem's plane is large enough for the 3x3 forms (64 x 64 sites and up): the
    engine's route log names them, and loss / gradient meet the fp64 oracle's on either route (301: a ragged plane, two pixel rows
    and columns that no window of the stem covers).  The gradient's bar is the fp32 ORACLE's own distance from fp64: at these sizes a
    ReLU / max-pool decision of NIN flips under fp32 rounding (fp32 oracle vs fp64 oracle: 4.4e-3 at 256, 9.6e-4 at 301, 3e-6 at 128;
    tools/nin_fp32_vs_fp64.py) - the loss does not see it (3e-8)."""
    import engine
    import plan
    import synth
    from conftest import NIN_FLAGS, NIN_LAYERS, make_cfg, product_args
    from oracle.style_oracle import OracleNet, build_spec
    from test_engine_gpu import build
    content, style, init = synth.images(S)
    cfg = make_cfg(use_covariance=True, **NIN_LAYERS)
    onet = OracleNet(build_spec(cfg), synth.nin_state_dict(), torch.float64)
    onet.capture_content(content)
    onet.capture_style([style], cfg.style_blend_weights)
    total_o, _, grad_o = onet.feval(init)
    o32 = OracleNet(build_spec(cfg), synth.nin_state_dict(), torch.float32)
    o32.capture_content(content)
    o32.capture_style([style], cfg.style_blend_weights)
    floor = rel_l2(o32.feval(init)[2], grad_o)
    args = product_args(weight_files, NIN_FLAGS + ["--use_covariance"], model="nin", S=S)
    for on in ("1", "0"):
        plan.OVERRIDES["strided_fwd_3x3"] = plan.OVERRIDES["strided_bwd_3x3"] = on
        try:
            net, losses = build(args, content, [style], S)
            eng = engine.StyleEngine(net, losses)
            _, total, grad = eng.feval(init.cuda())
            torch.cuda.synchronize()
            total, grad = float(total), grad.cpu()
            routes = [r for r in eng.describe_routes(init.cuda()) if r.get("strided_as_3x3")]
        finally:
            del plan.OVERRIDES["strided_fwd_3x3"], plan.OVERRIDES["strided_bwd_3x3"]
        assert len(routes) == (2 if on == "1" else 0), routes
        assert abs(total - float(total_o)) <= 2e-5 * abs(float(total_o)), (on, total, float(total_o))
        assert rel_l2(grad, grad_o) <= max(2e-5, 1.5 * floor), (on, floor)
