"""conv_few_mfma.hip (round 5): the backward-data pass of the image layer - autograd's conv_backward of
`nn.Conv2d(3, 64, 3, padding=1)` (/root/reference/models.py:129-130): 64 gradient channels -> 3 pixel channels - on the matrix cores: per
gradient pixel the 27 (tap, channel) sums over the 64 channels as one bf16x6 product block, then nine values gathered per output pixel.

Against fp64 `torch.nn.grad.conv2d_input` through the C ABI: <= 2e-6 rel-L2 (measured 1.25e-7; conv3x3_few_out_kernel's fp32 FMA chain:
1.2e-7), every tile height, 1 - 3 pixel channels, ragged planes (W % 62 != 0, H % rows != 0, planes smaller than a tile), batches, sparse
(pre-masked) gradients; the engine's route log and gradient with and without the route.
"""
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

BAR = 2e-6


@pytest.fixture(scope="module")
def hip():
    import hip as h
    h.lib()
    return h


@pytest.mark.parametrize("n,cin,h,w", [(1, 3, 64, 64), (2, 3, 70, 131), (1, 3, 5, 7), (1, 1, 33, 62), (3, 2, 17, 63), (1, 3, 256, 256), (1, 3, 61, 125), (2, 3, 45, 124), (1, 2, 9, 4), (1, 3, 130, 260), (1, 1, 300, 116)])
@pytest.mark.parametrize("tile", [0, 1, 2, 3])
def test_few_mfma_against_fp64(hip, n, cin, h, w, tile):
    g = torch.Generator(device="cuda").manual_seed(3)
    wt = torch.randn(64, cin, 3, 3, device="cuda", generator=g) * 0.1
    gy = torch.randn(n, 64, h, w, device="cuda", generator=g) * (torch.rand(n, 64, h, w, device="cuda", generator=g) > 0.5)
    bank = hip.conv_pack_filters_few_mfma(wt)
    out = torch.full((n, cin, h, w), float("nan"), device="cuda")
    hip.conv3x3_few_mfma(gy, bank, cin, out=out, tile=tile)
    ref = torch.nn.grad.conv2d_input((n, cin, h, w), wt.double(), gy.double(), padding=1)
    assert torch.isfinite(out).all()
    assert rel_l2(out, ref) <= BAR
    again = torch.empty_like(out)
    hip.conv3x3_few_mfma(gy, bank, cin, out=again, tile=tile)
    assert torch.equal(out, again)
    if tile:  # the tile height decides which workgroup computes a pixel's 27 sums, not how: same bits as the library's choice
        assert torch.equal(out, hip.conv3x3_few_mfma(gy, bank, cin, tile=0))


def test_few_mfma_wide_dynamic_range_and_zero_gradient(hip):
    """Gradient values over 30 binades (bf16x6 keeps fp32's exponent range: no per-chunk scale), and an all-zero gradient."""
    g = torch.Generator(device="cuda").manual_seed(4)
    wt = torch.randn(64, 3, 3, 3, device="cuda", generator=g)
    gy = torch.randn(1, 64, 40, 70, device="cuda", generator=g) * torch.exp2(torch.randint(-15, 15, (1, 64, 1, 1), device="cuda", generator=g).float())
    bank = hip.conv_pack_filters_few_mfma(wt)
    out = hip.conv3x3_few_mfma(gy, bank, 3)
    assert rel_l2(out, torch.nn.grad.conv2d_input((1, 3, 40, 70), wt.double(), gy.double(), padding=1)) <= BAR
    assert not hip.conv3x3_few_mfma(torch.zeros_like(gy), bank, 3).any()


def test_few_mfma_rejects_what_it_does_not_cover(hip):
    assert hip.conv_few_mfma_supported(1, 3, 64, 64, 64, 1)
    assert not hip.conv_few_mfma_supported(1, 3, 64, 64, 64, 0)   # padding 0: the gradient is not image-sized
    assert not hip.conv_few_mfma_supported(1, 3, 64, 64, 96, 1)   # 96 filters
    assert not hip.conv_few_mfma_supported(1, 4, 64, 64, 64, 1)   # four pixel channels: 36 columns
    assert not hip.conv_few_mfma_supported(1, 3, 4096, 4096, 64, 1)  # 64 maps of 64 MiB: beyond the descriptor's 2 GiB
    with pytest.raises(hip.HipError):
        hip.conv_pack_filters_few_mfma(torch.zeros(96, 3, 3, 3, device="cuda"))
    bank = hip.conv_pack_filters_few_mfma(torch.zeros(64, 3, 3, 3, device="cuda"))
    with pytest.raises(hip.HipError):
        hip.conv3x3_few_mfma(torch.zeros(1, 64, 8, 8, device="cuda"), bank, 3, tile=4)


@pytest.mark.parametrize("S", [64, 130, 256])
def test_engine_routes_the_image_layer_backward_and_matches_the_vector_kernel(weight_files, S):
    """The VGG-19 engine takes conv_few_mfma for conv1_1's backward pass (route log), and loss / pixel gradient
    agree with the vector-ALU kernel's route to fp32 rounding (each route is held to the goldens / the fp64 arbiter by test_engine_gpu)."""
    import engine
    import plan
    import synth
    from conftest import product_args
    from test_engine_gpu import build
    args = product_args(weight_files, S=S)
    content, style, init = synth.images(S)
    res = {}
    for on in ("1", "0"):
        plan.OVERRIDES["few_mfma"] = on
        try:
            net, losses = build(args, content, [style], S)
            eng = engine.StyleEngine(net, losses)
            _, total, grad = eng.feval(init.cuda())
            torch.cuda.synchronize()
            res[on] = (float(total), grad.clone(), [r["kernel"] for r in eng.describe_routes(init.cuda()) if r["pass"] == "bwd"][-1])
        finally:
            del plan.OVERRIDES["few_mfma"]
    assert res["1"][2] == "conv_few_mfma" and res["0"][2] == "conv3x3_few_out", (res["1"][2], res["0"][2])
    assert res["1"][0] == res["0"][0]  # (the forward pass is the same)
    assert rel_l2(res["1"][1], res["0"][1]) <= 1e-6
