"""conv_x3w.hip - the kernel the benchmark times - against the fp64 arithmetic of the reference's layers, directly.

Reference arithmetic: `nn.Conv2d(cin, c, 3, padding=1)` + `nn.ReLU(inplace=True)` (/root/reference/models.py:129-130) and the
backward-data pass autograd derives from it; fused variants: `nn.MaxPool2d(2, 2)` (models.py:120) behind the layer, and the
backward of `torch.mm(x, x.t())` (loss.py:91) riding on the backward-data pass of the following layer.

Every case calls through the C ABI (maua_conv3x3_x3w / _relu_pool / _gram) and compares with `F.conv2d(...double())`:
contractions <= 2e-6 rel-L2 (measured 1.1-1.8e-7), selections bit-exact.  Shapes cover every input-channel count of the
benched network (64 ... 512), ragged planes (H % 8 != 0, W % 32 != 0), ragged output-channel tiles (Cout % 64 != 0), batches,
both paddings, and every flag of the entry point, in the one-pass and in the split-K form.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import planar_codes, rel_l2

pytestmark = pytest.mark.gpu

BAR = 2e-6


@pytest.fixture(scope="module")
def hip():
    import hip as h
    h.lib()
    return h


def dev(t):
    return t.cuda().contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def one_pass_ws():
    """A workspace too small for split-K slabs: the entry point then makes one pass over the channels."""
    return torch.empty(16, dtype=torch.uint8, device="cuda")


# cin, cout, H, W, n, pad
X3W_CASES = [
    (16, 64, 64, 64, 1, 1),
    (16, 200, 70, 97, 2, 1),       # one chunk, ragged cout tile, ragged plane, batch
    (64, 64, 64, 64, 1, 1),        # conv1_2's channels on the smallest plane the engine sends here
    (64, 64, 130, 97, 2, 1),
    (64, 128, 67, 100, 1, 1),      # conv2_1
    (64, 200, 66, 65, 1, 0),       # no padding (backward-data pads by 2)
    (128, 128, 64, 96, 2, 1),      # conv2_2
    (128, 256, 75, 64, 1, 1),      # conv3_1
    (128, 64, 130, 97, 1, 0),
    (256, 256, 64, 64, 1, 1),      # conv3_2..4
    (256, 512, 65, 70, 1, 1),      # conv4_1
    (256, 200, 70, 67, 2, 1),
    (512, 512, 64, 64, 1, 1),      # conv4_2..4 / conv5_1: 40 % of the benched FLOPs
    (512, 512, 72, 97, 1, 1),
    (512, 200, 64, 66, 2, 1),
    (512, 64, 130, 97, 1, 0),
]


@pytest.mark.parametrize("cin,cout,H,W,n,pad", X3W_CASES)
def test_conv3x3_x3w_forward_and_backward(hip, cin, cout, H, W, n, pad):
    """Forward with bias + ReLU and backward-data with the ReLU mask of the layer's input, each in whatever form the cost
    model picks for the geometry AND forced into one pass over the channels."""
    assert hip.conv_x3w_supported(cin, H, W, pad)
    x = torch.relu(rnd(n, cin, H, W, seed=1))                       # post-ReLU activations: half the values are zero
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=pad))
    bank_f, bank_b, wsc = hip.conv_pack_filters_x3w(dev(w))
    assert math.log2(wsc) == int(math.log2(wsc)) and 32 <= float(w.abs().max()) * wsc < 64
    y = hip.conv3x3_x3w(dev(x), bank_f, wsc, dev(b), cout, pad, True)
    y1 = hip.conv3x3_x3w(dev(x), bank_f, wsc, dev(b), cout, pad, True, workspace=one_pass_ws())
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert rel_l2(y.cpu(), ref) <= BAR and rel_l2(y1.cpu(), ref) <= BAR
    assert torch.equal(y == 0, y1 == 0) or float(((y == 0) != (y1 == 0)).sum()) <= 1e-5 * y.numel()
    # backward-data: the gradient arrives masked (half zeros), the result is masked by the layer's input
    gy = rnd(*ref.shape, seed=4) * (ref > 0)
    refb = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=pad) * (x > 0)
    assert hip.conv_x3w_supported(cout, ref.shape[2], ref.shape[3], 2 - pad) == (cout % 16 == 0)
    if cout % 16:
        return
    gx = hip.conv3x3_x3w(dev(gy), bank_b, wsc, None, cin, 2 - pad, False, out_relu_mask=dev(x))
    gx1 = hip.conv3x3_x3w(dev(gy), bank_b, wsc, None, cin, 2 - pad, False, out_relu_mask=dev(x), workspace=one_pass_ws())
    torch.cuda.synchronize()
    assert gx.shape == x.shape
    assert rel_l2(gx.cpu(), refb) <= BAR and rel_l2(gx1.cpu(), refb) <= BAR
    assert torch.equal(gx == 0, dev(x) == 0) or float(((gx == 0) != (dev(x) == 0)).sum()) <= 1e-4 * gx.numel()


@pytest.mark.parametrize("cin,cout,H,W,n", [(64, 64, 66, 97, 2), (256, 200, 64, 64, 1), (512, 512, 64, 64, 1), (512, 128, 16, 16, 2)])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("masked", [False, True])
def test_conv3x3_x3w_every_flag_in_both_forms(hip, cin, cout, H, W, n, bias, relu, accumulate, masked):
    """y = [mask > 0] * relu?(conv(x) + bias? + y_before?) for all sixteen flag combinations, one-pass and split-K (the
    (512, 128, 16, 16) and 64 x 64 planes split; the others run one pass either way), bit-identical reruns."""
    x = rnd(n, cin, H, W, seed=11)
    w = rnd(cout, cin, 3, 3, seed=12, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=13, scale=0.1) if bias else None
    base = rnd(n, cout, H, W, seed=14)
    mask = rnd(n, cout, H, W, seed=15)
    ref = F.conv2d(x.double(), w.double(), b.double() if bias else None, padding=1)
    if accumulate:
        ref = ref + base.double()
    if relu:
        ref = torch.relu(ref)
    if masked:
        ref = ref * (mask > 0)
    bank_f, _, wsc = hip.conv_pack_filters_x3w(dev(w))
    outs = []
    for ws in (None, one_pass_ws(), None):
        y = hip.conv3x3_x3w(dev(x), bank_f, wsc, dev(b) if bias else None, cout, 1, relu, out=dev(base.clone()),
                            out_relu_mask=dev(mask) if masked else None, accumulate=accumulate, workspace=ws)
        outs.append(y)
    torch.cuda.synchronize()
    for y in outs:
        assert rel_l2(y.cpu(), ref) <= BAR
    assert torch.equal(outs[0], outs[2])
    if cin == 512:
        assert hip.conv_x3w_split(n, cin, H, W, cout, 1) > 1          # these geometries do exercise the slabs


@pytest.mark.parametrize("kind", ["wide_range", "tiny", "huge", "zeros", "one_hot", "hot_channel"])
@pytest.mark.parametrize("cin", [64, 512])
def test_conv3x3_x3w_scaling_survives_extreme_inputs(hip, kind, cin):
    """fp16 has 5 exponent bits: the per-workgroup, per-16-channel power-of-two scaling must keep every magnitude usable."""
    cout, H, W = 64, 72, 80
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, cin, H, W, generator=g)
    if kind == "wide_range":
        x = x * torch.exp(torch.randn(1, cin, H, W, generator=g) * 4.0) * 1e-6 * (torch.rand(1, cin, H, W, generator=g) > 0.5)
    elif kind == "tiny":
        x = x * 1e-30
    elif kind == "huge":
        x = x * 1e30
    elif kind == "zeros":
        x = torch.zeros_like(x)
    elif kind == "one_hot":
        x = torch.zeros_like(x)
        x[0, 17, 20, 21] = 3.0e-12
    elif kind == "hot_channel":
        x[0, 5] *= 1e6                                              # one chunk 10^6 above the other chunks of the same tile
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    ref = F.conv2d(x.double(), w.double(), padding=1)
    bank_f, _, wsc = hip.conv_pack_filters_x3w(dev(w))
    y = hip.conv3x3_x3w(dev(x), bank_f, wsc, None, cout, 1, False)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    if kind == "zeros":
        assert float(y.abs().max()) == 0.0
    else:
        assert rel_l2(y.cpu(), ref) <= BAR


@pytest.mark.parametrize("n,cin,cout,H,W", [(1, 256, 256, 64, 64), (1, 512, 512, 64, 96), (2, 128, 200, 66, 70), (1, 64, 64, 130, 96),
                                           (1, 256, 256, 45, 91), (2, 128, 200, 33, 70), (1, 512, 64, 181, 181)])  # odd planes: floor-mode pooling
def test_conv_relu_pool_in_one_launch_against_fp64(hip, n, cin, cout, H, W):
    """maua_conv3x3_x3w_relu_pool (conv3_4 / conv4_4 shapes among them) against max_pool2d(relu(conv2d)) in fp64: the pooled map
    to 2e-6, and decision bytes that name a maximum of the fp64 window wherever that maximum is clear of its runner-up."""
    x = torch.relu(rnd(n, cin, H, W, seed=21))
    w = rnd(cout, cin, 3, 3, seed=22, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=23, scale=0.1)
    full = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    ref, idx = F.max_pool2d(full, 2, 2, return_indices=True)
    bank_f, _, wsc = hip.conv_pack_filters_x3w(dev(w))
    pooled = torch.full((n, cout, H // 2, W // 2), float("nan"), device="cuda")
    codes = torch.full((n, cout, H // 2, W // 2), 255, dtype=torch.uint8, device="cuda")
    hip.conv3x3_x3w_relu_pool(dev(x), bank_f, wsc, dev(b), cout, 1, pooled, codes)
    torch.cuda.synchronize()
    assert rel_l2(pooled.cpu(), ref) <= BAR
    codes = planar_codes(codes.cpu())
    assert int(codes.max()) <= 7
    # bit 2 = "the maximum is not positive"
    clear_sign = ref.abs() > 1e-5
    assert torch.equal(((codes & 4) != 0)[clear_sign], (ref <= 0)[clear_sign])
    # bits 1:0 = position of the maximum in the window (row-major)
    pos = (codes & 3).long()
    oh, ow = H // 2, W // 2
    rows = (torch.arange(oh)[:, None] * 2 + pos // 2)
    cols = (torch.arange(ow)[None, :] * 2 + pos % 2)
    picked = full[torch.arange(n)[:, None, None, None], torch.arange(cout)[None, :, None, None], rows, cols]
    assert float((picked - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


@pytest.mark.parametrize("cin,c,H,W", [(256, 256, 64, 64), (256, 256, 70, 97), (256, 128, 64, 96), (512, 256, 64, 64)])
def test_backward_pass_with_the_gram_backward_along_at_256_channels(hip, cin, c, H, W):
    """maua_conv3x3_x3w_gram on relu3_1's channel count (the deepest layer the engine fuses, MAUA_FUSE_GRAM_MAX_C = 256) against
    fp64: [F > 0] * (conv_transpose(gy) + D . F)."""
    gy = rnd(1, cin, H, W, seed=71) * (rnd(1, cin, H, W, seed=72) > 0)
    w = rnd(cin, c, 3, 3, seed=73) * (2.0 / (9 * c)) ** 0.5
    f = torch.relu(rnd(1, c, H, W, seed=74))
    d = rnd(c, c, seed=75) * 1e-3
    d = (d + d.t()).contiguous()
    _, bb, wsc = hip.conv_pack_filters_x3w(dev(w))
    dbank, dinv = hip.conv_x3w_dmat_bank(c, "cuda")
    hip.conv_pack_dmat_x3w(dev(d), dbank, dinv)
    fused = hip.conv3x3_x3w_gram(dev(gy), bb, wsc, dev(f), dbank, dinv, c, 1)
    again = hip.conv3x3_x3w_gram(dev(gy), bb, wsc, dev(f), dbank, dinv, c, 1)
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(gy.double(), w.double(), padding=1) + torch.einsum("kc,bkhw->bchw", d.double(), f.double())
    ref = ref * (f > 0)
    assert rel_l2(fused.cpu(), ref) <= 3e-7
    assert torch.equal(fused, again)
    # the convolution part and the Gram part separately (a scale error in one of them must not hide behind the other)
    zero_d = torch.zeros_like(d)
    hip.conv_pack_dmat_x3w(dev(zero_d), dbank, dinv)
    conv_only = hip.conv3x3_x3w_gram(dev(gy), bb, wsc, dev(f), dbank, dinv, c, 1)
    hip.conv_pack_dmat_x3w(dev(d), dbank, dinv)
    gram_only = hip.conv3x3_x3w_gram(dev(torch.zeros_like(gy)), bb, wsc, dev(f), dbank, dinv, c, 1)
    torch.cuda.synchronize()
    assert rel_l2(conv_only.cpu(), F.conv_transpose2d(gy.double(), w.double(), padding=1) * (f > 0)) <= BAR
    assert rel_l2(gram_only.cpu(), torch.einsum("kc,bkhw->bchw", d.double(), f.double()) * (f > 0)) <= BAR


# channels of the gradient (= couts of the layer), channels produced, full-size H, W, images
UNPOOL_CASES = [
    (64, 64, 128, 96, 1),          # conv1_2's group
    (128, 128, 64, 64, 2),
    (256, 256, 66, 70, 1),         # ragged tiles (even plane)
    (512, 512, 64, 64, 1),         # conv4_4: split-K at this size
    (32, 200, 18, 260, 1),         # ragged cout tile of the produced gradient
    (256, 256, 65, 71, 1),         # odd planes (724 / 1448-px images: 181 -> 90): the last row / column belongs to no window
    (128, 128, 45, 45, 2),
    (512, 512, 91, 64, 1),
]


@pytest.mark.parametrize("cg,c,H,W,n", UNPOOL_CASES)
@pytest.mark.parametrize("relu_bit", [True, False])
def test_backward_pass_straight_from_the_pooled_gradient(hip, cg, c, H, W, n, relu_bit):
    """maua_conv3x3_x3w_unpool against maua_pool2x2_bwd_codes followed by maua_conv3x3_x3w: the same bits (one-pass and split-K forms,
    with and without the ReLU mask of the produced gradient), and against autograd's arithmetic in fp64
    (max_pool2d backward + threshold_backward + conv_transpose2d, /root/reference/models.py:120,129-130)."""
    act = torch.relu(rnd(n, cg, H, W, seed=31))
    act[:, :, :4, :4] = 0.0                                            # all-zero windows: bit 2 of their bytes
    gp = rnd(n, cg, H // 2, W // 2, seed=32)
    w = rnd(cg, c, 3, 3, seed=33, scale=math.sqrt(2.0 / (9 * c)))
    fmap = torch.relu(rnd(n, c, H, W, seed=34))
    _, bb, wsc = hip.conv_pack_filters_x3w(dev(w))
    pooled = torch.empty(n, cg, H // 2, W // 2, device="cuda")
    codes = torch.empty(n, cg, H // 2, W // 2, dtype=torch.uint8, device="cuda")
    hip.pool2x2_fwd_codes(dev(act), pooled, codes)
    full = hip.pool2x2_bwd_codes(dev(gp), codes, torch.empty(n, cg, H, W, device="cuda"), relu_bit)
    # fp64 reference of the unpooled gradient
    a64 = act.double().requires_grad_(True)
    F.max_pool2d(a64, 2, 2).backward(gp.double())
    gfull = a64.grad * (act > 0) if relu_bit else a64.grad
    assert torch.equal(full.cpu().double(), gfull)                     # (routing only: exact)
    ref = F.conv_transpose2d(gfull, w.double(), padding=1)
    for mask in (None, dev(fmap)):
        for ws in (None, one_pass_ws()):
            two = hip.conv3x3_x3w(full, bb, wsc, None, c, 1, False, out_relu_mask=mask, workspace=ws)
            one = hip.conv3x3_x3w_unpool(dev(gp), codes, relu_bit, bb, wsc, c, 1, out=torch.empty(n, c, H, W, device="cuda"), out_relu_mask=mask,
                                               workspace=ws)
            torch.cuda.synchronize()
            assert torch.equal(one, two), (mask is not None, ws is not None)
        assert rel_l2(one.cpu(), ref * (fmap > 0) if mask is not None else ref) <= BAR


@pytest.mark.parametrize("cg,c,H,W", [(64, 64, 128, 96), (128, 128, 66, 70), (256, 256, 64, 64)])
def test_unpooling_backward_pass_with_the_gram_backward_along(hip, cg, c, H, W):
    """The same launch carrying the Gram backward of the style loss on the layer's input (relu1_1 / relu2_1 / relu3_1 in the network):
    bit-identical to maua_pool2x2_bwd_codes + maua_conv3x3_x3w_gram, and [F > 0] (conv_transpose(unpool(g)) + D F) in fp64."""
    act = torch.relu(rnd(1, cg, H, W, seed=41))
    gp = rnd(1, cg, H // 2, W // 2, seed=42)
    w = rnd(cg, c, 3, 3, seed=43, scale=math.sqrt(2.0 / (9 * c)))
    f = torch.relu(rnd(1, c, H, W, seed=44))
    d = rnd(c, c, seed=45) * 1e-3
    d = (d + d.t()).contiguous()
    _, bb, wsc = hip.conv_pack_filters_x3w(dev(w))
    dbank, dinv = hip.conv_x3w_dmat_bank(c, "cuda")
    hip.conv_pack_dmat_x3w(dev(d), dbank, dinv)
    pooled = torch.empty(1, cg, H // 2, W // 2, device="cuda")
    codes = torch.empty(1, cg, H // 2, W // 2, dtype=torch.uint8, device="cuda")
    hip.pool2x2_fwd_codes(dev(act), pooled, codes)
    full = hip.pool2x2_bwd_codes(dev(gp), codes, torch.empty(1, cg, H, W, device="cuda"), True)
    two = hip.conv3x3_x3w_gram(full, bb, wsc, dev(f), dbank, dinv, c, 1)
    one = hip.conv3x3_x3w_unpool(dev(gp), codes, True, bb, wsc, c, 1, out_relu_mask=dev(f), dmat_bank=dbank, dmat_inv_scale=dinv)
    torch.cuda.synchronize()
    assert torch.equal(one, two)
    a64 = act.double().requires_grad_(True)
    F.max_pool2d(a64, 2, 2).backward(gp.double())
    ref = F.conv_transpose2d(a64.grad * (act > 0), w.double(), padding=1) + torch.einsum("kc,bkhw->bchw", d.double(), f.double())
    assert rel_l2(one.cpu(), ref * (f > 0)) <= 3e-7
