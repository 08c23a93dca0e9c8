"""conv_x3p.hip - conv_x3q's workgroup made persistent (round 5): one stream of 32-channel chunks per CU over a static list of work
items, the next tile staged under the current one, an item's epilogue among the MFMAs of the next item's first chunk - against the fp64
arithmetic of the reference's layers, against conv_x3q.hip (same sums), and bit for bit against the launches its fused forms replace.

Reference arithmetic: `nn.Conv2d(cin, c, 3, padding=1)` + `nn.ReLU(inplace=True)` (/root/reference/models.py:129-130) and the
backward-data pass autograd derives from it; fused forms: `nn.MaxPool2d(2, 2)` (models.py:120) behind the layer, that pool's backward
pass in front of the layer's backward-data pass, and `GramMatrix` / `StyleLoss` backward (`torch.mm(x, x.t())`, /root/reference/loss.py:91,
its gradient D . F) added to the layer's input gradient.

Every case calls through the C ABI (maua_conv3x3_x3p) and compares with `F.conv2d(...double())`: contractions <= 2e-6 rel-L2 (measured
1.6-2.0e-7: the fp32 CPU convolution's own distance from fp64 is 1.6-1.9e-7), selections bit-exact.  `groups` = the workgroups a launch
may use (maua_conv_x3p_set_max_groups): with 8 or 16 every workgroup walks many items, so the cross-item pipeline (epilogue under the
next item's first chunk, staging across tile / channel-tile / image / split boundaries) runs on small planes; results must not depend
on it.  Shapes: the channel counts of VGG (64 ... 512), ragged planes (H % 16 != 0, W % 32 != 0), the odd planes of the 724 / 1448-px
pyramid (181 x 181, 90 x 91, 45 x 45), batches, both paddings, one-pass and split-K forms.
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
    yield h
    h.conv_x3p_set_max_groups(0)


@pytest.fixture(params=[8, 16, 256])
def groups(request, hip):
    hip.conv_x3p_set_max_groups(request.param)
    yield request.param
    hip.conv_x3p_set_max_groups(0)


def dev(t):
    return t.cuda().contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def one_pass_ws():
    """A workspace too small for split-K slabs: the entry point then makes one pass over the channels."""
    return torch.empty(16, dtype=torch.uint8, device="cuda")


# cin, cout, H, W, n, pad
X3P_CASES = [
    (32, 64, 16, 32, 1, 1),        # one chunk, one tile
    (32, 192, 70, 97, 2, 1),       # one chunk per item: every chunk carries an epilogue; three channel tiles, ragged plane, batch
    (64, 64, 64, 64, 1, 1),
    (64, 128, 67, 100, 1, 1),
    (64, 192, 66, 65, 1, 0),       # no padding (backward-data pads by 2)
    (128, 128, 64, 96, 2, 1),
    (128, 256, 75, 64, 1, 1),
    (256, 256, 64, 64, 1, 1),      # conv3_2..4
    (256, 512, 65, 70, 1, 1),      # conv4_1
    (256, 192, 45, 45, 2, 1),      # the deep plane of a 724-px image
    (128, 128, 181, 181, 1, 1),    # conv2 of a 724-px image: odd plane, ragged tiles on both edges
    (512, 512, 64, 64, 1, 1),      # conv4_2..4 / conv5_1
    (512, 192, 90, 91, 2, 1),      # conv4 of a 724 / 1448-px image: odd plane
    (512, 64, 130, 97, 1, 0),
]


@pytest.mark.parametrize("cin,cout,H,W,n,pad", X3P_CASES)
def test_conv3x3_x3p_forward_and_backward(hip, groups, cin, cout, H, W, n, pad):
    """Forward with bias + ReLU and backward-data with the ReLU mask of the layer's input, each in whatever form the cost model picks
    for the geometry AND forced into one pass over the channels; the one-pass result is the same whatever the number of workgroups."""
    assert hip.conv_x3p_supported(cin, H, W, cout, pad)
    x = torch.relu(rnd(n, cin, H, W, seed=1))
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=pad))
    bank_f, bank_b, wsc = hip.conv_pack_filters_x3q(dev(w))
    y = hip.conv3x3_x3p(dev(x), bank_f, wsc, dev(b), cout, pad, True)
    y1 = hip.conv3x3_x3p(dev(x), bank_f, wsc, dev(b), cout, pad, True, workspace=one_pass_ws())
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert rel_l2(y.cpu(), ref) <= BAR and rel_l2(y1.cpu(), ref) <= BAR
    hip.conv_x3p_set_max_groups(256)
    y256 = hip.conv3x3_x3p(dev(x), bank_f, wsc, dev(b), cout, pad, True, workspace=one_pass_ws())
    hip.conv_x3p_set_max_groups(groups)
    assert torch.equal(y1, y256)                                      # a tile's sums do not depend on its place in a workgroup's list
    gy = rnd(*ref.shape, seed=4) * (ref > 0)
    refb = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=pad) * (x > 0)
    if cin % 64 or cout % 32:
        assert not hip.conv_x3p_supported(cout, ref.shape[2], ref.shape[3], cin, 2 - pad)
        return
    gx = hip.conv3x3_x3p(dev(gy), bank_b, wsc, None, cin, 2 - pad, False, out_relu_mask=dev(x))
    gx1 = hip.conv3x3_x3p(dev(gy), bank_b, wsc, None, cin, 2 - pad, False, out_relu_mask=dev(x), workspace=one_pass_ws())
    torch.cuda.synchronize()
    assert gx.shape == x.shape
    assert rel_l2(gx.cpu(), refb) <= BAR and rel_l2(gx1.cpu(), refb) <= BAR
    assert torch.equal(gx == 0, dev(x) == 0) or float(((gx == 0) != (dev(x) == 0)).sum()) <= 1e-4 * gx.numel()


@pytest.mark.parametrize("cin,cout,H,W,n", [(64, 64, 66, 97, 2), (256, 192, 64, 64, 1), (512, 512, 64, 64, 1), (512, 128, 16, 16, 2)])
@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("masked", [False, True])
def test_conv3x3_x3p_every_flag_in_both_forms(hip, groups, cin, cout, H, W, n, bias, relu, masked):
    """y = [mask > 0] * relu?(conv(x) + bias?) for all eight flag combinations, one-pass and split-K (the 512-channel cases split),
    bit-identical reruns; buffers start as NaN (every element must be written)."""
    x = rnd(n, cin, H, W, seed=11)
    w = rnd(cout, cin, 3, 3, seed=12, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=13, scale=0.1) if bias else None
    mask = rnd(n, cout, H, W, seed=15)
    ref = F.conv2d(x.double(), w.double(), b.double() if bias else None, padding=1)
    if relu:
        ref = torch.relu(ref)
    if masked:
        ref = ref * (mask > 0)
    bank_f, _, wsc = hip.conv_pack_filters_x3q(dev(w))
    outs = []
    for ws in (None, one_pass_ws(), None):
        y = hip.conv3x3_x3p(dev(x), bank_f, wsc, dev(b) if bias else None, cout, 1, relu, out=torch.full((n, cout, H, W), float("nan"), device="cuda"),
                            out_relu_mask=dev(mask) if masked else None, workspace=ws)
        outs.append(y)
    torch.cuda.synchronize()
    for y in outs:
        assert rel_l2(y.cpu(), ref) <= BAR
    assert torch.equal(outs[0], outs[2])
    if cin == 512:
        assert hip.conv_x3p_split(n, cin, H, W, cout, 1) > 1          # these geometries do exercise the slabs


def test_conv3x3_x3p_agrees_with_x3q(hip, groups):
    """The persistent kernel computes conv_x3q's sums; the first chunk of every item folds its nine taps into the masters once instead of
    twice (its masters are being stored meanwhile): the results differ by fp32 rounding only, and each is as close to fp64 as the other."""
    cin, cout, H, W = 256, 256, 96, 96
    x = torch.relu(rnd(1, cin, H, W, seed=51))
    w = rnd(cout, cin, 3, 3, seed=52, scale=math.sqrt(2.0 / (9 * cin)))
    fq, _, wsq = hip.conv_pack_filters_x3q(dev(w))
    yq = hip.conv3x3_x3q(dev(x), fq, wsq, None, cout, 1, False, workspace=one_pass_ws())
    yp = hip.conv3x3_x3p(dev(x), fq, wsq, None, cout, 1, False, workspace=one_pass_ws())
    torch.cuda.synchronize()
    ref = F.conv2d(x[:, :, :34, :34].double(), w.double(), padding=1)[:, :, :32, :32]
    eq, ep = rel_l2(yq[:, :, :32, :32].cpu(), ref), rel_l2(yp[:, :, :32, :32].cpu(), ref)
    assert rel_l2(yp.cpu(), yq.cpu().double()) <= 4e-7
    assert ep <= 1.25 * eq + 1e-8 and ep <= 3e-7


@pytest.mark.parametrize("kind", ["wide_range", "tiny", "huge", "zeros", "one_hot", "hot_channel"])
@pytest.mark.parametrize("cin", [64, 512])
def test_conv3x3_x3p_scaling_survives_extreme_inputs(hip, groups, kind, cin):
    """fp16 has 5 exponent bits: the per-workgroup, per-32-channel power-of-two scaling must keep every magnitude usable."""
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
        x[0, 5] *= 1e6
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    ref = F.conv2d(x.double(), w.double(), padding=1)
    bank_f, _, wsc = hip.conv_pack_filters_x3q(dev(w))
    y = hip.conv3x3_x3p(dev(x), bank_f, wsc, None, cout, 1, False)
    y4 = hip.conv3x3_x3p(dev(x * 4.0), bank_f, wsc, None, cout, 1, False)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    if kind == "zeros":
        assert float(y.abs().max()) == 0.0
    else:
        assert rel_l2(y.cpu(), ref) <= BAR
        if kind != "huge":
            assert torch.equal(y4, y * 4.0)                         # exact homogeneity under power-of-two scaling


def test_conv3x3_x3p_index_exact(hip, groups):
    """Inputs that encode (channel, row, column) and one-tap selector filters: every output must be the EXACT input value the tap
    names - a wrong lane, octet, row, item or half of the fp16 pair shows as a wrong integer, not as a rounding error."""
    cin, cout, H, W = 64, 128, 40, 70
    c = torch.arange(cin).view(cin, 1, 1).float()
    yy = torch.arange(H).view(1, H, 1).float()
    xx = torch.arange(W).view(1, 1, W).float()
    x = (c + 100 * yy + 10000 * xx).unsqueeze(0)
    for tap in range(9):
        w = torch.zeros(cout, cin, 3, 3)
        for co in range(cout):
            w[co, (co * 7 + tap) % cin, tap // 3, tap % 3] = 1.0
        bank_f, _, wsc = hip.conv_pack_filters_x3q(dev(w))
        y = hip.conv3x3_x3p(dev(x), bank_f, wsc, None, cout, 1, False)
        torch.cuda.synchronize()
        assert torch.equal(y.cpu(), F.conv2d(x, w, padding=1)), tap


@pytest.mark.parametrize("n,cin,cout,H,W", [(1, 256, 256, 64, 64), (1, 512, 512, 64, 96), (2, 128, 192, 66, 70), (1, 64, 64, 130, 96),
                                           (1, 256, 256, 45, 91), (2, 128, 192, 33, 70), (1, 512, 64, 181, 181)])  # odd planes: floor-mode pooling
def test_x3p_conv_relu_pool_in_one_launch(hip, groups, n, cin, cout, H, W):
    """The pooling form: bit for bit maua_conv3x3_x3p + maua_pool2x2_fwd_codes in the one-pass and the split-K form, and
    max_pool2d(relu(conv2d)) in fp64 to 2e-6 with decision bytes that name a maximum of the fp64 window."""
    x = torch.relu(rnd(n, cin, H, W, seed=21))
    w = rnd(cout, cin, 3, 3, seed=22, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=23, scale=0.1)
    full = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))
    ref, _ = F.max_pool2d(full, 2, 2, return_indices=True)
    bank_f, _, wsc = hip.conv_pack_filters_x3q(dev(w))
    for ws in (one_pass_ws(), torch.empty(max(hip.conv_x3p_workspace_bytes(n, cin, H, W, cout, 1), 16), dtype=torch.uint8, device="cuda")):
        pooled = torch.full((n, cout, H // 2, W // 2), float("nan"), device="cuda")
        codes = torch.full((n, cout, H // 2, W // 2), 255, dtype=torch.uint8, device="cuda")
        hip.conv3x3_x3p(dev(x), bank_f, wsc, dev(b), cout, 1, True, out=pooled, pool_codes=codes, workspace=ws)
        act = hip.conv3x3_x3p(dev(x), bank_f, wsc, dev(b), cout, 1, True, workspace=ws)
        pooled2 = torch.empty_like(pooled)
        codes2 = torch.empty_like(codes)
        hip.pool2x2_fwd_codes(act, pooled2, codes2)
        torch.cuda.synchronize()
        assert torch.equal(pooled, pooled2) and torch.equal(codes, codes2)
    assert rel_l2(pooled.cpu(), ref) <= BAR
    codes = planar_codes(codes.cpu())
    assert int(codes.max()) <= 7
    clear_sign = ref.abs() > 1e-5
    assert torch.equal(((codes & 4) != 0)[clear_sign], (ref <= 0)[clear_sign])
    pos = (codes & 3).long()
    oh, ow = H // 2, W // 2
    rows = (torch.arange(oh)[:, None] * 2 + pos // 2)
    cols = (torch.arange(ow)[None, :] * 2 + pos % 2)
    picked = full[torch.arange(n)[:, None, None, None], torch.arange(cout)[None, :, None, None], rows, cols]
    assert float((picked - ref).abs().max()) <= 1e-5 * float(ref.abs().max())


# channels of the gradient (= couts of the layer), channels produced, full-size H, W, images
UNPOOL_CASES = [
    (64, 64, 128, 96, 1),
    (128, 128, 64, 64, 2),
    (256, 256, 66, 70, 1),         # conv3_4's channels, ragged tiles (even plane)
    (512, 512, 64, 64, 1),         # conv4_4: split-K at this size
    (32, 192, 18, 260, 1),
    (256, 256, 65, 71, 1),         # odd planes (724 / 1448-px images: 181 -> 90): the last row / column belongs to no window
    (128, 128, 45, 45, 2),
    (512, 512, 91, 64, 1),
]


@pytest.mark.parametrize("cg,c,H,W,n", UNPOOL_CASES)
@pytest.mark.parametrize("relu_bit", [True, False])
def test_x3p_backward_pass_straight_from_the_pooled_gradient(hip, groups, cg, c, H, W, n, relu_bit):
    """The unpooling form against maua_pool2x2_bwd_codes followed by the plain form: the same bits (one-pass and split-K forms, with and
    without the ReLU mask of the produced gradient), and against autograd's arithmetic in fp64 (max_pool2d backward +
    threshold_backward + conv_transpose2d, /root/reference/models.py:120,129-130)."""
    act = torch.relu(rnd(n, cg, H, W, seed=31))
    act[:, :, :4, :4] = 0.0                                            # all-zero windows: bit 2 of their bytes
    gp = rnd(n, cg, H // 2, W // 2, seed=32)
    w = rnd(cg, c, 3, 3, seed=33, scale=math.sqrt(2.0 / (9 * c)))
    fmap = torch.relu(rnd(n, c, H, W, seed=34))
    _, bb, wsc = hip.conv_pack_filters_x3q(dev(w))
    pooled = torch.empty(n, cg, H // 2, W // 2, device="cuda")
    codes = torch.empty(n, cg, H // 2, W // 2, dtype=torch.uint8, device="cuda")
    hip.pool2x2_fwd_codes(dev(act), pooled, codes)
    full = hip.pool2x2_bwd_codes(dev(gp), codes, torch.empty(n, cg, H, W, device="cuda"), relu_bit)
    a64 = act.double().requires_grad_(True)
    F.max_pool2d(a64, 2, 2).backward(gp.double())
    gfull = a64.grad * (act > 0) if relu_bit else a64.grad
    ref = F.conv_transpose2d(gfull, w.double(), padding=1)
    for mask in (None, dev(fmap)):
        for ws in (None, one_pass_ws()):
            two = hip.conv3x3_x3p(full, bb, wsc, None, c, 1, False, out_relu_mask=mask, workspace=ws)
            one = hip.conv3x3_x3p(dev(gp), bb, wsc, None, c, 1, False, out=torch.full((n, c, H, W), float("nan"), device="cuda"), out_relu_mask=mask,
                                  workspace=ws, in_codes=codes, honour_relu_bit=relu_bit)
            torch.cuda.synchronize()
            assert torch.equal(one, two), (mask is not None, ws is not None)
        assert rel_l2(one.cpu(), ref * (fmap > 0) if mask is not None else ref) <= BAR


# channels of the gradient, channels of F (= produced), H, W, images
GRAM_CASES = [(64, 64, 64, 96, 1), (128, 64, 70, 67, 2), (128, 128, 64, 64, 1), (256, 256, 48, 64, 1), (256, 128, 181, 181, 1), (512, 256, 33, 45, 1)]


@pytest.mark.parametrize("cg,c,H,W,n", GRAM_CASES)
@pytest.mark.parametrize("unpool", [False, True])
def test_x3p_gram_backward_rides_along(hip, groups, cg, c, H, W, n, unpool):
    """out = [F > 0] * (backward-data of the layer + D . F): the style loss's Gram backward on the layer's input activation F
    (`StyleLoss` on `GramMatrix`, /root/reference/loss.py:87-91, 141-181: d/dF of mse(G(F), T) is (D + D^T) F / n with D the scaled Gram
    difference; the engine passes the symmetric D) in the same launch, plain and staged from a pooled gradient; against fp64, against
    conv_x3w's fused form (to rounding) and bit-identical reruns."""
    if unpool and (H % 2 or W % 2):
        H, W = H - H % 2, W - W % 2
    gy_full = rnd(n, cg, H, W, seed=41) * (rnd(n, cg, H, W, seed=42) > 0)
    w = rnd(cg, c, 3, 3, seed=43, scale=math.sqrt(2.0 / (9 * c)))
    fmap = torch.relu(rnd(n, c, H, W, seed=44))
    D = rnd(c, c, seed=45, scale=1e-3)
    D = D + D.t()
    _, bb, wsc = hip.conv_pack_filters_x3q(dev(w))
    _, bbw, wscw = hip.conv_pack_filters_x3w(dev(w))
    bank = hip.conv_x3w_dmat_bank(c, "cuda", n)
    for f in range(n):
        hip.conv_pack_dmat_x3w(dev(D), bank[0][f], bank[1][f:f + 1])
    if unpool:
        act = torch.relu(rnd(n, cg, H, W, seed=46))
        gp = rnd(n, cg, H // 2, W // 2, seed=47)
        pooled = torch.empty(n, cg, H // 2, W // 2, device="cuda")
        codes = torch.empty(n, cg, H // 2, W // 2, dtype=torch.uint8, device="cuda")
        hip.pool2x2_fwd_codes(dev(act), pooled, codes)
        gy_dev = hip.pool2x2_bwd_codes(dev(gp), codes, torch.empty(n, cg, H, W, device="cuda"), True)
        gy_full = gy_dev.cpu()
        run = lambda ws: hip.conv3x3_x3p(dev(gp), bb, wsc, None, c, 1, False, out=torch.full((n, c, H, W), float("nan"), device="cuda"),
                                         out_relu_mask=dev(fmap), in_codes=codes, honour_relu_bit=True, dmat_bank=bank[0], dmat_inv_scale=bank[1], workspace=ws)
    else:
        run = lambda ws: hip.conv3x3_x3p(dev(gy_full), bb, wsc, None, c, 1, False, out=torch.full((n, c, H, W), float("nan"), device="cuda"),
                                         out_relu_mask=dev(fmap), dmat_bank=bank[0], dmat_inv_scale=bank[1], workspace=ws)
    ref = (F.conv_transpose2d(gy_full.double(), w.double(), padding=1) + torch.einsum("ij,njhw->nihw", D.double(), fmap.double())) * (fmap > 0)
    got = [run(None), run(one_pass_ws()), run(None)]
    yw = hip.conv3x3_x3w_gram(dev(gy_full), bbw, wscw, dev(fmap), bank[0], bank[1], c, 1, workspace=one_pass_ws())
    torch.cuda.synchronize()
    for y in got:
        assert rel_l2(y.cpu(), ref) <= BAR
    assert torch.equal(got[0], got[2])
    assert rel_l2(got[1].cpu(), yw.cpu().double()) <= 6e-7


def test_x3p_routing_rule(hip):
    """maua_conv_x3p_preferred: two items and more per workgroup in one pass over the channels, eight chunks and more per workgroup, an
    evenly dealt list; maua_conv_x3p_supported: whole 64-channel output tiles, 32-channel chunks, at most 512 output channels."""
    assert hip.conv_x3p_supported(64, 1024, 1024, 64, 1) and hip.conv_x3p_supported(512, 128, 128, 512, 1)
    assert not hip.conv_x3p_supported(48, 64, 64, 64, 1) and not hip.conv_x3p_supported(64, 64, 64, 96, 1) and not hip.conv_x3p_supported(64, 64, 64, 1024, 1)
    assert hip.conv_x3p_preferred(1, 64, 1024, 1024, 64, 1)          # conv1_2 of a 1024-px image: 2048 items of two chunks
    assert hip.conv_x3p_preferred(1, 256, 256, 256, 256, 1)          # conv3_2: 512 items of eight chunks
    assert not hip.conv_x3p_preferred(1, 512, 128, 128, 512, 1)      # conv4_2: one item per workgroup - conv_x3q
    assert not hip.conv_x3p_preferred(1, 64, 512, 512, 64, 1)        # conv1_2 of a 512-px image: two items of two chunks
    assert not hip.conv_x3p_preferred(1, 512, 64, 64, 512, 1)        # conv5_1: long enough only with the channel loop split
