"""conv_img.hip (the image layer: a 3x3 convolution that consumes 1-3 channels, exact bf16x6 arithmetic, bias as a filter column) against fp64.

Reference arithmetic: `nn.Conv2d(3, 64, 3, padding=1)` + `nn.ReLU(inplace=True)` (/root/reference/models.py:129-130).  The bar is the one of the
general bf16x6 kernel it replaces for this layer: rel-L2 <= 3e-7 against fp64 and not worse than 1.5x the fp32 CPU convolution."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import hip as h
    h.lib()
    return h


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


# cin, cout, H, W, n, pad
CASES = [
    (3, 64, 64, 64, 1, 1),
    (3, 64, 70, 97, 2, 1),        # ragged row blocks, batch
    (3, 64, 33, 31, 1, 1),        # a single partial block per row
    (3, 96, 40, 130, 1, 1),       # second 64-channel tile half empty
    (3, 200, 17, 64, 1, 1),
    (1, 64, 48, 80, 1, 1),
    (2, 32, 19, 63, 1, 0),        # no padding
    (3, 64, 24, 40, 1, 2),
    (3, 64, 256, 256, 1, 1),
]


@pytest.mark.parametrize("cin,cout,H,W,n,pad", CASES)
@pytest.mark.parametrize("relu", [True, False])
@pytest.mark.parametrize("with_bias", [True, False])
def test_image_layer_against_fp64(hip, cin, cout, H, W, n, pad, relu, with_bias):
    # image-like input: large common magnitude (network-space pixels are around +-100), neighbouring pixels close
    x = rnd(n, cin, H, W, seed=1) * 3.0 + 100.0 * torch.sin(torch.arange(W) / 7.0)[None, None, None, :]
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=3, scale=5.0) if with_bias else None
    ref = F.conv2d(x.double(), w.double(), b.double() if with_bias else None, padding=pad)
    f32 = F.conv2d(x, w, b, padding=pad)
    if relu:
        ref, f32 = torch.relu(ref), torch.relu(f32)
    bank = hip.conv_pack_filters_image(w.cuda(), b.cuda() if with_bias else None)
    y = hip.conv3x3_image(x.cuda().contiguous(), bank, cout, pad, relu)
    y2 = hip.conv3x3_image(x.cuda().contiguous(), bank, cout, pad, relu)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    err, floor = rel_l2(y.cpu(), ref), rel_l2(f32, ref)
    assert err <= 3e-7 and err <= 1.5 * floor + 2e-8, (err, floor)
    assert torch.equal(y, y2)
    # against the general kernel it stands in for: the same six products per pair, another summation order
    bf, _ = hip.conv_pack_filters_x6(w.cuda())
    y6 = hip.conv3x3_x6(x.cuda().contiguous(), bf, b.cuda() if with_bias else None, cout, pad, relu)
    torch.cuda.synchronize()
    assert rel_l2(y.cpu(), y6.cpu().double()) <= 3e-7


def test_image_layer_is_exactly_homogeneous_and_survives_extremes(hip):
    """bf16 triples carry all 24 bits of every operand: scaling the image by a power of two scales the output exactly; huge and tiny
    magnitudes stay finite and accurate."""
    x = rnd(1, 3, 40, 72, seed=5) * 50.0
    w = rnd(64, 3, 3, 3, seed=6, scale=0.3)
    bank = hip.conv_pack_filters_image(w.cuda(), None)
    y = hip.conv3x3_image(x.cuda(), bank, 64, 1, False)
    for s in (2.0 ** 20, 2.0 ** -30):
        ys = hip.conv3x3_image((x * s).cuda(), bank, 64, 1, False)
        torch.cuda.synchronize()
        assert torch.equal(ys, y * s)
    ref = F.conv2d(x.double() * 1e30, w.double(), padding=1)
    yb = hip.conv3x3_image((x * 1e30).cuda(), bank, 64, 1, False)
    torch.cuda.synchronize()
    assert torch.isfinite(yb).all() and rel_l2(yb.cpu(), ref) <= 3e-7


@pytest.mark.parametrize("H,W", [(64, 64), (70, 97), (33, 31), (256, 256), (512, 384)])
def test_image_layer_with_the_gram_matrix_along(hip, H, W):
    """maua_conv3x3_image_gram: the activation bit-identical to maua_conv3x3_image, and the slabs' sum (upper channel blocks) = Y Y^T of
    that activation in fp64 (reference: torch.mm(x, x.T), /root/reference/loss.py:91, on relu1_1)."""
    x = rnd(1, 3, H, W, seed=11) * 40.0
    w = rnd(64, 3, 3, 3, seed=12, scale=0.3)
    b = rnd(64, seed=13, scale=5.0)
    bank = hip.conv_pack_filters_image(w.cuda(), b.cuda())
    y0 = hip.conv3x3_image(x.cuda(), bank, 64, 1, True)
    n = hip.conv_image_gram_slabs(H, W, 1)
    slabs = torch.full((n, 64, 64), float("nan"), device="cuda")
    y1 = hip.conv3x3_image_gram(x.cuda(), bank, 1, torch.empty_like(y0), slabs)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    g = slabs.double().sum(0).cpu()
    f = y0.double().cpu().reshape(64, -1)
    ref = f @ f.t()
    upper = torch.ones(64, 64, dtype=torch.bool).triu()
    upper[32:, :32] = False
    err = float((g - ref)[upper].norm() / ref[upper].norm())
    assert err <= 3e-7, err
    # the finishing kernels read the lower triangle of the diagonal blocks from the upper one: what is there must be finite at least
    assert torch.isfinite(g[:32, :32]).all() and torch.isfinite(g[32:, 32:]).all() and torch.isfinite(g[:32, 32:]).all()
    again = torch.empty_like(slabs)
    hip.conv3x3_image_gram(x.cuda(), bank, 1, torch.empty_like(y0), again)
    torch.cuda.synchronize()
    assert torch.equal(again[:, :32], slabs[:, :32]) and torch.equal(again[:, 32:, 32:], slabs[:, 32:, 32:])


@pytest.mark.parametrize("cout", [22, 24, 32, 64, 96])
def test_image_layer_writes_nothing_outside_its_output(hip, cout):
    """The kernel masks channels past `cout` (and lanes past a row's end) by the range of its store descriptor: a 64-channel tile of which
    only 22 - 32 exist (the pruned VGG-16's first layer has 24) must leave what lies behind the output tensor alone."""
    H, W = 37, 50
    x = rnd(1, 3, H, W, seed=5, scale=100.0).cuda()
    w = rnd(cout, 3, 3, 3, seed=6, scale=0.3)
    b = rnd(cout, seed=7, scale=1.0)
    bank = hip.conv_pack_filters_image(w.cuda(), b.cuda())
    n_out = cout * H * W
    guard = 70 * H * W                      # more than a whole 64-channel tile on either side
    buf = torch.full((guard + n_out + guard,), 12345.0, device="cuda")
    out = buf[guard:guard + n_out].view(1, cout, H, W)
    hip.conv3x3_image(x, bank, cout, 1, True, out=out)
    torch.cuda.synchronize()
    assert bool((buf[:guard] == 12345.0).all()) and bool((buf[guard + n_out:] == 12345.0).all())
    want = torch.relu(torch.nn.functional.conv2d(x.cpu().double(), w.double(), b.double(), padding=1))
    assert float((out.cpu().double() - want).norm() / want.norm()) <= 2e-6
