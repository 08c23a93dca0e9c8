"""(Experiment, outside the product since round 4: build with tools/wino/build.sh, run `python -m pytest tools/wino/test_conv_wino_gpu.py`.)
conv_wino.hip (Winograd F(2x2, 3x3) on the fp16x3 split) against the fp64 arithmetic of the reference's layers.

Reference arithmetic: `nn.Conv2d(cin, c, 3, padding=1)` + `nn.ReLU(inplace=True)` (/root/reference/models.py:129-130) and the
backward-data pass autograd derives from it.  Winograd's transforms add roundings of their own: the bar is rel-L2 <= 3e-7
against fp64 per layer AND <= 1.5x the error of the fp32 CPU convolution (the reference's own arithmetic) on the same data.
"""
import math

import pytest
import torch
import torch.nn.functional as F

import os
import sys

sys.path[:0] = [os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests")]
import wino  # noqa: E402
from conftest import rel_l2  # noqa: E402

pytestmark = pytest.mark.gpu


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


# cin, cout, H, W, n, pad
WINO_CASES = [
    (16, 64, 16, 32, 1, 1),        # one chunk, one tile
    (64, 64, 64, 64, 1, 1),
    (64, 200, 70, 97, 2, 1),       # ragged cout tile, odd width, batch
    (128, 128, 64, 96, 2, 1),
    (128, 256, 75, 64, 1, 1),      # odd height
    (256, 256, 64, 64, 1, 1),
    (256, 512, 65, 70, 1, 1),
    (512, 512, 64, 64, 1, 1),
    (512, 512, 72, 98, 1, 1),
    (512, 64, 130, 97, 1, 0),      # no padding (backward-data pads by 2)
    (48, 24, 9, 11, 1, 1),         # tiny plane, few channels
]


@pytest.mark.parametrize("cin,cout,H,W,n,pad", WINO_CASES)
def test_conv3x3_wino_forward_and_backward(hip, cin, cout, H, W, n, pad):
    assert wino.conv_wino_supported(cin, H, W, pad)
    x = torch.relu(rnd(n, cin, H, W, seed=1))
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=pad))
    floor = rel_l2(torch.relu(F.conv2d(x, w, b, padding=pad)), ref)
    bank_f, bank_b = wino.conv_pack_filters_wino(dev(w))
    y = wino.conv3x3_wino(dev(x), bank_f, dev(b), cout, pad, True)
    y2 = wino.conv3x3_wino(dev(x), bank_f, dev(b), cout, pad, True)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    err = rel_l2(y.cpu(), ref)
    assert err <= 3e-7 and err <= 1.5 * floor + 2e-8, (err, floor)
    assert torch.equal(y, y2)
    if cout % 16:
        return
    gy = rnd(*ref.shape, seed=4) * (ref > 0)
    refb = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=pad) * (x > 0)
    floor_b = rel_l2(torch.nn.grad.conv2d_input(x.shape, w, gy, padding=pad) * (x > 0), refb)
    gx = wino.conv3x3_wino(dev(gy), bank_b, None, cin, 2 - pad, False, out_relu_mask=dev(x))
    torch.cuda.synchronize()
    assert gx.shape == x.shape
    err = rel_l2(gx.cpu(), refb)
    assert err <= 3e-7 and err <= 1.5 * floor_b + 2e-8, (err, floor_b)


@pytest.mark.parametrize("bias", [False, True])
@pytest.mark.parametrize("relu", [False, True])
@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("masked", [False, True])
def test_conv3x3_wino_every_flag(hip, bias, relu, accumulate, masked):
    n, cin, cout, H, W = 2, 64, 72, 34, 66
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
    bank_f, _ = wino.conv_pack_filters_wino(dev(w))
    y = wino.conv3x3_wino(dev(x), bank_f, dev(b) if bias else None, cout, 1, relu, out=dev(base.clone()),
                         out_relu_mask=dev(mask) if masked else None, accumulate=accumulate)
    torch.cuda.synchronize()
    assert rel_l2(y.cpu(), ref) <= 4e-7


@pytest.mark.parametrize("kind", ["wide_range", "tiny", "huge", "zeros", "one_hot", "hot_channel", "growing", "shrinking"])
def test_conv3x3_wino_scaling_survives_extreme_inputs(hip, kind):
    """The monotone power-of-two scale of the running sums: magnitudes that grow or shrink by many decades along the channels,
    values near the fp32 extremes, all-zero chunks, a single non-zero element; and exact homogeneity under powers of two."""
    cin, cout, H, W = 128, 64, 40, 48
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
    elif kind == "growing":
        x = x * torch.logspace(-12, 12, cin)[None, :, None, None]
    elif kind == "shrinking":
        x = x * torch.logspace(12, -12, cin)[None, :, None, None]
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    ref = F.conv2d(x.double(), w.double(), padding=1)
    bank_f, _ = wino.conv_pack_filters_wino(dev(w))
    y = wino.conv3x3_wino(dev(x), bank_f, None, cout, 1, False)
    y4 = wino.conv3x3_wino(dev(x * 4.0), bank_f, None, cout, 1, False)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    if kind == "zeros":
        assert float(y.abs().max()) == 0.0
    else:
        assert rel_l2(y.cpu(), ref) <= 4e-7
        if kind != "huge":
            assert torch.equal(y4, y * 4.0)
