"""Per-kernel parity of libmaua_hip.so (through the C ABI) against the CPU oracle's arithmetic.

fp32 tolerance: rel-L2 <= 2e-5 for contractions (the summation order differs from MKL-DNN's; K up to
9*512), exact equality where the op is a selection (pooling forward, ReLU).
"""
import math
import os
import tempfile

import pytest
import torch
import torch.nn.functional as F

from conftest import GOLDEN, PKG, REPO, rel_l2

pytestmark = pytest.mark.gpu

TOL = 2e-5


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


CONV_CASES = [
    # cin, cout, H, W, k, stride, pad
    (3, 64, 37, 45, 3, 1, 1),
    (64, 64, 32, 32, 3, 1, 1),
    (64, 128, 16, 24, 3, 1, 1),
    (128, 256, 8, 8, 3, 1, 1),
    (512, 512, 4, 4, 3, 1, 1),
    (256, 512, 2, 2, 3, 1, 1),
    (20, 40, 13, 70, 3, 1, 1),       # ragged channel counts (not multiples of the tiles)
    (64, 3, 33, 31, 3, 1, 1),        # narrow output (shape of conv1_1's backward-data)
    (96, 96, 30, 30, 1, 1, 0),
    (384, 1024, 7, 7, 1, 1, 0),
    (96, 256, 15, 17, 5, 1, 2),
    (3, 96, 99, 99, 11, 4, 0),
    (3, 96, 128, 128, 11, 4, 0),
    (3, 96, 203, 310, 11, 4, 0),     # strided stem kernels: width not a multiple of the stride, ragged last block
    (3, 32, 75, 1030, 11, 4, 0),     # ... more than one 256-thread block per row in the backward pass
    (8, 16, 9, 9, 3, 1, 0),          # no padding
    (3, 64, 70, 200, 3, 1, 1),       # few-output-channel kernel (backward) over several 62-column tiles
    (64, 4, 21, 130, 3, 1, 1),       # ... forward direction, 4 channels
    (32, 1, 40, 64, 3, 1, 1),        # ... 1 channel
    (2, 32, 19, 63, 3, 1, 0),        # ... 2 channels, no padding (backward pads by 2)
]


@pytest.mark.parametrize("cin,cout,H,W,k,stride,pad", CONV_CASES)
@pytest.mark.parametrize("relu", [False, True])
def test_conv_fwd(hip, cin, cout, H, W, k, stride, pad, relu):
    x = rnd(2, cin, H, W, seed=1)
    w = rnd(cout, cin, k, k, seed=2, scale=math.sqrt(2.0 / (k * k * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    ref = F.conv2d(x, w, b, stride=stride, padding=pad)
    if relu:
        ref = torch.relu(ref)
    wf, _ = hip.conv_pack_filters(dev(w))
    y = hip.conv2d_fwd(dev(x), wf, dev(b), k, stride, pad, relu)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert rel_l2(y.cpu(), ref) <= TOL


@pytest.mark.parametrize("cin,cout,H,W,k,stride,pad", CONV_CASES)
@pytest.mark.parametrize("masked", [False, True])
def test_conv_bwd_data(hip, cin, cout, H, W, k, stride, pad, masked):
    x = rnd(1, cin, H, W, seed=1)
    w = rnd(cout, cin, k, k, seed=2, scale=math.sqrt(2.0 / (k * k * cin)))
    y = torch.relu(F.conv2d(x, w, None, stride=stride, padding=pad))
    gy = rnd(*y.shape, seed=4)
    g_in = gy * (y > 0) if masked else gy
    ref = torch.nn.grad.conv2d_input(x.shape, w, g_in, stride=stride, padding=pad)
    _, wb = hip.conv_pack_filters(dev(w))
    gx = hip.conv2d_bwd_data(dev(gy), dev(y) if masked else None, wb, dev(w), x.shape, k, stride, pad)
    torch.cuda.synchronize()
    assert rel_l2(gx.cpu(), ref) <= TOL
    # accumulate flag adds into the destination
    base = rnd(*x.shape, seed=5)
    gx2 = hip.conv2d_bwd_data(dev(gy), dev(y) if masked else None, wb, dev(w), x.shape, k, stride, pad,
                              out=dev(base.clone()), accumulate=True)
    torch.cuda.synchronize()
    assert rel_l2(gx2.cpu(), ref + base) <= TOL


@pytest.mark.parametrize("cin,cout,H,W,k,pad", [(1024, 1024, 31, 31, 1, 0), (384, 96, 15, 17, 3, 1), (256, 96, 20, 24, 5, 2)])
def test_conv_fp32_split_k_matches_single_pass(hip, cin, cout, H, W, k, pad):
    """Deep layers on small maps split the channel loop over workgroups (deterministic two-stage sum)."""
    assert hip.conv_workspace_bytes(1, cin, H, W, cout, k, 1, pad) > 0
    x = rnd(1, cin, H, W, seed=1)
    w = rnd(cout, cin, k, k, seed=2, scale=math.sqrt(2.0 / (k * k * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    ref = torch.relu(F.conv2d(x, w, b, padding=pad))
    wf, wb = hip.conv_pack_filters(dev(w))
    none = torch.empty(0, dtype=torch.uint8, device="cuda")
    one = hip.conv2d_fwd(dev(x), wf, dev(b), k, 1, pad, True, workspace=none)
    two = hip.conv2d_fwd(dev(x), wf, dev(b), k, 1, pad, True)
    three = hip.conv2d_fwd(dev(x), wf, dev(b), k, 1, pad, True)
    torch.cuda.synchronize()
    assert rel_l2(one.cpu(), ref) <= TOL and rel_l2(two.cpu(), ref) <= TOL
    assert torch.equal(two, three)
    gy = rnd(*ref.shape, seed=4)
    refb = torch.nn.grad.conv2d_input(x.shape, w, gy, padding=pad)
    mask = rnd(*x.shape, seed=6)
    base = rnd(*x.shape, seed=7)
    gx = hip.conv2d_bwd_data(dev(gy), None, wb, dev(w), x.shape, k, 1, pad, out=dev(base.clone()), accumulate=True,
                             in_relu_mask=dev(mask))
    torch.cuda.synchronize()
    assert rel_l2(gx.cpu(), (refb + base) * (mask > 0)) <= TOL


def test_conv_rejects_bad_arguments(hip):
    x = dev(rnd(1, 3, 8, 8))
    wf, _ = hip.conv_pack_filters(dev(rnd(4, 3, 3, 3)))
    with pytest.raises(hip.HipError):
        hip.conv2d_fwd(x, wf, None, 3, 0, 1, False)  # stride 0
    with pytest.raises(hip.HipError):
        hip.conv2d_fwd(torch.zeros(1, 3, 8, 8), wf, None, 3, 1, 1, False)  # CPU tensor: no CPU path
    small = dev(rnd(1, 3, 2, 2))
    with pytest.raises(hip.HipError):
        hip.conv2d_fwd(small, wf, None, 3, 1, 0, False)  # input smaller than the filter


@pytest.mark.parametrize("C,H,W,k,s,ceil", [(64, 32, 32, 2, 2, False), (5, 7, 9, 2, 2, False), (96, 30, 30, 3, 2, True),
                                            (16, 31, 29, 3, 2, True), (8, 63, 63, 3, 2, True), (4, 3, 3, 3, 2, True),
                                            (3, 253, 253, 3, 2, True), (2, 127, 90, 3, 2, True), (2, 64, 33, 3, 2, False)])
@pytest.mark.parametrize("mode", ["max", "avg"])
def test_pool(hip, C, H, W, k, s, ceil, mode):
    x = torch.relu(rnd(2, C, H, W, seed=7))  # post-ReLU inputs: many exact ties at zero
    if mode == "max":
        ref, idx = F.max_pool2d(x, k, s, 0, ceil_mode=ceil, return_indices=True)
    else:
        ref = F.avg_pool2d(x, k, s, 0, ceil_mode=ceil)
    y = hip.pool2d_fwd(dev(x), k, s, ceil, mode)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    if mode == "max":
        assert torch.equal(y.cpu(), ref)
    else:
        assert rel_l2(y.cpu(), ref) <= 1e-6
    gy = rnd(*ref.shape, seed=8)
    xr = x.clone().requires_grad_(True)
    (F.max_pool2d(xr, k, s, 0, ceil_mode=ceil) if mode == "max" else F.avg_pool2d(xr, k, s, 0, ceil_mode=ceil)).backward(gy)
    gx = hip.pool2d_bwd(dev(gy), dev(x), k, s, ceil, mode)
    torch.cuda.synchronize()
    assert rel_l2(gx.cpu(), xr.grad) <= 1e-6


def test_relu(hip):
    x = rnd(3, 1000, seed=9)
    y = hip.relu_(dev(x.clone()))
    assert torch.equal(y.cpu(), torch.relu(x))
    gy = rnd(3, 1000, seed=10)
    gx = hip.relu_bwd(dev(gy), y)
    assert torch.equal(gx.cpu(), gy * (x > 0))


@pytest.mark.parametrize("C,HW", [(64, 4096), (64, 1000), (96, 900), (128, 77), (256, 256), (512, 64), (512, 4), (384, 3969),
                                  (1024, 961), (3, 5),
                                  # several tile pairs x several stages, ragged last stage, odd row pitch, ragged last tile
                                  (128, 16384), (256, 8100), (512, 4131), (192, 5000), (1000, 4200), (1024, 4096)])
@pytest.mark.parametrize("center", [False, True])
def test_gram_fwd_bwd(hip, C, HW, center):
    f = torch.relu(rnd(1, C, HW, 1, seed=11)) + (0.5 if center else 0.0)
    ff = f.reshape(C, HW)
    fc = ff - ff.mean(1, keepdim=True) if center else ff
    scale = 1.0 / (C * HW)
    ref = (fc.double() @ fc.double().t()) * scale
    gram, mean = hip.gram_fwd(dev(f), scale, center)
    torch.cuda.synchronize()
    assert rel_l2(gram.cpu(), ref) <= TOL
    assert torch.equal(gram.cpu(), gram.cpu().t())  # mirrored exactly
    if center:
        assert rel_l2(mean.cpu(), ff.mean(1)) <= 1e-6
    # backward: gf (+)= D (F - mean)
    d = rnd(C, C, seed=12)
    d = d + d.t()
    base = rnd(C, HW, seed=13)
    refb = d.double() @ fc.double()
    gf = hip.gram_bwd(dev(d), dev(f), mean, dev(torch.zeros(C, HW)), accumulate=False)
    torch.cuda.synchronize()
    assert rel_l2(gf.cpu(), refb) <= TOL
    gf = hip.gram_bwd(dev(d), dev(f), mean, dev(base.clone()), accumulate=True)
    torch.cuda.synchronize()
    assert rel_l2(gf.cpu(), refb + base.double()) <= TOL


@pytest.mark.parametrize("kind", ["wide_range", "tiny", "huge", "zeros", "one_hot", "big_mean"])
def test_gram_survives_extreme_inputs(hip, kind):
    """Rows spanning many decades, values near the fp32 extremes (squares stay finite), all-zero maps, one non-zero element,
    and (covariance form) a mean far above the spread; the result is exactly symmetric."""
    C, HW = 128, 4096
    g = torch.Generator().manual_seed(21)
    f = torch.randn(1, C, HW, 1, generator=g)
    center = False
    if kind == "wide_range":
        f = f * torch.exp(torch.randn(1, C, 1, 1, generator=g) * 4.0)
    elif kind == "tiny":
        f = f * 1e-15
    elif kind == "huge":
        f = f * 1e15
    elif kind == "zeros":
        f = torch.zeros_like(f)
    elif kind == "one_hot":
        f = torch.zeros_like(f)
        f[0, 5, 77, 0] = 3.0e-6
    elif kind == "big_mean":
        f, center = f + 50.0, True
    ff = f.reshape(C, HW).double()
    fc = ff - ff.mean(1, keepdim=True) if center else ff
    ref = fc @ fc.t() / (C * HW)
    gram, _ = hip.gram_fwd(dev(f), 1.0 / (C * HW), center)
    torch.cuda.synchronize()
    assert torch.isfinite(gram).all()
    if kind == "zeros":
        assert float(gram.abs().max()) == 0.0
    else:
        assert rel_l2(gram.cpu(), ref) <= (2e-5 if kind == "big_mean" else 2e-6)
    assert torch.equal(gram.cpu(), gram.cpu().t())


@pytest.mark.parametrize("C,HW,center", [(128, 4096, False), (256, 65536, False), (512, 16384, True), (1024, 4096, False), (512, 1024, False)])
def test_gram_deterministic(hip, C, HW, center):
    """The same bits on every launch, many workgroups per CU side by side."""
    f = dev(torch.relu(rnd(1, C, HW, 1, seed=14)))
    a, _ = hip.gram_fwd(f, 1e-3, center)
    for _ in range(12):
        b, _ = hip.gram_fwd(f, 1e-3, center)
        assert torch.equal(a, b)


@pytest.mark.parametrize("n", [1, 7, 4096, 512 * 512, 3 * 1000 * 1000 + 3])
def test_mse_fwd_bwd(hip, n):
    x, t = rnd(n, seed=15), rnd(n, seed=16)
    base = rnd(n, seed=17)
    loss = torch.zeros(1, device="cuda")
    g = dev(base.clone())
    hip.mse_fwd_bwd(dev(x), dev(t), g, 0.37 / n, 1.7, True, loss)
    torch.cuda.synchronize()
    ref = 0.37 * ((x.double() - t.double()) ** 2).mean()
    assert abs(loss.item() - ref.item()) <= 1e-6 * abs(ref.item())
    assert rel_l2(g.cpu(), base + 1.7 * (x - t)) <= 1e-6
    g2 = dev(torch.zeros(n))
    hip.mse_fwd_bwd(dev(x), dev(t), g2, 1.0, -2.0, False, loss)
    assert rel_l2(g2.cpu(), -2.0 * (x - t)) <= 1e-6


@pytest.mark.parametrize("shape", [(1, 3, 32, 32), (1, 3, 1, 9), (2, 3, 17, 5), (1, 3, 257, 300)])
def test_tv(hip, shape):
    x = rnd(*shape, seed=18)
    x[0, 0, 0, :2] = 0.25  # an exact tie: sign(0) = 0 on both sides
    xr = x.clone().requires_grad_(True)
    ref = 0.02 * ((xr[:, :, 1:, :] - xr[:, :, :-1, :]).abs().sum() + (xr[:, :, :, 1:] - xr[:, :, :, :-1]).abs().sum())
    ref.backward()
    loss = torch.zeros(1, device="cuda")
    g = dev(torch.zeros(*shape))
    hip.tv_fwd_bwd(dev(x), g, 0.02, False, loss)
    torch.cuda.synchronize()
    assert abs(loss.item() - ref.item()) <= 1e-5 * abs(ref.item())
    assert rel_l2(g.cpu(), xr.grad) <= 1e-6


def test_adam_matches_torch(hip):
    n = 10007
    x0 = rnd(n, seed=19)
    p = torch.nn.Parameter(x0.clone())
    opt = torch.optim.Adam([p], lr=1.0)
    x = dev(x0.clone())
    m, v = dev(torch.zeros(n)), dev(torch.zeros(n))
    for step in range(1, 8):
        g = rnd(n, seed=100 + step) * (1.0 + step)
        p.grad = g.clone()
        opt.step()
        hip.adam_step(x, dev(g), m, v, step, 1.0)
    torch.cuda.synchronize()
    assert rel_l2(x.cpu(), p.detach()) <= 1e-6


def _toy_problem(n, seed, sigma=0.1):
    """Smooth non-quadratic objective with its minimiser at 0, evaluated with torch ops on x's device.

    fp32 L-BFGS loses digits in y = g - g_prev whenever the step is tiny compared with the curvature scale
    (SURVEY.md §7 'catastrophic cancellation'); starting with |x0|_1 ~ 1 and sigma*|H| ~ 0.3 keeps |y| ~ |g|, so
    this test measures the algorithm (ring, two-loop, step rules) and not that noise."""
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(n, generator=g, dtype=torch.float64) * 3 + 0.5
    q = 0.1 * n * n

    def fg(x):
        aa = a.to(x.device, x.dtype)
        r = torch.roll(x, 1)
        loss = sigma * ((aa * x * x).sum() + q * (x ** 4).sum() + 0.5 * (x * r).sum())
        grad = sigma * (2 * aa * x + 4 * q * x ** 3 + 0.5 * (r + torch.roll(x, -1)))
        return float(loss), grad
    return fg


@pytest.mark.parametrize("n,history,iters", [(1000, 100, 30), (5000, 5, 40), (3 * 64 * 64 + 1, 100, 25), (40000, 7, 12)])
def test_lbfgs_matches_oracle(hip, n, history, iters):
    """Device L-BFGS (coefficient-space two-loop) vs the oracle's restatement of torch.optim.LBFGS in fp64."""
    from oracle import lbfgs_run
    fg = _toy_problem(n, seed=5)
    gen = torch.Generator().manual_seed(6)
    x0 = torch.randn(n, generator=gen, dtype=torch.float64) * (1.25 / n)  # |x0|_1 ~ 1
    x0 = x0.float().double()
    trace, stats = [], {}
    ref, _ = lbfgs_run(fg, x0, iters, history=history, trace=trace, stats=stats)
    x = dev(x0.float())
    st = hip.LbfgsState(n, history, x.device)
    for it in range(iters):
        _, g = fg(x)
        st.iterate(x, g.contiguous())
        if it in (0, 1, 2, 3, 5, 8, 11):
            torch.cuda.synchronize()
            assert rel_l2(x.cpu(), trace[it]) <= 2e-4, f"move {it}"
    torch.cuda.synchronize()
    s = st.status()
    # pairs with y.s <= 1e-10 are skipped by both (late, tiny steps); allow the borderline ones to differ
    assert s["n_iter"] == iters and abs(s["history_len"] - stats["history_len"]) <= 2 and not s["stopped"]
    assert s["history_len"] <= history
    assert float((x.cpu().double() - ref).norm() / x0.norm()) <= 1e-5


def test_lbfgs_far_start_converges_like_the_oracle(hip):
    """Far from the minimiser the first step is t = 1/|g|_1 << 1 and fp32 cancellation dominates the early moves;
    both runs must still land on the same point."""
    from oracle import lbfgs_run
    n, iters = 6000, 60
    fg = _toy_problem(n, seed=7, sigma=1.0)
    x0 = torch.linspace(-2, 2, n, dtype=torch.float64) / n
    ref, _ = lbfgs_run(fg, x0, iters)
    x = dev(x0.float())
    st = hip.LbfgsState(n, 100, x.device)
    for it in range(iters):
        st.iterate(x, fg(x)[1].contiguous())
    torch.cuda.synchronize()
    assert float((x.cpu().double() - ref).norm() / x0.norm()) <= 1e-3
    assert float(x.cpu().double().norm() / x0.norm()) <= 1e-3  # converged towards the minimiser


def test_lbfgs_stop_flag_and_first_step(hip):
    """g.d > -tolerance_change stops the update (reference `break`), and the first step is min(1, 1/|g|_1)."""
    n = 512
    x0 = rnd(n, seed=21)
    x = dev(x0.clone())
    g = rnd(n, seed=22)
    st = hip.LbfgsState(n, 100, x.device)
    st.iterate(x, dev(g))
    torch.cuda.synchronize()
    t = min(1.0, 1.0 / float(g.abs().sum()))
    assert rel_l2(x.cpu(), x0 - t * g) <= 1e-6
    s = st.status()
    assert abs(s["t"] - t) <= 1e-6 * t and abs(s["gtd"] + float(g.dot(g))) <= 1e-4 * float(g.dot(g))
    # a huge tolerance_change forces the stop branch on the next call
    before = x.clone()
    st.iterate(x, dev(rnd(n, seed=23)), tolerance_change=1e30)
    torch.cuda.synchronize()
    assert st.status()["stopped"] and torch.equal(x, before)


@pytest.mark.parametrize("which", ["grad", "step", "loss"])
def test_lbfgs_tolerances_stop_where_the_oracle_stops(hip, which):
    """--lbfgs_tolerance_grad / --lbfgs_tolerance_change (reference optim.py:183-188 hands them to torch.optim.LBFGS): the
    max|g|, max|t d| and |loss - prev_loss| tests run on the device and end the run at the oracle's move count."""
    from oracle import lbfgs_run
    n, iters = 2000, 60
    fg = _toy_problem(n, seed=9)
    x0 = (torch.randn(n, generator=torch.Generator().manual_seed(10), dtype=torch.float64) * (1.25 / n)).float().double()
    # thresholds met part of the way through the unconstrained 60-iteration run of this problem
    trace = []
    lbfgs_run(fg, x0, iters, trace=trace)
    xs = [x0] + trace
    k = 12
    if which == "grad":
        tg, tc = float(fg(xs[k])[1].abs().max()) * 1.0001, -1.0
    elif which == "step":
        tg, tc = -1.0, float((xs[k] - xs[k - 1]).abs().max()) * 1.0001
    else:
        tg, tc = -1.0, abs(fg(xs[k])[0] - fg(xs[k - 1])[0]) * 1.0001
    trace2, stats = [], {}
    ref, _ = lbfgs_run(fg, x0, iters, tol_grad=tg, tol_change=tc, trace=trace2, stats=stats)
    assert 3 <= len(trace2) < iters  # the oracle really stopped early
    x = dev(x0.float())
    st = hip.LbfgsState(n, 100, x.device)
    moves = 0
    for it in range(iters):
        loss, g = fg(x)
        st.iterate(x, g.contiguous(), 1.0, tc, tg, loss=torch.tensor([loss], dtype=torch.float32, device=x.device))
        if st.status()["stopped"]:
            break
        moves += 1
    assert abs(moves - len(trace2)) <= 1, (moves, len(trace2))  # a threshold 1e-4 off the boundary: fp32 may land either side
    if moves == len(trace2):
        assert float((x.cpu().double() - ref).norm() / x0.norm()) <= 1e-4


def test_lbfgs_history_limit_is_refused(hip):
    with pytest.raises(hip.HipError):
        hip.LbfgsState(1000, 255, torch.device("cuda"))
    hip.LbfgsState(1000, 254, torch.device("cuda"))


# ---------------------------------------------------------------------------------------------------------
# bf16x6 convolution (fp32 accuracy on the bf16 matrix cores) and producer-side ReLU masks
# ---------------------------------------------------------------------------------------------------------
X6_CASES = [(3, 64, 37, 45, 1), (64, 64, 32, 32, 1), (64, 128, 16, 24, 1), (128, 256, 8, 8, 1), (512, 512, 4, 4, 1),
            (256, 512, 2, 2, 1), (20, 40, 13, 70, 1), (96, 70, 9, 33, 1), (8, 16, 9, 9, 0), (384, 1024, 7, 7, 1)]


@pytest.mark.parametrize("cin,cout,H,W,pad", X6_CASES)
def test_conv3x3_x6_forward_and_backward(hip, cin, cout, H, W, pad):
    x = rnd(2, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=pad))
    bank_f, bank_b = hip.conv_pack_filters_x6(dev(w))
    y = hip.conv3x3_x6(dev(x), bank_f, dev(b), cout, pad, True)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert rel_l2(y.cpu(), ref) <= 2e-6  # fp32-level: the fp32 CPU conv itself is ~3e-7 from fp64
    gy = rnd(*ref.shape, seed=4)
    refb = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=pad)
    mask = rnd(*x.shape, seed=6)  # "ReLU output" of the layer below: its sign decides what survives
    gx = hip.conv3x3_x6(dev(gy), bank_b, None, cin, 2 - pad, False, out_relu_mask=dev(mask))
    torch.cuda.synchronize()
    assert rel_l2(gx.cpu(), refb * (mask > 0)) <= 2e-6
    base = rnd(*x.shape, seed=5)
    gx2 = hip.conv3x3_x6(dev(gy), bank_b, None, cin, 2 - pad, False, out=dev(base.clone()), accumulate=True)
    torch.cuda.synchronize()
    assert rel_l2(gx2.cpu(), refb + base.double()) <= 2e-6


@pytest.mark.parametrize("cin,cout,H,W", [(512, 512, 16, 16), (256, 256, 33, 40), (512, 64, 8, 8)])
def test_conv3x3_x6_split_k_matches_single_pass(hip, cin, cout, H, W):
    # small output grids split the channel loop over workgroups; both forms are the same convolution
    assert hip.conv_x6_workspace_bytes(1, cin, H, W, cout, 1) > 0
    assert hip.conv_x6_workspace_bytes(1, 64, 1024, 1024, 64, 1) == 0
    x = rnd(1, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    base, mask = rnd(1, cout, H, W, seed=5), rnd(1, cout, H, W, seed=6)
    ref = (torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1) + base.double())) * (mask > 0)
    bank_f, _ = hip.conv_pack_filters_x6(dev(w))
    none = torch.empty(0, dtype=torch.uint8, device="cuda")
    one = hip.conv3x3_x6(dev(x), bank_f, dev(b), cout, 1, True, out=dev(base.clone()), accumulate=True, out_relu_mask=dev(mask),
                         workspace=none)
    split = hip.conv3x3_x6(dev(x), bank_f, dev(b), cout, 1, True, out=dev(base.clone()), accumulate=True,
                           out_relu_mask=dev(mask))
    split2 = hip.conv3x3_x6(dev(x), bank_f, dev(b), cout, 1, True, out=dev(base.clone()), accumulate=True,
                            out_relu_mask=dev(mask))
    torch.cuda.synchronize()
    assert rel_l2(one.cpu(), ref) <= 2e-6 and rel_l2(split.cpu(), ref) <= 2e-6
    assert rel_l2(split.cpu(), one.cpu().double()) <= 5e-7
    assert torch.equal(split, split2)


# ---------------------------------------------------------------------------------------------------------
# fp16x3 convolution (two-part fp16 split with per-workgroup, per-chunk scaling): same cases and bounds as bf16x6
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,H,W,pad", X6_CASES)
def test_conv3x3_x3_forward_and_backward(hip, cin, cout, H, W, pad):
    x = rnd(2, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=pad))
    bank_f, bank_b, wsc = hip.conv_pack_filters_x3(dev(w))
    assert math.log2(wsc) == int(math.log2(wsc)) and 32 <= float(w.abs().max()) * wsc < 64
    y = hip.conv3x3_x3(dev(x), bank_f, wsc, dev(b), cout, pad, True)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert rel_l2(y.cpu(), ref) <= 2e-6
    gy = rnd(*ref.shape, seed=4)
    refb = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=pad)
    mask = rnd(*x.shape, seed=6)
    gx = hip.conv3x3_x3(dev(gy), bank_b, wsc, None, cin, 2 - pad, False, out_relu_mask=dev(mask))
    torch.cuda.synchronize()
    assert rel_l2(gx.cpu(), refb * (mask > 0)) <= 2e-6
    base = rnd(*x.shape, seed=5)
    gx2 = hip.conv3x3_x3(dev(gy), bank_b, wsc, None, cin, 2 - pad, False, out=dev(base.clone()), accumulate=True)
    gx3 = hip.conv3x3_x3(dev(gy), bank_b, wsc, None, cin, 2 - pad, False, out=dev(base.clone()), accumulate=True)
    torch.cuda.synchronize()
    assert rel_l2(gx2.cpu(), refb + base.double()) <= 2e-6
    assert torch.equal(gx2, gx3)


@pytest.mark.parametrize("cin,cout,H,W", [(512, 512, 16, 16), (256, 256, 33, 40), (512, 64, 8, 8)])
def test_conv3x3_x3_split_k_matches_single_pass(hip, cin, cout, H, W):
    assert hip.conv_x3_workspace_bytes(1, cin, H, W, cout, 1) > 0
    assert hip.conv_x3_workspace_bytes(1, 64, 1024, 1024, 64, 1) == 0
    x = rnd(1, cin, H, W, seed=1)
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    base, mask = rnd(1, cout, H, W, seed=5), rnd(1, cout, H, W, seed=6)
    ref = (torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1) + base.double())) * (mask > 0)
    bank_f, _, wsc = hip.conv_pack_filters_x3(dev(w))
    none = torch.empty(0, dtype=torch.uint8, device="cuda")
    one = hip.conv3x3_x3(dev(x), bank_f, wsc, dev(b), cout, 1, True, out=dev(base.clone()), accumulate=True,
                         out_relu_mask=dev(mask), workspace=none)
    split = hip.conv3x3_x3(dev(x), bank_f, wsc, dev(b), cout, 1, True, out=dev(base.clone()), accumulate=True,
                           out_relu_mask=dev(mask))
    torch.cuda.synchronize()
    assert rel_l2(one.cpu(), ref) <= 2e-6 and rel_l2(split.cpu(), ref) <= 2e-6
    assert rel_l2(split.cpu(), one.cpu().double()) <= 5e-7


@pytest.mark.parametrize("kind", ["wide_range", "tiny", "huge", "zeros", "one_hot"])
def test_conv3x3_x3_scaling_survives_extreme_inputs(hip, kind):
    """fp16 has 5 exponent bits: the per-chunk power-of-two scaling must keep every magnitude usable.  Gradients spanning
    ten decades, values near the fp32 extremes, all-zero tiles and a single non-zero element."""
    cin, cout, H, W = 64, 64, 40, 40
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
    w = rnd(cout, cin, 3, 3, seed=2, scale=math.sqrt(2.0 / (9 * cin)))
    ref = F.conv2d(x.double(), w.double(), padding=1)
    bank_f, _, wsc = hip.conv_pack_filters_x3(dev(w))
    y = hip.conv3x3_x3(dev(x), bank_f, wsc, None, cout, 1, False)
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    if kind == "zeros":
        assert float(y.abs().max()) == 0.0
    else:
        assert rel_l2(y.cpu(), ref) <= 2e-6


# ---------------------------------------------------------------------------------------------------------
# fp16x3 1x1 convolution / channel-mixing product (both operands split in the kernel): NIN's 1x1 layers, Gram backward
# ---------------------------------------------------------------------------------------------------------
P1_CASES = [(96, 96, 40, 41), (256, 256, 27, 27), (100, 70, 9, 37), (64, 64, 64, 65), (1024, 1000, 6, 7), (32, 128, 1, 5),
            (384, 384, 13, 13), (8, 8, 16, 16)]


@pytest.mark.parametrize("cin,cout,H,W", P1_CASES)
def test_conv1x1_x3_forward_and_backward(hip, cin, cout, H, W):
    x = rnd(2, cin, H, W, seed=1)
    w = rnd(cout, cin, seed=2, scale=math.sqrt(2.0 / cin))
    b = rnd(cout, seed=3, scale=0.1)
    ref = torch.relu(torch.einsum("oc,nchw->nohw", w.double(), x.double()) + b.double()[None, :, None, None])
    y = hip.conv1x1_x3(dev(x), dev(w), dev(b), relu=True)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert rel_l2(y.cpu(), ref) <= 2e-6
    gy = rnd(*ref.shape, seed=4)
    refb = torch.einsum("oc,nohw->nchw", w.double(), gy.double())
    mask, base = rnd(*x.shape, seed=6), rnd(*x.shape, seed=5)
    wt = dev(w.t().contiguous())
    gx = hip.conv1x1_x3(dev(gy), wt, out_relu_mask=dev(mask))
    torch.cuda.synchronize()
    assert rel_l2(gx.cpu(), refb * (mask > 0)) <= 2e-6
    gx2 = hip.conv1x1_x3(dev(gy), wt, out=dev(base.clone()), accumulate=True)
    gx3 = hip.conv1x1_x3(dev(gy), wt, out=dev(base.clone()), accumulate=True)
    torch.cuda.synchronize()
    assert rel_l2(gx2.cpu(), refb + base.double()) <= 2e-6
    assert torch.equal(gx2, gx3)
    # the order in which the workgroups walk the (pixel tile, channel tile) items (round 5: XCD bands; planner field p1_order) changes no bit
    L = hip.lib()
    try:
        assert L.maua_set_tuning(b"p1_order", 0.0) == 0
        y_plain = hip.conv1x1_x3(dev(x), dev(w), dev(b), relu=True)
    finally:
        L.maua_set_tuning(b"p1_order", 1.0)
    assert torch.equal(y, y_plain)


@pytest.mark.parametrize("cin,cout,hw", [(128, 128, 4096), (100, 100, 777), (512, 512, 1024)])
def test_conv1x1_x3_channel_shift_is_the_gram_centring(hip, cin, cout, hw):
    """y = W (x - shift) with the shift applied before the split: features with a mean well above their spread (post-ReLU
    maps) keep full accuracy - the subtraction is exact in fp32 terms and only the centred values are split."""
    x = torch.relu(rnd(1, cin, hw, seed=1)) + 5.0
    shift = x[0].mean(1)
    w = rnd(cout, cin, seed=2)
    w = w + w.t() if cin == cout else w
    ref = torch.einsum("oc,ncp->nop", w.double(), (x - shift[None, :, None]).double())
    y = hip.conv1x1_x3(dev(x), dev(w), x_shift=dev(shift))
    torch.cuda.synchronize()
    assert rel_l2(y.cpu(), ref) <= 2e-6


def test_gram_128_blocks_stress_and_the_64_route_subprocess():
    """Layers of 128+ channels and 1024+ pixels in whole 64-pixel stages multiply in 128 x 128 blocks (planner field gram_t128, set per
    process: 0 = 64 x 64 everywhere, 2 = ragged maps too).  tools/stress_gram.py under the three settings: every layer set against fp64 (<= 2e-5 with the means of the covariance sets), the batched
    launches bit for bit what the per-layer launches leave, and the same bits on every one of 12 launches between LDS-scribbling
    convolutions (the first form of the 128 x 128 kernel passed every single-launch test and failed this one: probes_r04.md section 2)."""
    import subprocess
    import sys
    code = ("import sys; sys.path[:0] = [%r, %r]; import hip\n"
            "print('blocks', hip.gram_block(128, 4096), hip.gram_block(512, 16384), hip.gram_block(256, 8100), hip.gram_block(512, 484), hip.gram_block(64, 1 << 20))\n") % (REPO, PKG)
    for flag, want in (("1", "blocks 128 128 64 64 64"), ("2", "blocks 128 128 128 64 64"), ("0", "blocks 64 64 64 64 64")):
        env = dict(os.environ, MAUA_PLAN="gram_t128=" + flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and want in r.stdout, (r.stdout[-500:], r.stderr[-1500:])
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "stress_gram.py"), "12"], env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0 and "differing results: 0" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])


def test_gram_bwd_fp32_route_subprocess():
    """gram_bwd_x3=0 (MAUA_PLAN) keeps the fp32-MFMA Gram backward; both routes agree to fp32 level."""
    import subprocess
    import sys
    code = (
        "import sys, torch; sys.path[:0] = [%r, %r]; import hip\n"
        "g = torch.Generator().manual_seed(3)\n"
        "f = torch.relu(torch.randn(1, 128, 64, 72, generator=g)).cuda(); d = torch.randn(128, 128, generator=g); d = (d + d.t()).cuda()\n"
        "gram, mean = hip.gram_fwd(f, 1e-3, True)\n"
        "gf = hip.gram_bwd(d, f, mean, torch.zeros(128, 64 * 72, device='cuda'), False, relu_mask=f); torch.cuda.synchronize()\n"
        "torch.save(gf.cpu(), sys.argv[1])\n") % (REPO, PKG)
    outs = []
    for flag in ("0", "1"):
        path = os.path.join(tempfile.mkdtemp(), "gf.pt")
        env = dict(os.environ, MAUA_PLAN="gram_bwd_x3=" + flag)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(torch.load(path))
    assert not torch.equal(outs[0], outs[1])  # different arithmetic really ran
    assert rel_l2(outs[1], outs[0].double()) <= 2e-6


@pytest.mark.parametrize("cin,cout,hw", [(512, 512, 4096), (1024, 1024, 36), (256, 64, 1000)])
def test_conv1x1_x3_split_k_matches_single_pass(hip, cin, cout, hw):
    assert hip.conv1x1_x3_workspace_bytes(1, cin, hw, cout) > 0
    assert hip.conv1x1_x3_workspace_bytes(1, 64, 1 << 20, 64) == 0
    x = rnd(1, cin, hw, seed=1)
    w = rnd(cout, cin, seed=2, scale=math.sqrt(2.0 / cin))
    b = rnd(cout, seed=3, scale=0.1)
    base, mask = rnd(1, cout, hw, seed=5), rnd(1, cout, hw, seed=6)
    ref = torch.relu(torch.einsum("oc,ncp->nop", w.double(), x.double()) + b.double()[None, :, None] + base.double()) * (mask > 0)
    none = torch.empty(0, dtype=torch.uint8, device="cuda")
    one = hip.conv1x1_x3(dev(x), dev(w), dev(b), relu=True, out=dev(base.clone()), accumulate=True, out_relu_mask=dev(mask),
                         workspace=none)
    split = hip.conv1x1_x3(dev(x), dev(w), dev(b), relu=True, out=dev(base.clone()), accumulate=True, out_relu_mask=dev(mask))
    torch.cuda.synchronize()
    assert rel_l2(one.cpu(), ref) <= 2e-6 and rel_l2(split.cpu(), ref) <= 2e-6
    assert rel_l2(split.cpu(), one.cpu().double()) <= 5e-7


@pytest.mark.parametrize("kind", ["wide_range", "tiny", "huge", "zeros", "one_hot", "wide_weights"])
def test_conv1x1_x3_scaling_survives_extreme_inputs(hip, kind):
    cin, cout, hw = 128, 128, 1600
    g = torch.Generator().manual_seed(9)
    x = torch.randn(1, cin, hw, generator=g)
    w = rnd(cout, cin, seed=2, scale=math.sqrt(2.0 / cin))
    if kind == "wide_range":
        x = x * torch.exp(torch.randn(1, cin, hw, generator=g) * 4.0) * 1e-6 * (torch.rand(1, cin, hw, generator=g) > 0.5)
    elif kind == "tiny":
        x = x * 1e-30
    elif kind == "huge":
        x = x * 1e30
    elif kind == "zeros":
        x = torch.zeros_like(x)
    elif kind == "one_hot":
        x = torch.zeros_like(x)
        x[0, 17, 420] = 3.0e-12
    elif kind == "wide_weights":  # a Gram difference whose rows span many decades (fresh style targets vs a noise image)
        w = w * torch.exp(torch.randn(cout, 1, generator=g) * 5.0) * 1e-4
    ref = torch.einsum("oc,ncp->nop", w.double(), x.double())
    y = hip.conv1x1_x3(dev(x), dev(w))
    torch.cuda.synchronize()
    assert torch.isfinite(y).all()
    if kind == "zeros":
        assert float(y.abs().max()) == 0.0
    else:
        assert rel_l2(y.cpu(), ref) <= 2e-6


# ---------------------------------------------------------------------------------------------------------
# fp16x3 k x k convolution (5x5: NIN's conv2), forward and backward-data
# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,H,W,pad", [(96, 256, 31, 29, 2), (40, 70, 9, 37, 2), (256, 96, 20, 33, 2), (16, 64, 12, 12, 0),
                                              (8, 40, 5, 70, 4)])
def test_conv5x5_x3_forward_and_backward(hip, cin, cout, H, W, pad):
    x = rnd(2, cin, H, W, seed=1)
    w = rnd(cout, cin, 5, 5, seed=2, scale=math.sqrt(2.0 / (25 * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=pad))
    bank_f, bank_b, wsc = hip.conv_pack_filters_kxk_x3(dev(w))
    assert 32 <= float(w.abs().max()) * wsc < 64
    y = hip.conv_kxk_x3(dev(x), bank_f, wsc, dev(b), cout, 5, pad, True)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    assert rel_l2(y.cpu(), ref) <= 2e-6
    gy = rnd(*ref.shape, seed=4)
    refb = torch.nn.grad.conv2d_input(x.shape, w.double(), gy.double(), padding=pad)
    mask, base = rnd(*x.shape, seed=6), rnd(*x.shape, seed=5)
    gx = hip.conv_kxk_x3(dev(gy), bank_b, wsc, None, cin, 5, 4 - pad, False, out_relu_mask=dev(mask))
    torch.cuda.synchronize()
    assert gx.shape == x.shape
    assert rel_l2(gx.cpu(), refb * (mask > 0)) <= 2e-6
    gx2 = hip.conv_kxk_x3(dev(gy), bank_b, wsc, None, cin, 5, 4 - pad, False, out=dev(base.clone()), accumulate=True)
    gx3 = hip.conv_kxk_x3(dev(gy), bank_b, wsc, None, cin, 5, 4 - pad, False, out=dev(base.clone()), accumulate=True)
    torch.cuda.synchronize()
    assert rel_l2(gx2.cpu(), refb + base.double()) <= 2e-6
    assert torch.equal(gx2, gx3)


def test_conv5x5_x3_split_k_matches_single_pass(hip):
    """Backward geometry of NIN's conv2 scaled down: few output tiles -> the channel loop is split and finished in order."""
    cin, cout, H, W = 256, 96, 24, 30
    assert hip.conv_kxk_x3_workspace_bytes(1, cin, H, W, cout, 5, 2) > 0
    assert hip.conv_kxk_x3_workspace_bytes(1, 96, 126, 126, 256, 5, 2) == 0
    x = rnd(1, cin, H, W, seed=1)
    w = rnd(cout, cin, 5, 5, seed=2, scale=math.sqrt(2.0 / (25 * cin)))
    b = rnd(cout, seed=3, scale=0.1)
    base, mask = rnd(1, cout, H, W, seed=5), rnd(1, cout, H, W, seed=6)
    ref = (torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=2) + base.double())) * (mask > 0)
    bank_f, _, wsc = hip.conv_pack_filters_kxk_x3(dev(w))
    none = torch.empty(0, dtype=torch.uint8, device="cuda")
    one = hip.conv_kxk_x3(dev(x), bank_f, wsc, dev(b), cout, 5, 2, True, out=dev(base.clone()), accumulate=True,
                          out_relu_mask=dev(mask), workspace=none)
    split = hip.conv_kxk_x3(dev(x), bank_f, wsc, dev(b), cout, 5, 2, True, out=dev(base.clone()), accumulate=True,
                            out_relu_mask=dev(mask))
    torch.cuda.synchronize()
    assert rel_l2(one.cpu(), ref) <= 2e-6 and rel_l2(split.cpu(), ref) <= 2e-6
    assert rel_l2(split.cpu(), one.cpu().double()) <= 5e-7


def test_conv5x5_x3_rejects_other_filter_sizes(hip):
    x = dev(rnd(1, 8, 16, 16))
    bank = torch.empty(1 << 20, dtype=torch.uint8, device="cuda")
    with pytest.raises(hip.HipError):
        hip.conv_kxk_x3(x, bank, 1.0, None, 64, 7, 3, False)


def test_conv3x3_x6_persistent_workgroups_subprocess():
    """x6_persist=1 (MAUA_PLAN): several tiles per workgroup with cross-tile prefetch, in-loop epilogue
    and accumulator re-initialisation must give the same bits as one workgroup per tile."""
    import subprocess
    import sys
    code = (
        "import sys, torch, math; sys.path[:0] = [%r, %r]; import hip\n"
        "g = torch.Generator().manual_seed(3)\n"
        "x = torch.randn(1, 64, 300, 260, generator=g).cuda(); w = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda()\n"
        "b = torch.randn(64, generator=g).cuda(); f6, _ = hip.conv_pack_filters_x6(w)\n"
        "y = hip.conv3x3_x6(x, f6, b, 64, 1, True); torch.cuda.synchronize()\n"
        "torch.save(y.cpu(), sys.argv[1])\n") % (REPO, PKG)
    outs = []
    for flag in ("0", "1"):
        path = os.path.join(tempfile.mkdtemp(), "y.pt")
        env = dict(os.environ, MAUA_PLAN="x6_persist=" + flag)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(torch.load(path))
    assert torch.equal(outs[0], outs[1])


def test_conv3x3_x6_is_deterministic(hip):
    x, w = dev(rnd(1, 64, 40, 40, seed=1)), dev(rnd(128, 64, 3, 3, seed=2, scale=0.05))
    bank_f, _ = hip.conv_pack_filters_x6(w)
    a = hip.conv3x3_x6(x, bank_f, None, 128, 1, False)
    b = hip.conv3x3_x6(x, bank_f, None, 128, 1, False)
    torch.cuda.synchronize()
    assert torch.equal(a, b)


@pytest.mark.parametrize("cin,cout,H,W,k,stride,pad", [(64, 64, 20, 20, 3, 1, 1), (40, 24, 9, 31, 1, 1, 0), (3, 16, 31, 31, 11, 4, 0)])
def test_conv_bwd_producer_side_relu_mask(hip, cin, cout, H, W, k, stride, pad):
    x = rnd(1, cin, H, W, seed=1)
    w = rnd(cout, cin, k, k, seed=2, scale=math.sqrt(2.0 / (k * k * cin)))
    oh, ow = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    gy = rnd(1, cout, oh, ow, seed=4)
    ref = torch.nn.grad.conv2d_input(x.shape, w, gy, stride=stride, padding=pad) * (x > 0)
    _, wb = hip.conv_pack_filters(dev(w))
    gx = hip.conv2d_bwd_data(dev(gy), None, wb, dev(w), x.shape, k, stride, pad, in_relu_mask=dev(x))
    torch.cuda.synchronize()
    assert rel_l2(gx.cpu(), ref) <= TOL


def test_pool3x3s2_backward_with_relu_mask_equals_unmasked_times_mask(hip):
    """The LDS-tiled 3x3 stride-2 kernel: bit-equal to gradient * (x > 0), and bit-reproducible (gather form, no atomics)."""
    x = torch.relu(rnd(2, 6, 127, 125, seed=7))
    gy = rnd(2, 6, 63, 62, seed=8)
    plain = hip.pool2d_bwd(dev(gy), dev(x), 3, 2, True, "max")
    masked = hip.pool2d_bwd(dev(gy), dev(x), 3, 2, True, "max", relu_mask_by_x=True)
    again = hip.pool2d_bwd(dev(gy), dev(x), 3, 2, True, "max", relu_mask_by_x=True)
    torch.cuda.synchronize()
    assert torch.equal(masked.cpu(), plain.cpu() * (x > 0))
    assert torch.equal(masked, again)
    xr = x.clone().requires_grad_(True)
    F.max_pool2d(xr, 3, 2, 0, ceil_mode=True).backward(gy)
    assert rel_l2(plain.cpu(), xr.grad) <= 1e-6


def test_masks_on_pool_mse_gram_backward(hip):
    x = torch.relu(rnd(1, 8, 12, 12, seed=7))
    gy = rnd(1, 8, 6, 6, seed=8)
    xr = x.clone().requires_grad_(True)
    F.max_pool2d(xr, 2, 2).backward(gy)
    gx = hip.pool2d_bwd(dev(gy), dev(x), 2, 2, False, "max", relu_mask_by_x=True)
    assert rel_l2(gx.cpu(), xr.grad * (x > 0)) <= 1e-6
    n = 5000
    f, t, base = torch.relu(rnd(n, seed=1)), rnd(n, seed=2), rnd(n, seed=3)
    g = dev(base.clone())
    loss = torch.zeros(1, device="cuda")
    hip.mse_fwd_bwd(dev(f), dev(t), g, 1.0 / n, 0.5, True, loss, mask_grad_by_x=True)
    assert rel_l2(g.cpu(), (base + 0.5 * (f - t)) * (f > 0)) <= 1e-6
    C, HW = 96, 500
    fm = torch.relu(rnd(1, C, HW, 1, seed=4))
    d = rnd(C, C, seed=5)
    d = d + d.t()
    base = rnd(C, HW, seed=6)
    gf = hip.gram_bwd(dev(d), dev(fm), None, dev(base.clone()), True, relu_mask=dev(fm))
    torch.cuda.synchronize()
    ref = (d.double() @ fm.reshape(C, HW).double() + base.double()) * (fm.reshape(C, HW) > 0)
    assert rel_l2(gf.cpu(), ref) <= TOL


# ---------------------------------------------------------------------------------------------------------
# image-space steps between two optimisation runs (csrc/image.hip): colour transfer, bilinear resize, deprocess
# ---------------------------------------------------------------------------------------------------------
def _hist_fixture():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "match_histogram.npz"))


@pytest.mark.parametrize("on_device", [True, False])
def test_match_histogram_device_vs_reference_fixture(hip, on_device):
    """utils.match_histogram on the MI355X against outputs of the reference's own function (tests/golden/match_histogram.npz,
    tools/make_golden.py::gen_hist): same seeds -> same jitter -> same result to 1e-5; device targets stay on the device,
    CPU targets come back on the CPU."""
    import numpy as np
    import utils
    g = _hist_fixture()
    put = (lambda t: t.cuda()) if on_device else (lambda t: t)
    target, src1, src2 = (torch.from_numpy(g[k]) for k in ("target", "src1", "src2"))
    cases = []
    for tag, srcs in (("one", [src1]), ("two", [src1, src2])):
        cases += [(target, srcs, True, f"out_{tag}", 1234, None), (target, srcs, "avg", f"out_avg_{tag}", 1234, None)]
    clip, vsrc = torch.from_numpy(g["clip"]), torch.from_numpy(g["vsrc"])
    cases += [(clip, [vsrc, src2], "avg", "out_clip_avg", 77, 5), (clip, [vsrc, src2], True, "out_clip_rand", 77, 5),
              (torch.from_numpy(g["big"]), [src1], True, "out_big", 99, None)]
    for tgt, srcs, mode, key, seed, npseed in cases:
        torch.manual_seed(seed)
        if npseed is not None:
            np.random.seed(npseed)
        out = utils.match_histogram(put(tgt.clone()), srcs, mode=mode)
        assert out.is_cuda == on_device and out.shape == tgt.shape
        want = torch.from_numpy(g[key])
        tol = 1e-4 if key == "out_big" else 1e-5  # cond(cov) ~ 1e3 there: the reference's own fp32 eigen-solve is that far from fp64
        assert rel_l2(out.cpu(), want) <= tol, (key, rel_l2(out.cpu(), want))
    same = utils.match_histogram(put(target.clone()), [src1], mode=False)
    assert torch.equal(same.cpu(), target)


def test_match_histogram_device_keeps_the_global_rng_order(hip):
    """The jitter comes from torch's global CPU generator in the reference's order, so whatever is drawn afterwards
    (e.g. the --init random pastiche, style.py:55) is the same as after the oracle's run of the reference formula."""
    from oracle import match_histogram as oracle_mh
    import utils
    g = _hist_fixture()
    target, src1, src2 = (torch.from_numpy(g[k]) for k in ("target", "src1", "src2"))
    torch.manual_seed(5)
    oracle_mh(target.clone(), [src1, src2], mode=True)
    after_oracle = torch.randn(7)
    torch.manual_seed(5)
    utils.match_histogram(target.clone().cuda(), [src1, src2], mode=True)
    assert torch.equal(torch.randn(7), after_oracle)


def test_match_histogram_device_full_size_vs_fp64_oracle(hip):
    """1024x1024 target against a 900x700 source: the device result agrees with the same formula evaluated in fp64 on the
    CPU (same jitter draws) to 1e-5, is deterministic, and moves the channel statistics onto the source's."""
    from oracle import match_histogram as oracle_mh
    import utils
    gen = torch.Generator().manual_seed(3)
    base = torch.rand(1, 1, 1024, 1024, generator=gen)
    target = torch.cat([base * (150 + 20 * k) + torch.rand(1, 1, 1024, 1024, generator=gen) * 40 for k in range(3)], 1) - 110
    source = torch.rand(1, 3, 900, 700, generator=gen) * torch.tensor([200.0, 120.0, 60.0]).view(1, 3, 1, 1) - 90
    torch.manual_seed(11)
    ref = oracle_mh(target.clone(), [source], mode=True, dtype=torch.float64)
    outs = []
    for _ in range(2):
        torch.manual_seed(11)
        outs.append(utils.match_histogram(target.clone().cuda(), [source], mode=True))
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    assert rel_l2(outs[0].cpu(), ref) <= 1e-5
    assert torch.allclose(outs[0].mean((0, 2, 3)).cpu(), source.mean((0, 2, 3)), atol=0.05)
    cov = lambda t: torch.cov(t[0].reshape(3, -1).double())
    assert rel_l2(cov(outs[0].cpu()), cov(source)) <= 1e-2


def test_match_histogram_device_non_finite_input_returns_the_input(hip):
    """Reference utils.py:147-150: a RuntimeError inside (symeig / inverse on non-finite statistics) returns the backup copy."""
    import utils
    g = _hist_fixture()
    target, src1 = torch.from_numpy(g["target"]), torch.from_numpy(g["src1"])
    bad = target.clone()
    bad[0, 1, 3, 3] = float("inf")
    out = utils.match_histogram(bad.clone().cuda(), [src1], mode=True).cpu()
    assert torch.equal(out, bad)


def test_resize_bilinear_device_vs_reference_fixture(hip):
    import numpy as np
    g = np.load(os.path.join(GOLDEN, "resize_bilinear.npz"))
    img = torch.from_numpy(g["img"]).cuda()
    for k in range(4):
        got = hip.resize_bilinear(img, scale_factor=float(g[f"sf_{k}"])).cpu()
        want = torch.from_numpy(g[f"out_sf_{k}"])
        assert got.shape == want.shape and float((got - want).abs().max()) <= 2e-4 and rel_l2(got, want) <= 1e-6, ("sf", k)
        got = hip.resize_bilinear(img, size=tuple(int(v) for v in g[f"hw_{k}"])).cpu()
        want = torch.from_numpy(g[f"out_hw_{k}"])
        assert got.shape == want.shape and float((got - want).abs().max()) <= 2e-4 and rel_l2(got, want) <= 1e-6, ("hw", k)


@pytest.mark.parametrize("src,dst", [(512, 1024), (1024, 2048), (1024, 724), (256, 362)])
def test_resize_bilinear_device_full_size(hip, src, dst):
    """The scale chain of img_img (style.py:57-66) at real sizes against ATen's CPU kernel."""
    import torch.nn.functional as F
    x = rnd(1, 3, src, src, seed=31) * 100
    want = F.interpolate(x, (dst, dst), mode="bilinear", align_corners=False)
    got = hip.resize_bilinear(dev(x), size=(dst, dst)).cpu()
    assert rel_l2(got, want) <= 1e-6
    want = F.interpolate(x, scale_factor=dst / src, mode="bilinear", align_corners=False)
    got = hip.resize_bilinear(dev(x), scale_factor=dst / src).cpu()
    assert got.shape == want.shape and rel_l2(got, want) <= 1e-6


def test_deprocess_u8_is_bit_exact(hip):
    """load.deprocess on the device: the same bytes as the CPU arithmetic of the reference (load.py:47-52), including
    clamping, truncation and values exactly on a byte boundary."""
    from oracle import deprocess_u8
    import load
    x = rnd(1, 3, 517, 389, seed=41) * 120
    x[0, :, 0, :8] = torch.tensor([-200.0, 300.0, 0.0, 1.0, -103.939, 151.061, 0.5, 254.999])  # out of range / boundaries
    want = deprocess_u8(x.clone())
    got = hip.deprocess_u8(dev(x), load._MEAN_BGR).cpu()
    assert torch.equal(got, want)
    import numpy as np
    assert np.array_equal(np.asarray(load.deprocess(x.clone())), want.numpy())
    assert np.array_equal(np.asarray(load.deprocess(dev(x))), want.numpy())  # device tensors take the kernel


@pytest.mark.parametrize("shape", [(1, 64, 128, 256), (2, 8, 34, 66), (1, 512, 16, 16), (1, 64, 181, 181), (2, 8, 33, 66), (1, 16, 90, 45)])
def test_pool2x2_with_kept_decisions_equals_the_recomputing_pair(hip, shape):
    """maua_pool2x2_fwd_codes / _bwd_codes against maua_pool2d_fwd / _bwd (which recompute the arg-max from the input): the
    same bits, ties, zeros (ReLU inputs) and NaNs included."""
    n, c, h, w = shape
    x = torch.relu(rnd(*shape, seed=51))
    x[..., 0:2, 0:2] = 0.5           # a four-way tie
    x[..., 2:4, 2:4] = 0.0           # an all-zero window
    x[0, 0, 4, 5] = float("nan")
    gy = rnd(n, c, h // 2, w // 2, seed=52)
    xd, gyd = dev(x), dev(gy)
    y0 = hip.pool2d_fwd(xd, 2, 2, False, "max")
    codes = torch.empty(n, c, h // 2, w // 2, dtype=torch.uint8, device="cuda")
    y1 = hip.pool2x2_fwd_codes(xd, torch.empty_like(y0), codes)
    for mask in (False, True):
        g0 = hip.pool2d_bwd(gyd, xd, 2, 2, False, "max", relu_mask_by_x=mask)
        g1 = hip.pool2x2_bwd_codes(gyd, codes, torch.empty_like(xd), mask)
        torch.cuda.synchronize()
        assert torch.equal(torch.nan_to_num(g0, nan=7.0), torch.nan_to_num(g1, nan=7.0))
    assert torch.equal(torch.nan_to_num(y0, nan=7.0), torch.nan_to_num(y1, nan=7.0))
    assert int(codes.max()) <= 7


@pytest.mark.parametrize("C,HW,center", [(64, 4096, False), (96, 1000, True), (512, 256, False), (1024, 100, False)])
def test_loss_ledger_equals_the_immediate_entry_points(hip, C, HW, center):
    """Deferred loss finishing (maua_*_ledger + maua_loss_ledger_sum) against the entry points that finish every loss with a
    launch of their own: Gram matrix, gradient matrices and pixel gradients bit-identical; MSE / TV losses bit-identical; the
    fused Gram + MSE loss to fp32 rounding (its partial sums are cut along tiles, not along rows); empty records leave
    the slot alone; the total is the left-to-right sum."""
    f = torch.relu(rnd(1, C, HW, 1, seed=61))
    tgt_g = rnd(C, C, seed=62) * 0.01
    tgt_g = dev((tgt_g + tgt_g.t()).contiguous())
    x, t = rnd(1, 8, 33, 47, seed=63), rnd(1, 8, 33, 47, seed=64)
    fd, xd, td = dev(f), dev(x), dev(t)
    n = f.numel()
    # immediate
    g0, m0 = hip.gram_fwd(fd, 1.0 / n, center)
    d0, l0 = torch.empty_like(g0), torch.zeros(4, device="cuda")
    hip.mse_fwd_bwd(g0, tgt_g, d0, 3.0 / (C * C), 0.7, False, l0[1:2])
    gm0 = torch.zeros_like(xd)
    hip.mse_fwd_bwd(xd, td, gm0, 0.25, 1.5, False, l0[2:3], mask_grad_by_x=True)
    gt0 = torch.zeros_like(xd)
    hip.tv_fwd_bwd(xd, gt0, 1e-3, False, l0[3:4])
    # ledger: slot 0 stays empty and keeps the value found there
    led = hip.loss_ledger(1, 4, "cuda")
    l1 = torch.zeros(1, 4, device="cuda")
    l1[0, 0] = 42.0
    g1, d1 = torch.empty_like(g0), torch.empty_like(g0)
    m1 = torch.empty(C, device="cuda") if center else None
    assert hip.gram_mse_ledger_supported(C)
    hip.gram_fwd_mse_ledger(fd, 1.0 / n, center, g1, m1, tgt_g, d1, 3.0 / (C * C), 0.7, led[0], 1)
    gm1, gt1 = torch.zeros_like(xd), torch.zeros_like(xd)
    hip.mse_fwd_bwd_ledger(xd, td, gm1, 0.25, 1.5, False, led[0], 2, mask_grad_by_x=True)
    hip.tv_fwd_bwd_ledger(xd, gt1, 1e-3, False, led[0], 3)
    tot = torch.zeros(1, device="cuda")
    hip.loss_ledger_sum(led, l1, tot)
    torch.cuda.synchronize()
    assert torch.equal(g0, g1) and torch.equal(d0, d1) and torch.equal(gm0, gm1) and torch.equal(gt0, gt1)
    if center:
        assert torch.equal(m0, m1)
    assert float(l1[0, 0]) == 42.0
    assert float(l1[0, 2]) == float(l0[2]) and float(l1[0, 3]) == float(l0[3])
    assert abs(float(l1[0, 1]) - float(l0[1])) <= 2e-7 * abs(float(l0[1]))
    want = torch.tensor(42.0) + l1[0, 1].cpu()
    want = (want + l1[0, 2].cpu()) + l1[0, 3].cpu()
    assert float(tot[0]) == float(want)
    assert float(led[:, :, 0].abs().max()) == 0.0  # records are empty again
    # a second evaluation through the same ledger gives the same values
    hip.mse_fwd_bwd_ledger(xd, td, gm1, 0.25, 1.5, False, led[0], 2, mask_grad_by_x=True)
    hip.loss_ledger_sum(led, l1, tot)
    torch.cuda.synchronize()
    assert float(l1[0, 2]) == float(l0[2])
    assert hip.gram_mse_ledger_supported(1408) and not hip.gram_mse_ledger_supported(1409)


@pytest.mark.parametrize("cin,c,H,W", [(64, 64, 64, 96), (128, 128, 40, 72), (128, 64, 33, 50), (64, 128, 16, 520)])
def test_backward_pass_with_the_gram_backward_along(hip, cin, c, H, W):
    """maua_conv3x3_x3w_gram (backward-data pass + D . F of the style loss on the layer's input activation, masked by that
    activation) against fp64 and against the two separate passes it replaces (maua_conv3x3_x3w backward + maua_gram_bwd)."""
    gy = rnd(1, cin, H, W, seed=71) * (rnd(1, cin, H, W, seed=72) > 0)     # a ReLU-masked gradient
    w = rnd(cin, c, 3, 3, seed=73) * (2.0 / (9 * c)) ** 0.5               # layer c -> cin channels; backward-data: cin -> c
    f = torch.relu(rnd(1, c, H, W, seed=74))
    d = rnd(c, c, seed=75) * 1e-3
    d = (d + d.t()).contiguous()
    _, bb, wsc = hip.conv_pack_filters_x3w(dev(w))
    dbank, dinv = hip.conv_x3w_dmat_bank(c, "cuda")
    hip.conv_pack_dmat_x3w(dev(d), dbank, dinv)
    fused = hip.conv3x3_x3w_gram(dev(gy), bb, wsc, dev(f), dbank, dinv, c, 1)
    two = hip.conv3x3_x3w(dev(gy), bb, wsc, None, c, 1, False)
    hip.gram_bwd(dev(d), dev(f), None, two, True, relu_mask=dev(f))
    torch.cuda.synchronize()
    ref = F.conv_transpose2d(gy.double(), w.double(), padding=1) + torch.einsum("kc,bkhw->bchw", d.double(), f.double())
    ref = ref * (f > 0)
    e_fused, e_two = rel_l2(fused.cpu(), ref), rel_l2(two.cpu(), ref)
    assert e_fused <= 3e-7 and e_fused <= 1.5 * e_two + 1e-8, (e_fused, e_two)
    assert torch.equal(fused == 0, two == 0) or float(((fused == 0) != (two == 0)).sum()) < 1e-5 * fused.numel()
    again = hip.conv3x3_x3w_gram(dev(gy), bb, wsc, dev(f), dbank, dinv, c, 1)
    assert torch.equal(fused, again)
    # the bank's scale: max |D| lands in [32, 64)
    assert 32.0 <= float(d.abs().max()) / float(dinv[0]) < 64.0


@pytest.mark.parametrize("n,cin,cout,H,W", [(1, 64, 64, 64, 96), (2, 128, 128, 40, 72), (1, 64, 128, 34, 62), (1, 16, 200, 18, 260),
                                            (1, 512, 512, 64, 64), (1, 256, 256, 32, 64)])
def test_conv_relu_pool_in_one_launch_equals_the_three_steps(hip, n, cin, cout, H, W):
    """maua_conv3x3_x3w_relu_pool against maua_conv3x3_x3w (+ ReLU) followed by maua_pool2x2_fwd_codes: the same pooled map
    and the same decision bytes, bit for bit (zero windows, ties and all), and the backward routing from those bytes."""
    x = torch.relu(rnd(n, cin, H, W, seed=81))
    x[:, :, :4, :4] = 0.0                                  # windows that are zero after ReLU whatever the filters
    w = rnd(cout, cin, 3, 3, seed=82) * (2.0 / (9 * cin)) ** 0.5
    b = rnd(cout, seed=83) * 0.1
    b[:4] = -100.0                                         # whole channels at zero: every window a four-way tie
    bf, _, wsc = hip.conv_pack_filters_x3w(dev(w))
    one_pass = torch.empty(16, dtype=torch.uint8, device="cuda")  # no room for split-K slabs: one pass over the channels
    y = hip.conv3x3_x3w(dev(x), bf, wsc, dev(b), cout, 1, True, workspace=one_pass)
    p0 = torch.empty(n, cout, H // 2, W // 2, device="cuda")
    c0 = torch.empty(n, cout, H // 2, W // 2, dtype=torch.uint8, device="cuda")
    hip.pool2x2_fwd_codes(y, p0, c0)
    p1, c1 = torch.full_like(p0, float("nan")), torch.full_like(c0, 255)
    hip.conv3x3_x3w_relu_pool(dev(x), bf, wsc, dev(b), cout, 1, p1, c1)
    torch.cuda.synchronize()
    assert torch.equal(p0, p1) and torch.equal(c0, c1)
    # with room for slabs the channel loop may be split: ReLU and pool then happen in the pass that adds the slabs - against the same
    # three steps under the same split
    ws = torch.empty(max(hip.conv_x3w_workspace_bytes(n, cin, H, W, cout, 1), 16), dtype=torch.uint8, device="cuda")
    y = hip.conv3x3_x3w(dev(x), bf, wsc, dev(b), cout, 1, True, workspace=ws)
    hip.pool2x2_fwd_codes(y, p0, c0)
    p2, c2 = torch.full_like(p0, float("nan")), torch.full_like(c0, 255)
    hip.conv3x3_x3w_relu_pool(dev(x), bf, wsc, dev(b), cout, 1, p2, c2, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(p0, p2) and torch.equal(c0, c2)
    from conftest import planar_codes
    assert int((planar_codes(c1)[:, :4] == 4).all()) == 1  # zero channels: first position, "not positive"


def test_fused_launches_give_the_same_bits_every_time():
    """tools/stress_fused.py: the fused convolution launches (Gram backward in the K loop, ReLU + pool in the epilogue) repeated
    while a GEMM on another stream keeps the chip busy, and two L-BFGS states fed the same gradients: not one differing bit."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "stress_fused.py"), "40"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "TOTAL MISMATCHES 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_gram_finish_batch_equals_the_per_layer_launches(hip):
    """maua_gram_partial + ONE maua_gram_finish_mse_batch over five style-layer shapes (covariance form among them, a layer with the
    first-level slab fold) against maua_gram_fwd_mse_ledger per layer: Gram, D and the ledger losses bit for bit."""
    shapes = [(64, 300 * 300, False), (128, 150 * 150, True), (256, 75 * 75, False), (512, 37 * 37, False), (96, 1000, True)]
    led_a, led_b = hip.loss_ledger(1, 8, "cuda"), hip.loss_ledger(1, 8, "cuda")
    layers, want = [], []
    for k, (c, hw, center) in enumerate(shapes):
        f = dev(torch.relu(rnd(1, c, hw, 1, seed=200 + k)))
        t = dev(rnd(c, c, seed=300 + k) * 0.1)
        ws = torch.empty(hip.gram_workspace_bytes(c, hw), dtype=torch.uint8, device="cuda")
        g0, d0, m0 = torch.empty(c, c, device="cuda"), torch.empty(c, c, device="cuda"), torch.empty(c, device="cuda")
        hip.gram_fwd_mse_ledger(f, 1.0 / (c * hw), center, g0, m0, t, d0, 0.5 / (c * c), 3.0 / (c * c), led_a[0], k, workspace=ws)
        want.append((g0, d0, m0))
        ws2 = torch.full_like(ws, 255)
        g1, d1, m1 = torch.full_like(g0, float("nan")), torch.full_like(d0, float("nan")), torch.full_like(m0, float("nan"))
        hip.gram_partial(f, center, m1, ws2)
        layers.append(dict(workspace=ws2, gram=g1, target=t, dmat=d1, c=c, hw=hw, scale=1.0 / (c * hw), loss_scale=0.5 / (c * c),
                           grad_scale=3.0 / (c * c), ledger=led_b[0], slot=k, mean=m1, center=center))
    hip.GramFinishBatch(layers).run()
    la, ta = torch.zeros(8, device="cuda"), torch.zeros(1, device="cuda")
    lb, tb = torch.zeros(8, device="cuda"), torch.zeros(1, device="cuda")
    hip.loss_ledger_sum(led_a, la, ta)
    hip.loss_ledger_sum(led_b, lb, tb)
    torch.cuda.synchronize()
    for (g0, d0, m0), l in zip(want, layers):
        assert torch.equal(g0, l["gram"]) and torch.equal(d0, l["dmat"])
        if l["center"]:
            assert torch.equal(m0, l["mean"])
    assert torch.equal(la, lb) and torch.equal(ta, tb) and float(ta) > 0
    with pytest.raises(hip.HipError):
        hip.GramFinishBatch(layers + layers).run()       # at most eight layers per call


def test_gram_partial_batch_leaves_the_slabs_of_the_per_layer_launches(hip):
    """maua_gram_partial_batch (the partial kernels of all style layers in at most two launches, their first-level folds in one)
    against maua_gram_partial per layer: the five VGG style-layer shapes of a 512 x 512 image (two of them fold), a ragged one and two
    covariance-form layers (their row means come out of the batched call) - the same workspaces byte for byte, and through the
    batched finishing launch the Gram, D and losses of maua_gram_fwd_mse_ledger."""
    shapes = [(64, 512 * 512, False), (128, 256 * 256, False), (256, 128 * 128, False), (512, 64 * 64, False), (512, 32 * 32, False),
              (200, 37 * 41, False), (96, 1000, True), (384, 64 * 64, True)]
    led_a, led_b = hip.loss_ledger(1, 8, "cuda"), hip.loss_ledger(1, 8, "cuda")
    layers, want, slabs = [], [], []
    for k, (c, hw, center) in enumerate(shapes):
        f = dev(torch.relu(rnd(1, c, hw, 1, seed=400 + k)))
        t = dev(rnd(c, c, seed=500 + k) * 0.1)
        ws = torch.zeros(hip.gram_workspace_bytes(c, hw), dtype=torch.uint8, device="cuda")
        g0, d0, m0 = torch.empty(c, c, device="cuda"), torch.empty(c, c, device="cuda"), torch.empty(c, device="cuda")
        hip.gram_fwd_mse_ledger(f, 1.0 / (c * hw), center, g0, m0, t, d0, 0.5 / (c * c), 3.0 / (c * c), led_a[0], k, workspace=ws)
        want.append((g0, d0, m0))
        ws1 = torch.zeros_like(ws)
        hip.gram_partial(f, center, torch.empty_like(m0), ws1)
        slabs.append(ws1)
        g1, d1 = torch.full_like(g0, float("nan")), torch.full_like(d0, float("nan"))
        wsb, m1 = torch.zeros_like(ws), None
        if center:
            m1 = torch.full_like(m0, float("nan"))
            m2 = torch.full_like(m0, float("nan"))
            hip.gram_row_means(f, m2, torch.zeros_like(ws))      # (the means on their own: the same bits)
            assert torch.equal(m2, m0)
        layers.append(dict(workspace=wsb, gram=g1, target=t, dmat=d1, c=c, hw=hw, scale=1.0 / (c * hw), loss_scale=0.5 / (c * c),
                           grad_scale=3.0 / (c * c), ledger=led_b[0], slot=k, f=f, mean=m1))
    fin = hip.GramFinishBatch(layers)
    fin.run_partial()
    torch.cuda.synchronize()
    for ws1, l, (c, hw, center) in zip(slabs, layers, shapes):
        n = l["workspace"].numel() if not center else None  # (covariance: the row-mean partial sums sit where no slab reaches; compare the slabs)
        assert torch.equal(ws1[:n], l["workspace"][:n]) if not center else True
    fin.run()
    la, ta = torch.zeros(8, device="cuda"), torch.zeros(1, device="cuda")
    lb, tb = torch.zeros(8, device="cuda"), torch.zeros(1, device="cuda")
    hip.loss_ledger_sum(led_a, la, ta)
    hip.loss_ledger_sum(led_b, lb, tb)
    torch.cuda.synchronize()
    for (g0, d0, m0), l in zip(want, layers):
        assert torch.equal(g0, l["gram"]) and torch.equal(d0, l["dmat"])
        if l["mean"] is not None:
            assert torch.equal(m0, l["mean"])
    assert torch.equal(la, lb) and torch.equal(ta, tb) and float(ta) > 0


def test_dmat_pack_batch_equals_the_per_layer_launches(hip):
    """maua_conv_pack_dmat_x3w_batch (the one-tap banks of relu1_1 / 2_1 / 3_1's D matrices in one launch) against the per-layer call."""
    items, want = [], []
    for k, c in enumerate((64, 128, 256, 48)):
        d = rnd(c, c, seed=700 + k) * 10.0 ** (k - 4)
        d = dev((d + d.t()).contiguous())
        b0, i0 = hip.conv_x3w_dmat_bank(c, "cuda")
        hip.conv_pack_dmat_x3w(d, b0[0], i0)
        want.append((b0, i0))
        b1, i1 = hip.conv_x3w_dmat_bank(c, "cuda")
        b1.fill_(255)
        items.append((d, b1[0], i1))
    hip.DmatPackBatch(items).run()
    torch.cuda.synchronize()
    for (b0, i0), (_, b1, i1) in zip(want, items):
        assert torch.equal(b0[0], b1) and torch.equal(i0, i1)
    with pytest.raises(hip.HipError):
        hip.DmatPackBatch(items + items[:1]).run()


def test_soak_no_silent_differences_at_the_1e_6_level():
    """The fault the first 128 x 128 Gram form had (one accumulator register of one wave short of a k-step's products in lanes 48-63,
    7e-5 per workgroup launch with two of its workgroups per CU: tools/mfma_probe/gram128_zero_lanes.hip reproduces it in seconds,
    profiles/probes_r05.md section 4 lists what was ruled out) would pass every single-launch test.  tools/soak_kernels.py launches every
    matrix kernel family that runs two MFMA waves per SIMD - conv_x3w, conv_x3q, conv_x3p, conv_x3w's fused Gram form, the Gram forward
    kernels (two-role 128 x 128 and 64 x 64) and the Gram backward on conv1x1_x3 - 400 times each on fixed inputs and compares every
    result with the first on the device: ~1e6 - 3e6 workgroup launches per family, a rate of 1e-5 would show dozens of times.
    (Reference arithmetic of the families: /root/reference/models.py:129-130 convolutions, /root/reference/loss.py:91 Gram.)"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "soak_kernels.py"), "400"], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "differing launches in all: 0" in r.stdout, (r.stdout[-2000:], r.stderr[-1500:])
    for flag in ("gram_t128=0", "gram_t128=2"):
        env = dict(os.environ, MAUA_PLAN=flag)
        r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "soak_kernels.py"), "200"], env=env, capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0 and "differing launches in all: 0" in r.stdout, (flag, r.stdout[-2000:], r.stderr[-1500:])


def test_stream_soak_of_the_pairings_the_product_runs():
    """Round 6 (profiles/probes_r06.md section 2): tools/soak_streams.py runs kernels on two streams at once and compares every result of
    the one with a run made alone.  It found what round 5 suspected: a kernel WITH packed fp32 instructions beside an MFMA kernel of
    another stream loses results (conv3x3_few_out beside conv1x1_x3 under the MFMA-padding build: 16951 of 17008 runs, lanes 32-63; the
    L-BFGS sweeps beside the 128 x 128 Gram kernel in the product build: 2-14 of ~2000 runs).  The product never forms such a pair - the
    kernels that keep packed fp32 (L-BFGS sweeps, Adam, resize) run between an evaluation's last join and the next one's first launch
    (test_packed_fp32_kernels_never_share_the_gpu_with_matrix_kernels), everything that can stand beside an MFMA kernel is built without
    them (tests/test_abi.py) - and this soaks the pairs it DOES form, a frame batch's side streams (reference style.py:192-290: frames are
    independent problems): the per-frame loss / pooling kernels and the split-K finishing kernels beside the per-frame Gram and Gram-backward
    kernels, and the L-BFGS update of one frame beside another frame's.  ~2.5 s per pairing, every victim result compared bit for bit."""
    import subprocess
    import sys
    pairs = ",".join(f"{v}:{a}" for v in ("pointwise", "splitk_finish") for a in ("gram128", "gram64", "conv1x1")) + ",lbfgs_512:lbfgs,lbfgs_128:lbfgs"
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "soak_streams.py"), "--seconds", "2.5", "--pairs", pairs],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "# total differing runs: 0;" in r.stdout, (r.stdout[-3000:], r.stderr[-1500:])
    assert r.stdout.count(" runs differ") >= 8 + 4
