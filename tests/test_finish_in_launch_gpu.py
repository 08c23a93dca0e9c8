"""Split channel loops finished INSIDE the producing launch (round 6, maua_conv_arm_workspace; conv_x3q.hip / conv_x3w.hip): with arrival
counters the workgroup that arrives last at a tile adds the other splits' slabs to its own sums and runs the one-pass epilogue - instead of
a conv_splitk_finish launch behind every split convolution (19 of them were 9 % of a 512 x 512 iteration).

The arithmetic is the finishing launch's (slabs added in split order, then bias, previous contents, ReLU, mask; the pooling form's decisions),
so the bar is BIT IDENTITY with the two-launch form for every fused form, split count and ragged geometry, on poisoned workspaces, and
launch after launch (the last arriver leaves the counter at zero).  The two-launch form itself is pinned to fp64 in
tests/test_conv_x3q_gpu.py / test_conv_x3w_gpu.py; one fp64 comparison here guards against both forms sharing a mistake.
Reference arithmetic: `nn.Conv2d(c, c, 3, padding=1)` + ReLU (+ `nn.MaxPool2d(2, 2)`), /root/reference/models.py:120,129-130."""
import math

import pytest
import torch
import torch.nn.functional as F

import plan
from conftest import rel_l2

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    import hip as h
    h.lib()
    return h


@pytest.fixture(autouse=True)
def _restore(hip):
    yield
    for k in ("x3q_ks", "x3w_ks", "finish_in_launch_max_ks"):
        plan.OVERRIDES.pop(k, None)
    plan.forward_to_library(hip.lib())
    hip.conv_arm_workspace(None)


def force(hip, **fields):
    for k, v in fields.items():
        plan.OVERRIDES[k] = str(v)
    plan.forward_to_library(hip.lib())


def data(cin, cout, H, W, n, seed=0):
    g = torch.Generator().manual_seed(seed)
    x = torch.relu(torch.randn(n, cin, H, W, generator=g)).cuda()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))).cuda()
    b = (torch.randn(cout, generator=g) * 0.1).cuda()
    return x, w, b, g


FAMILIES = ["x3q", "x3w"]
# cin, cout, H, W, n
SHAPES = [(256, 256, 64, 64, 1), (512, 200, 33, 45, 1), (128, 64, 66, 70, 2), (256, 128, 16, 32, 1), (512, 512, 32, 32, 1), (256, 192, 90, 91, 1)]


def _entry(hip, family):
    if family == "x3q":
        return hip.conv_pack_filters_x3q, hip.conv3x3_x3q, hip.conv3x3_x3q_relu_pool, hip.conv3x3_x3q_unpool, hip.conv_x3q_workspace_bytes, hip.conv_x3q_split
    return hip.conv_pack_filters_x3w, hip.conv3x3_x3w, hip.conv3x3_x3w_relu_pool, hip.conv3x3_x3w_unpool, hip.conv_x3w_workspace_bytes, hip.conv_x3w_split


def _poisoned(nbytes):
    ws = torch.empty(max(nbytes, 16) // 4 * 4 + 4, dtype=torch.uint8, device="cuda")
    ws[:ws.numel() // 4 * 4].view(torch.float32)[:] = float("nan")
    return ws


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("ks", [2, 3, 4])
@pytest.mark.parametrize("cin,cout,H,W,n", SHAPES)
def test_in_launch_finish_equals_the_finishing_launch_bit_for_bit(hip, family, ks, cin, cout, H, W, n):
    """Every epilogue form of the plain entry point: bias / ReLU / previous contents / mask - sixteen combinations - armed against unarmed."""
    pack, conv, _, _, ws_bytes, split = _entry(hip, family)
    if cin // (32 if family == "x3q" else 16) < 2 * ks:
        pytest.skip("the layer has too few chunks for this split")
    force(hip, **{f"{family}_ks": ks})
    x, w, b, g = data(cin, cout, H, W, n)
    bf, _, wsc = pack(w)
    assert split(n, cin, H, W, cout, 1) == ks
    mask = torch.relu(torch.randn(n, cout, H, W, generator=g)).cuda()
    prev = torch.randn(n, cout, H, W, generator=g).cuda()
    ws = _poisoned(ws_bytes(n, cin, H, W, cout, 1))
    for bias in (None, b):
        for relu in (False, True):
            for accumulate in (False, True):
                for m in (None, mask):
                    hip.conv_arm_workspace(None)
                    want = conv(x, bf, wsc, bias, cout, 1, relu, out=prev.clone(), out_relu_mask=m, accumulate=accumulate, workspace=ws)
                    counters = hip.conv_arm_workspace(ws)
                    got = conv(x, bf, wsc, bias, cout, 1, relu, out=prev.clone(), out_relu_mask=m, accumulate=accumulate, workspace=ws)
                    again = conv(x, bf, wsc, bias, cout, 1, relu, out=prev.clone(), out_relu_mask=m, accumulate=accumulate, workspace=ws)
                    torch.cuda.synchronize()
                    assert torch.equal(got, want) and torch.equal(again, want), (bias is not None, relu, accumulate, m is not None)
                    assert int(counters.view(torch.int32).abs().max()) == 0   # every tile's counter is back at zero
    # and against fp64 (the armed form, bias + ReLU)
    ref = torch.relu(F.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), padding=1))
    assert rel_l2(conv(x, bf, wsc, b, cout, 1, True, workspace=ws).cpu(), ref) <= 2e-6


@pytest.mark.parametrize("family", FAMILIES)
@pytest.mark.parametrize("ks", [2, 4])
@pytest.mark.parametrize("cin,cout,H,W,n", [(256, 256, 64, 64, 1), (128, 192, 66, 70, 2), (512, 64, 45, 45, 1)])
def test_in_launch_finish_of_the_pooling_and_unpooling_forms(hip, family, ks, cin, cout, H, W, n):
    """conv + ReLU + 2x2 max pool (the last arriver pools complete sums in its epilogue; the two-launch form pools in
    conv_splitk_finish_pool_kernel) and the backward pass staged from a pooled gradient, masked and not: armed = unarmed, bit for bit,
    pooled map and decision bytes alike; odd planes (45 x 45: floor-mode pooling) included."""
    pack, _, relu_pool, unpool, ws_bytes, split = _entry(hip, family)
    if cin // (32 if family == "x3q" else 16) < 2 * ks or cout % 8:
        pytest.skip("not a shape of this form")
    force(hip, **{f"{family}_ks": ks})
    assert split(n, cin, H, W, cout, 1) == ks
    x, w, b, _ = data(cin, cout, H, W, n, seed=3)
    bf, bb, wsc = pack(w)
    ws = _poisoned(max(ws_bytes(n, cin, H, W, cout, 1), ws_bytes(n, cout, H, W, cin, 1)))
    PH, PW = H // 2, W // 2
    res = []
    for armed in (False, True, True):
        counters = hip.conv_arm_workspace(ws if armed else None)
        pooled = torch.full((n, cout, PH, PW), float("nan"), device="cuda")
        codes = torch.full((n * cout * PH * PW,), 255, dtype=torch.uint8, device="cuda")
        relu_pool(x, bf, wsc, b, cout, 1, pooled, codes, workspace=ws)
        # backward: the gradient of the pooled map through the decision bytes into the layer's input gradient
        gp = torch.randn(n, cout, PH, PW, generator=torch.Generator().manual_seed(9)).cuda()
        gx = torch.full((n, cin, H, W), float("nan"), device="cuda")
        unpool(gp, codes, True, bb, wsc, cin, 1, out=gx, workspace=ws)
        gxm = torch.full((n, cin, H, W), float("nan"), device="cuda")
        unpool(gp, codes, False, bb, wsc, cin, 1, out=gxm, out_relu_mask=x, workspace=ws)
        torch.cuda.synchronize()
        assert counters is None or int(counters.view(torch.int32).abs().max()) == 0
        res.append((pooled, codes, gx, gxm))
    for k in (1, 2):
        for a_, b_ in zip(res[0], res[k]):
            assert torch.equal(a_, b_)
    ref = F.max_pool2d(torch.relu(F.conv2d(x.cpu().double(), w.cpu().double(), b.cpu().double(), padding=1)), 2, 2)
    assert rel_l2(res[1][0].cpu(), ref) <= 2e-6


def test_larger_splits_and_other_workspaces_keep_the_finishing_launch(hip):
    """Splits beyond finish_in_launch_max_ks, and launches on a workspace the thread did not arm, take the two-launch form: same bits, and
    the armed workspace's counters are not touched."""
    force(hip, x3q_ks=8)
    cin = cout = 512
    x, w, b, _ = data(cin, cout, 32, 64, 1, seed=5)
    bf, _, wsc = hip.conv_pack_filters_x3q(w)
    assert hip.conv_x3q_split(1, cin, 32, 64, cout, 1) == 8
    ws = _poisoned(hip.conv_x3q_workspace_bytes(1, cin, 32, 64, cout, 1))
    other = _poisoned(hip.conv_x3q_workspace_bytes(1, cin, 32, 64, cout, 1))
    hip.conv_arm_workspace(None)
    want = hip.conv3x3_x3q(x, bf, wsc, b, cout, 1, True, workspace=ws)
    counters = hip.conv_arm_workspace(ws)
    got8 = hip.conv3x3_x3q(x, bf, wsc, b, cout, 1, True, workspace=ws)       # 8 slabs > 4: second launch
    force(hip, x3q_ks=2)
    want2 = hip.conv3x3_x3q(x, bf, wsc, b, cout, 1, True, workspace=other)   # not the armed workspace: second launch
    got2 = hip.conv3x3_x3q(x, bf, wsc, b, cout, 1, True, workspace=ws)       # armed: in the launch
    torch.cuda.synchronize()
    assert torch.equal(got8, want) and torch.equal(got2, want2)
    assert int(counters.view(torch.int32).abs().max()) == 0
    force(hip, finish_in_launch_max_ks=1)                                      # the planner's switch for the library: never in the launch
    got_off = hip.conv3x3_x3q(x, bf, wsc, b, cout, 1, True, workspace=ws)
    torch.cuda.synchronize()
    assert torch.equal(got_off, want2)


def test_in_launch_finish_under_a_captured_graph_and_many_replays(hip):
    """The engine's use: armed once, launches captured into a hipGraph, replayed - no memset in the graph, the counters return to zero by
    themselves; 200 replays of a chain of split launches equal the eager two-launch results."""
    force(hip, x3q_ks=2, x3w_ks=4)
    x, w, b, _ = data(256, 256, 64, 64, 1, seed=7)
    bq, bbq, wsq = hip.conv_pack_filters_x3q(w)
    bw, _, wsw = hip.conv_pack_filters_x3w(w)
    ws = _poisoned(max(hip.conv_x3q_workspace_bytes(1, 256, 64, 64, 256, 1), hip.conv_x3w_workspace_bytes(1, 256, 64, 64, 256, 1)))
    y1, y2, y3 = (torch.empty(1, 256, 64, 64, device="cuda") for _ in range(3))

    def chain():
        hip.conv3x3_x3q(x, bq, wsq, b, 256, 1, True, out=y1, workspace=ws)
        hip.conv3x3_x3w(y1, bw, wsw, b, 256, 1, True, out=y2, workspace=ws)
        hip.conv3x3_x3q(y2, bbq, wsq, None, 256, 1, False, out=y3, out_relu_mask=x, workspace=ws)
    hip.conv_arm_workspace(None)
    chain()
    torch.cuda.synchronize()
    want = y3.clone()
    counters = hip.conv_arm_workspace(ws)
    chain()
    torch.cuda.synchronize()
    assert torch.equal(y3, want)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        hip.conv_arm_workspace(ws, counters, zero=False)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            chain()
        bad = 0
        for _ in range(200):
            y3.fill_(float("nan"))
            graph.replay()
            bad += int(not torch.equal(y3, want))
    torch.cuda.synchronize()
    assert bad == 0 and int(counters.view(torch.int32).abs().max()) == 0
