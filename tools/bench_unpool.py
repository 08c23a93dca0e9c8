"""Backward pass of a conv + ReLU + pool group: maua_pool2x2_bwd_codes + maua_conv3x3_x3w(_gram) against maua_conv3x3_x3w_unpool.
    python tools/bench_unpool.py [image side] [rounds] [launches per round]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "maua-style_amd"))
import torch
import hip

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
torch.manual_seed(0)
ws = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")


def timed(fn):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / reps


for name, c, s, gram in (("conv1_2", 64, side, True), ("conv2_2", 128, side // 2, True), ("conv3_4", 256, side // 4, False), ("conv4_4", 512, side // 8, False)):
    act = torch.relu(torch.randn(1, c, s, s, device="cuda"))
    f = torch.relu(torch.randn(1, c, s, s, device="cuda"))
    gp = torch.randn(1, c, s // 2, s // 2, device="cuda") * (torch.rand(1, c, s // 2, s // 2, device="cuda") > 0.5)
    w = torch.randn(c, c, 3, 3, device="cuda") * (2.0 / (9 * c)) ** 0.5
    _, bb, wsc = hip.conv_pack_filters_x3w(w)
    pooled = torch.empty_like(gp); codes = torch.empty(1, c, s // 2, s // 2, dtype=torch.uint8, device="cuda")
    hip.pool2x2_fwd_codes(act, pooled, codes)
    full = torch.empty_like(act); out = torch.empty_like(act)
    dbank = dinv = None
    if gram:
        d = torch.randn(c, c, device="cuda") * 1e-3
        dbank, dinv = hip.conv_x3w_dmat_bank(c, "cuda")
        hip.conv_pack_dmat_x3w((d + d.t()).contiguous(), dbank, dinv)

    def two():
        hip.pool2x2_bwd_codes(gp, codes, full, True)
        if gram:
            hip.conv3x3_x3w_gram(full, bb, wsc, f, dbank, dinv, c, 1, out=out, workspace=ws)
        else:
            hip.conv3x3_x3w(full, bb, wsc, None, c, 1, False, out=out, out_relu_mask=f, workspace=ws)

    def conv_only():
        if gram:
            hip.conv3x3_x3w_gram(full, bb, wsc, f, dbank, dinv, c, 1, out=out, workspace=ws)
        else:
            hip.conv3x3_x3w(full, bb, wsc, None, c, 1, False, out=out, out_relu_mask=f, workspace=ws)

    def one():
        hip.conv3x3_x3w_unpool(gp, codes, True, bb, wsc, c, 1, out=out, out_relu_mask=f, dmat_bank=dbank, dmat_inv_scale=dinv, workspace=ws)

    t2, tc, t1 = [], [], []
    for _ in range(rounds):
        t2.append(timed(two)); tc.append(timed(conv_only)); t1.append(timed(one))
    two(); ref = out.clone(); one(); torch.cuda.synchronize()
    print(f"{name} {c}ch @{s}{' +gram' if gram else ''}: pool_bwd + conv {min(t2):7.1f} us   conv alone {min(tc):7.1f}   unpool-conv {min(t1):7.1f} us   equal {torch.equal(ref, out)}")
