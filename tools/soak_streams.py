"""Cross-stream soak for the packed-fp32 x MFMA exposure (profiles/probes_r05.md section 4).

Round 4's Gram fault was a packed fp32 vector instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) losing its low destination register in
lanes 48-63 while ANOTHER wave of the same SIMD issued MFMAs.  Since round 5 no kernel of the library has both kinds of instruction
(tests/test_abi.py disassembles it), but the L-BFGS sweeps, the split-K finishing kernels, conv3x3_few_out, the strided stem kernels, the
bilinear resize and Adam keep their packed instructions (they issue no MFMA; removing them costs 11 % at 1024 x 1024).  Inside one stream no
MFMA wave shares a SIMD with them.  Where kernels of different kinds can meet on a CU is two streams: a frame batch's side streams (config 4:
per-frame Gram / loss / L-BFGS kernels of several frames at once, reference style.py:192-290 - frames are independent B = 1 problems).

This tool co-schedules every such VICTIM (stream A; every result compared bit for bit, on the device, with a run made alone) with every MFMA
AGGRESSOR (stream B, looping), with random delays in front of the victims so that the two streams drift through each other's phases:

    python tools/soak_streams.py [--seconds S] [--victims a,b] [--aggressors x,y] [--out FILE]
    MAUA_HIP_LIB=tools/_build/libmaua_pad.so python tools/soak_streams.py ...    (the amplifier build: tools/build_amplified.py - LLVM's
        --amdgpu-mfma-padding-ratio=100 puts s_nop between all MFMAs, which made the faulty kernel fail in EVERY launch)

The aggressors include launches that leave half of every SIMD's registers free (256 workgroups of four 256-register waves: one wave per SIMD),
so that victim waves really become their SIMD neighbours - a kernel that fills the CU (conv_x3q, conv_x3p, gram 128 x 128) only meets victims in
its tails.  Prints one line per pairing: victim launches, victim workgroups, differing launches.  Exit code 1 if anything differed."""
import argparse
import json
import math
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip  # noqa: E402


def build_victims(g):
    """name -> (run() -> list of result tensors, workgroups per run, kernels with packed fp32 it exercises)."""
    dev = "cuda"
    V = {}

    # ---- L-BFGS: a fixed sequence of iterations from a fresh state (pair, pair_dots, finish_dots, coeffs_tri, combine)
    def lbfgs(n, history, iters):
        st = hip.LbfgsState(n, history, dev)
        x0 = torch.randn(n, device=dev, generator=g)
        grads = [torch.randn(n, device=dev, generator=g) * (1.0 + 0.1 * k) for k in range(iters)]
        x = torch.empty_like(x0)

        def run():
            st.reset()
            x.copy_(x0)
            for k in range(iters):
                # g_k = fixed direction + a part that follows x, so that y_k . s_k > 0 and pairs are accepted
                gk = grads[k] * 1e-3 + x * (0.5 + 0.01 * k)
                st.iterate(x, gk, lr=1.0)
            return [x, st.buf]
        wgs = iters * (n // 256 // 4 + n // 64 // 4 + 4)
        return run, wgs
    V["lbfgs_512"] = lbfgs(3 * 512 * 512, 20, 24) + ("lbfgs_pair_dots<true>, lbfgs_combine_v4",)
    V["lbfgs_1024"] = lbfgs(3 * 1024 * 1024, 10, 12) + ("lbfgs_pair_dots, lbfgs_combine_kernel<16>",)
    V["lbfgs_128"] = lbfgs(3 * 128 * 128, 50, 60) + ("lbfgs_pair_dots (grid-y groups), lbfgs_combine_v4 / <4>",)

    # ---- split-K finishing kernels behind their (MFMA) convolution: conv_splitk_finish, conv_splitk_finish_pool<true / false>
    cin = cout = 512
    xs = torch.relu(torch.randn(1, cin, 64, 64, device=dev, generator=g))
    w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * math.sqrt(2.0 / (9 * cin))
    fq, _, wsq = hip.conv_pack_filters_x3q(w)
    bias = torch.randn(cout, device=dev, generator=g) * 0.1
    ws = torch.empty(max(hip.conv_x3q_workspace_bytes(1, cin, 64, 64, cout, 1), hip.conv_x3w_workspace_bytes(1, cin, 64, 64, cout, 1), 16),
                     dtype=torch.uint8, device=dev)
    assert hip.conv_x3q_split(1, cin, 64, 64, cout, 1) > 1
    ys = torch.empty(1, cout, 64, 64, device=dev)
    pooled = torch.empty(1, cout, 32, 32, device=dev)
    codes = torch.empty(cout * 32 * 32, dtype=torch.uint8, device=dev)

    def split_conv():
        hip.conv3x3_x3q(xs, fq, wsq, bias, cout, 1, True, out=ys, workspace=ws)
        hip.conv3x3_x3q_relu_pool(xs, fq, wsq, bias, cout, 1, pooled, codes, workspace=ws)
        return [ys, pooled, codes]
    V["splitk_finish"] = (split_conv, 2 * (256 + 1024), "conv_splitk_finish_kernel, conv_splitk_finish_pool_kernel<true>")

    # ---- conv3x3_few_out (the vector-ALU form of the image layer's backward pass; the planner's default is conv_few_mfma)
    gy = torch.randn(1, 64, 512, 512, device=dev, generator=g)
    w11 = torch.randn(64, 3, 3, 3, device=dev, generator=g) * 0.1
    _, wb11 = hip.conv_pack_filters(w11)
    gx = torch.empty(1, 3, 512, 512, device=dev)

    def few_out():
        hip.conv2d_bwd_data(gy, None, wb11, w11, (1, 3, 512, 512), 3, 1, 1, out=gx)
        return [gx]
    V["few_out"] = (few_out, 1024, "conv3x3_few_out_kernel<3, .., 4 / 8>")

    # ---- the strided stem kernels (NIN's 11x11 / 4 on planes below 64 x 64 sites and other geometries)
    xi = torch.randn(1, 3, 227, 227, device=dev, generator=g)
    w_st = torch.randn(96, 3, 11, 11, device=dev, generator=g) * 0.05
    wf_st, wb_st = hip.conv_pack_filters(w_st)
    b_st = torch.randn(96, device=dev, generator=g) * 0.1
    y_st = torch.empty(1, 96, 55, 55, device=dev)
    gy_st = torch.randn(1, 96, 55, 55, device=dev, generator=g)
    gx_st = torch.empty(1, 3, 227, 227, device=dev)

    def strided():
        hip.conv2d_fwd(xi, wf_st, b_st, 11, 4, 0, True, out=y_st)
        hip.conv2d_bwd_data(gy_st, None, wb_st, w_st, (1, 3, 227, 227), 11, 4, 0, out=gx_st)
        return [y_st, gx_st]
    V["strided_stem"] = (strided, 400, "conv_strided_fwd_kernel<11, 4, 16>, conv_strided_bwd_kernel<11, 4, 3>")

    # ---- bilinear resize, Adam
    img = torch.randn(1, 3, 512, 512, device=dev, generator=g)

    def resize():
        return [hip.resize_bilinear(img, size=(724, 724))]
    V["resize"] = (resize, 724 * 724 * 3 // 256, "resize_bilinear_kernel")
    n_ad = 3 * 512 * 512
    x_ad0, g_ad = torch.randn(n_ad, device=dev, generator=g), torch.randn(n_ad, device=dev, generator=g)
    x_ad, m_ad, v_ad = torch.empty_like(x_ad0), torch.empty_like(x_ad0), torch.empty_like(x_ad0)

    def adam():
        x_ad.copy_(x_ad0)
        m_ad.zero_()
        v_ad.zero_()
        for step in range(1, 9):
            hip.adam_step(x_ad, g_ad, m_ad, v_ad, step, 1.0)
        return [x_ad, m_ad, v_ad]
    V["adam"] = (adam, 8 * n_ad // 256, "adam_kernel")

    # ---- per-frame kernels without packed fp32 today, on the side streams all the same: mse, tv, pools (cheap to keep under watch)
    f_a = torch.relu(torch.randn(1, 512, 64, 64, device=dev, generator=g))
    f_t = torch.relu(torch.randn(1, 512, 64, 64, device=dev, generator=g))
    gbuf = torch.empty_like(f_a)
    slot = torch.zeros(1, device=dev)
    x_tv = torch.randn(1, 3, 512, 512, device=dev, generator=g)
    g_tv = torch.empty_like(x_tv)
    slot2 = torch.zeros(1, device=dev)
    f_p = torch.relu(torch.randn(1, 64, 256, 256, device=dev, generator=g))
    p_out = torch.empty(1, 64, 128, 128, device=dev)
    p_codes = torch.empty(64 * 128 * 128, dtype=torch.uint8, device=dev)
    p_g = torch.empty_like(f_p)

    def pointwise():
        hip.mse_fwd_bwd(f_a, f_t, gbuf, 1.0 / f_a.numel(), 2.0 / f_a.numel(), False, slot, mask_grad_by_x=True)
        hip.tv_fwd_bwd(x_tv, g_tv, 1e-3, False, slot2)
        hip.pool2x2_fwd_codes(f_p, p_out, p_codes)
        hip.pool2x2_bwd_codes(p_out, p_codes, p_g, True)
        return [gbuf, slot, g_tv, slot2, p_out, p_codes, p_g]
    V["pointwise"] = (pointwise, 2048 + 768 + 2 * 4096, "mse_kernel, tv_kernel, pool2x2_fwd_codes / bwd_codes (no packed fp32 today)")
    return V


def build_aggressors(g):
    """name -> (launch(), description).  Each call enqueues ONE MFMA launch on the current stream."""
    dev = "cuda"
    A = {}
    small = torch.empty(16, dtype=torch.uint8, device=dev)

    def conv(cin, cout, H, pack, fn, **kw):
        x = torch.relu(torch.randn(1, cin, H, H, device=dev, generator=g))
        w = torch.randn(cout, cin, 3, 3, device=dev, generator=g) * math.sqrt(2.0 / (9 * cin))
        f, _, wsc = pack(w)
        y = torch.empty(1, cout, H, H, device=dev)
        return lambda: fn(x, f, wsc, None, cout, 1, True, out=y, workspace=small)
    A["x3w_half"] = (conv(256, 256, 128, hip.conv_pack_filters_x3w, hip.conv3x3_x3w),
                     "conv_x3w 256->256 @128, one pass: 256 workgroups of four 254-register waves = ONE wave per SIMD, half the registers free")
    A["x3w_full"] = (conv(128, 128, 256, hip.conv_pack_filters_x3w, hip.conv3x3_x3w), "conv_x3w 128->128 @256: 512 workgroups, two per CU")
    A["x3q"] = (conv(512, 512, 128, hip.conv_pack_filters_x3q, hip.conv3x3_x3q), "conv_x3q 512->512 @128: 256 workgroups of eight waves (fills the CU)")
    A["x3p"] = (conv(128, 128, 512, hip.conv_pack_filters_x3q, hip.conv3x3_x3p), "conv_x3p 128->128 @512: persistent, one workgroup per CU")
    f5 = torch.relu(torch.randn(1, 512, 128, 128, device=dev, generator=g))
    A["gram128"] = (lambda: hip.gram_fwd(f5, 1.0 / f5.numel(), False), "gram_x3_partial128 512 x 16384 (+ its fold / finish)")
    f96 = torch.relu(torch.randn(1, 96, 254, 254, device=dev, generator=g))
    A["gram64"] = (lambda: hip.gram_fwd(f96, 1.0 / f96.numel(), True), "gram_x3_partial 96 x 64516 (64 x 64 blocks, two workgroups per CU)")
    Ds = torch.randn(512, 512, device=dev, generator=g) * 1e-3
    Ds = Ds + Ds.t()
    gf = torch.zeros(512, 128 * 128, device=dev)
    A["conv1x1"] = (lambda: hip.gram_bwd(Ds, f5, None, gf, False, relu_mask=f5), "conv1x1_x3 (Gram backward) 512 x 16384")
    # what the product really runs beside an update kernel: the update kernels of the batch's other frames (no MFMA anywhere)
    n_l = 3 * 512 * 512
    st_l = hip.LbfgsState(n_l, 20, dev)
    x_l = torch.randn(n_l, device=dev, generator=g)
    g_l = torch.randn(n_l, device=dev, generator=g)
    A["lbfgs"] = (lambda: st_l.iterate(x_l, g_l * 1e-3 + x_l * 0.5, lr=1.0), "the L-BFGS update of another frame (3 x 512 x 512, history 20): packed fp32, no MFMA")
    return A


def describe_difference(out, ref, limit=6):
    """Where and by how much a differing run differs: per result tensor the number of differing elements, the histogram of their innermost
    index modulo 64 in four bins (the lane a row-major vector kernel gives them; the round-4 fault sat in lanes 48-63) and the ratio to the
    right value."""
    lines = []
    for k, (o, r) in enumerate(zip(out, ref)):
        if o.dtype != torch.float32:
            nd = int((o.view(torch.uint8) != r.view(torch.uint8)).sum())
            if nd:
                lines.append(f"      result {k} ({o.dtype}, {o.numel()} elements): {nd} bytes differ")
            continue
        bad = (o.view(torch.int32) != r.view(torch.int32)).flatten().nonzero().flatten()
        if bad.numel() == 0:
            continue
        inner = o.shape[-1] if o.dim() > 1 else 1 << 30
        lane = (bad % inner) % 64
        hist = [int(((lane >= a) & (lane < a + 16)).sum()) for a in (0, 16, 32, 48)]
        of, rf = o.flatten()[bad[:limit]].tolist(), r.flatten()[bad[:limit]].tolist()
        lines.append(f"      result {k} {tuple(o.shape)}: {bad.numel()} of {o.numel()} elements differ; innermost index mod 64 in [0,16) [16,32) [32,48) [48,64): {hist}; "
                     f"first flat indices {bad[:limit].tolist()}; got / want {[f'{a:.6g}/{b:.6g}' for a, b in zip(of, rf)]}")
    return lines


def soak(vname, victim, aname, aggressor, seconds, rng, diagnose=0):
    run, wgs, _ = victim
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    with torch.cuda.stream(sa):
        ref = [t.clone() for t in run()]  # alone: nothing else is running
    torch.cuda.synchronize()
    bad = torch.zeros((), dtype=torch.int64, device="cuda")
    launches = 0
    marks = []  # stream B's backlog stays bounded: the host waits for the aggressor launches of two rounds ago
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        if aggressor is not None:
            ev = torch.cuda.Event()
            ev.record(sb)
            marks.append(ev)
            if len(marks) > 2:
                marks.pop(0).synchronize()
        for _ in range(8):
            if aggressor is not None:
                with torch.cuda.stream(sb):
                    for _ in range(4):
                        aggressor()
            with torch.cuda.stream(sa):
                torch.cuda._sleep(int(rng.integers(0, 60000)))  # drift the two streams through each other's phases
                out = run()
                d = torch.zeros((), dtype=torch.bool, device="cuda")
                for o, r in zip(out, ref):
                    d |= (o.view(torch.uint8) != r.view(torch.uint8)).any()
                bad += d.long()
                if diagnose and bool(d):  # (synchronises: diagnosis mode only)
                    print(f"    {vname} x {aname}: run {launches} differs", flush=True)
                    for line in describe_difference(out, ref):
                        print(line, flush=True)
                    diagnose -= 1
            launches += 1
        sa.synchronize()  # bounded queue depth; stream B keeps its backlog while stream A is refilled
    torch.cuda.synchronize()
    return launches, launches * wgs, int(bad)


def main():
    import numpy as np
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=20.0, help="wall time per (victim, aggressor) pairing")
    ap.add_argument("--victims", default="")
    ap.add_argument("--aggressors", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--diagnose", type=int, default=0, help="describe the first N differing runs of every pairing (where, by how much)")
    ap.add_argument("--pairs", default="", help="exact pairings victim:aggressor,victim:aggressor (instead of the product of --victims and --aggressors)")
    a = ap.parse_args()
    g = torch.Generator(device="cuda").manual_seed(17)
    rng = np.random.default_rng(5)
    V, A = build_victims(g), build_aggressors(g)
    vs = [v for v in a.victims.split(",") if v] or list(V)
    ags = [x for x in a.aggressors.split(",") if x] or list(A)
    pairs = [tuple(p.split(":")) for p in a.pairs.split(",") if p]
    if pairs:
        vs = list(dict.fromkeys(v for v, _ in pairs))
        ags = list(dict.fromkeys(x for _, x in pairs))
    lib = os.environ.get("MAUA_HIP_LIB", "maua-style_amd/libmaua_hip.so")
    print(f"# soak_streams: library {lib}, {a.seconds:.0f} s per pairing", flush=True)
    rows, total_bad = [], 0
    for vn in vs:
        # control: the victim against its own repetition with no neighbour (a difference here is not a cross-stream effect)
        n, w, b = soak(vn, V[vn], "-", None, min(a.seconds, 3.0), rng)
        print(f"{vn:14s} alone          : {b} of {n} runs differ ({w / 1e6:.1f} M victim workgroups)   [{V[vn][2]}]", flush=True)
        rows.append(dict(victim=vn, aggressor=None, runs=n, victim_workgroups=w, differing=b))
        total_bad += b
        for an in ags:
            if pairs and (vn, an) not in pairs:
                continue
            n, w, b = soak(vn, V[vn], an, A[an][0], a.seconds, rng, a.diagnose)
            print(f"{vn:14s} x {an:12s} : {b} of {n} runs differ ({w / 1e6:.1f} M victim workgroups)", flush=True)
            rows.append(dict(victim=vn, aggressor=an, runs=n, victim_workgroups=w, differing=b))
            total_bad += b
    print(f"# total differing runs: {total_bad}; aggressors: " + "; ".join(f"{k} = {A[k][1]}" for k in ags), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(library=lib, seconds=a.seconds, rows=rows, total_differing=total_bad), f, indent=1)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())
