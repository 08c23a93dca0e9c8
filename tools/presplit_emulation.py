"""CPU emulation for the producer-side split (profiles/probes_r06.md section 3): how much accuracy the fp16-pair
arithmetic of conv_x3q keeps when the activation's power-of-two scale is NOT chosen per (consumer tile, 32-channel chunk) from the staged
patch - what the kernels do today, and what forces every staged value through vector registers - but per (32-channel chunk, WHOLE PLANE),
which is the only granularity a producing layer can bake into ready fp16 pairs (a consumer tile's patch spans up to nine producer tiles).

Representation error only: operands are split exactly as the kernels split them (x s = hi + lo, both rounded to nearest fp16; filters
pre-scaled into [32, 64) and split the same way; products hi wh + hi wl + lo wh), the sums are accumulated exactly (fp64), so what is
measured is what the choice of scale does - fp32 accumulation noise (1.3e-7 on these shapes, the same for every scheme) comes on top.

    python tools/presplit_emulation.py [image size, default 256]

Schemes:  tile   = today (maximum of the 18 x 34 patch of a 16 x 32 tile, per 32-channel chunk, into [2^11, 2^12))
          plane  = maximum of the whole plane per 32-channel chunk, into [2^11, 2^12)
          stale4 = plane with four bits of headroom (a scale carried over from the previous evaluation: the maximum may grow 16 x)
          stale8 = eight bits of headroom
Data: the real activations of the synthetic-weight VGG-19 in front of conv1_2 / conv2_2 / conv3_2 / conv4_2 (oracle forward pass, test
infrastructure), and gradient-like data on the same shapes (Gaussian times a smooth log-normal field spanning four decades across the plane,
masked like a ReLU gradient)."""
import math
import os
import sys

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import synth  # noqa: E402
from oracle import OracleNet, build_spec  # noqa: E402  (tools may use the checker; the product never does)

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
torch.manual_seed(0)


def pow2_scale(m, top=11):
    """Power of two s with m s in [2^top, 2^(top + 1)) (m > 0), as chunk_scale() of conv_x3q.hip."""
    e = torch.floor(torch.log2(m.clamp_min(1e-38)))
    return torch.where(m > 0, torch.exp2(top - e), torch.ones_like(m))


def split16(v):
    hi = v.float().half()
    lo = (v.float() - hi.float()).half()
    return hi.double(), lo.double()


def emulate(x, w, scheme, region=64):
    """x (C, H, W) fp32, w (Cout, C, 3, 3): rel-L2 error of the fp16-pair convolution of the top-left `region` x `region` outputs (padding
    1) against fp64, and the smallest / median number of significant bits the scheme leaves to non-zero values."""
    C = x.shape[0]
    ws = 2.0 ** (5 - math.floor(math.log2(float(w.abs().max()))))
    wh, wl = split16(w * ws)
    xp = F.pad(x.double(), (1, 1, 1, 1))
    ref = F.conv2d(xp[None, :, :region + 2, :region + 2], w.double())[0]
    out = torch.zeros_like(ref)
    plane_max = x.abs().reshape(C // 32, 32, -1).amax(dim=(1, 2))  # per 32-channel chunk
    for ty in range(0, region, 16):
        for tx in range(0, region, 32):
            patch = xp[:, ty:ty + 18, tx:tx + 34]
            acc = torch.zeros(w.shape[0], 16, 32, dtype=torch.float64)
            for c in range(C // 32):
                pc = patch[32 * c:32 * c + 32]
                if scheme == "tile":
                    s = pow2_scale(pc.abs().max())
                elif scheme == "plane":
                    s = pow2_scale(plane_max[c].double())
                else:
                    s = pow2_scale(plane_max[c].double()) / 2.0 ** int(scheme[5:])
                hi, lo = split16(pc * s)
                sl = slice(32 * c, 32 * c + 32)
                t = F.conv2d(hi[None], wh[:, sl]) + F.conv2d(hi[None], wl[:, sl]) + F.conv2d(lo[None], wh[:, sl])
                acc += t[0] / (s * ws)
            out[:, ty:ty + 16, tx:tx + 32] = acc
    return float((out - ref).norm() / ref.norm())


def main():
    import argparse as ap
    cfg = ap.Namespace(model_file="vgg19", pooling="max", content_layers="relu4_2", style_layers="relu1_1,relu2_1,relu3_1,relu4_1,relu5_1",
                       tv_weight=1e-3, temporal_weight=50.0, content_weight=5.0, style_weight=100.0, use_covariance=False,
                       normalize_gradients=True, video_style_factor=100.0)
    sd = synth.vgg19_state_dict()
    net = OracleNet(build_spec(cfg), sd)
    _, _, init = synth.images(S)
    acts, _ = net._forward(init)
    convs = [(i, l) for i, l in enumerate(net.spec) if l.kind == "conv"]
    want = {"conv1_2": 1, "conv2_2": 3, "conv3_2": 5, "conv4_2": 9}
    print(f"# representation error (exact accumulation) of the fp16-pair convolution, {S} x {S} image, top-left 64 x 64 outputs")
    print(f"{'layer / data':34s} {'tile (today)':>13s} {'plane':>10s} {'stale4':>10s} {'stale8':>10s}   max / median |x| of the plane")
    for name, k in want.items():
        i, l = convs[k]
        x = acts[i - 1][0]  # the activation this convolution reads
        w = net.w[l.feat_idx]
        region = min(64, x.shape[1])
        nz = x[x > 0]
        row = [emulate(x, w, s, region) for s in ("tile", "plane", "stale4", "stale8")]
        print(f"{name + ' forward, real activations':34s} " + " ".join(f"{e:10.2e}" for e in row) +
              f"   {float(x.max()):.3g} / {float(nz.median()):.3g}")
        # backward-data geometry: gradient-like data of the output's shape against the transposed, flipped filters
        co = w.shape[0]
        g = torch.randn(co, x.shape[1], x.shape[2])
        yy, xx = torch.meshgrid(torch.linspace(0, 1, x.shape[1]), torch.linspace(0, 1, x.shape[2]), indexing="ij")
        field = torch.exp(math.log(1e4) * (0.5 * torch.sin(3.1 * yy + 1.0) * torch.cos(2.3 * xx) + 0.5 * xx))  # four decades across the plane
        g = g * field * 1e-6 * (torch.rand_like(g) > 0.5)
        wb = w.flip(2, 3).transpose(0, 1).contiguous()
        row = [emulate(g, wb, s, region) for s in ("tile", "plane", "stale4", "stale8")]
        gz = g[g != 0].abs()
        print(f"{name + ' backward, 4-decade gradient':34s} " + " ".join(f"{e:10.2e}" for e in row) +
              f"   {float(gz.max()):.3g} / {float(gz.median()):.3g}")


if __name__ == "__main__":
    main()
