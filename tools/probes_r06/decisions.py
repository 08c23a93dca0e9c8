"""Probe (round 6): for chosen random configurations of tests/test_random_shapes_gpu.py (RANDOM_CONFIG_SCALE / RANDOM_SHAPES_BASE in the
environment, seeds on the command line) - how far the engine's pixel gradient and the fp32 oracle's are from the fp64 oracle's, and how many
ReLU / max-pool decisions each of the two takes differently from fp64.
    RANDOM_CONFIG_SCALE=4 RANDOM_SHAPES_BASE=2 python tools/probes_r06/decisions.py 0 9"""
import os
import sys
import tempfile

import torch
import torch.nn.functional as F

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd"), os.path.join(REPO, "tests")]
import synth  # noqa: E402
from conftest import planar_codes, product_args  # noqa: E402
import test_random_shapes_gpu as T  # noqa: E402
from oracle import OracleNet, build_spec  # noqa: E402
import models  # noqa: E402
import optim  # noqa: E402

d = tempfile.mkdtemp()
files = {"vgg19": os.path.join(d, "vgg19_synth.pth"), "nin": os.path.join(d, "nin_synth.pth")}
torch.save(synth.vgg19_state_dict(), files["vgg19"])
torch.save(synth.nin_state_dict(), files["nin"])
torch.set_num_threads(16)
for seed in [int(a) for a in sys.argv[1:]]:
    S, extra, two_styles, nin = T.draw_config(seed)
    styles = ("s.png", "t.png") if two_styles else ("s.png",)
    args = product_args(files, extra, model="nin" if nin else "vgg19", S=S, N=3, styles=styles)
    content, style, init = synth.images(S)
    style_images = [style] + ([synth.images(S, seed=77)[1]] if two_styles else [])
    optim.set_model_args(args, S)
    net, losses = models.load_model(args)
    optim.set_content_targets(net, content, args)
    optim.set_style_targets(net, style_images, args)
    if args.normalize_weights:
        for mod in net.content_losses + net.style_losses:
            mod.strength = mod.strength / max(mod.target.size())
    for m in losses:
        m.mode = "loss"
    opt = optim.PixelOptimizer(net, losses, init, args)
    _, _, grad = opt.feval()
    torch.cuda.synchronize()
    grad = grad.clone().cpu()
    sd = synth.nin_state_dict() if nin else synth.vgg19_state_dict()
    out = {}
    for dt in (torch.float64, torch.float32):
        o = OracleNet(build_spec(args), sd, dtype=dt)
        o.capture_content(content)
        o.capture_style(style_images, args.style_blend_weights)
        if args.normalize_weights:
            o.normalize_weights()
        _, _, g = o.feval(init)
        acts, aux = o._forward(init.to(dt))
        out[dt] = (g, acts, aux, o.spec)
    g64, a64, x64, spec = out[torch.float64]
    g32, a32, x32, _ = out[torch.float32]
    rel = float((grad.double() - g64).norm() / g64.norm())
    theirs = float((g32.double() - g64).norm() / g64.norm())
    # decisions of the fp32 oracle against the fp64 oracle
    o_relu = sum(int(((a > 0) ^ (b > 0)).sum()) for a, b, l in zip(a32, a64, spec) if l.kind == "relu")
    o_pool = sum(int(((i32 != i64) & (b > 0)).sum()) for i32, i64, b, l in zip(x32, x64, a64, spec) if l.kind == "pool" and i64 is not None)
    # decisions of the engine against the fp64 oracle
    e_relu = 0
    worst = 0.0
    for k, v in opt.engine.act.items():
        if k == 0 or v is None or v.is_meta:
            continue
        ev = v.cpu().double()
        cands = [oa for oa in a64 if tuple(oa.shape) == tuple(ev.shape)]
        oa = min(cands, key=lambda t: float((ev - t).norm()))
        worst = max(worst, float((ev - oa).norm() / oa.norm()))
        e_relu += int(((ev > 0) ^ (oa > 0)).sum())
    e_pool = 0
    epools = [st for st in opt.engine.steps if st.kind == "pool"]
    opools = [i for i, l in enumerate(spec) if l.kind == "pool"]
    for st, i in zip(epools, opools):
        l = spec[i]
        if l.pool_mode != "max":
            continue
        src = opt.engine.act[st.src]
        if not src.is_meta:
            eidx = F.max_pool2d(src.cpu(), l.k, l.stride, 0, ceil_mode=l.ceil, return_indices=True)[1]
        else:
            codes = planar_codes(opt.engine.pool_codes[id(st)].cpu()).long()
            hp, wp = codes.shape[2:]
            w_in = a64[i - 1].shape[3]
            eidx = (2 * torch.arange(hp).view(1, 1, hp, 1) + ((codes & 3) >> 1)) * w_in + 2 * torch.arange(wp).view(1, 1, 1, wp) + (codes & 1)
        e_pool += int(((eidx != x64[i]) & (a64[i] > 0)).sum())
    print(f"seed {seed} S={S} {'nin' if nin else 'vgg19'}: engine vs fp64 {rel:.2e} (ReLU flips {e_relu}, arg-max flips {e_pool}, worst activation {worst:.1e});  "
          f"fp32 oracle vs fp64 {theirs:.2e} (ReLU flips {o_relu}, arg-max flips {o_pool})   {' '.join(extra)}", flush=True)
    del opt, net, losses
    torch.cuda.empty_cache()
