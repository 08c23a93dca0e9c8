#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the UNMODIFIED reference in this container.

Build-container only: imports /root/reference (which never travels to the GPU box)
with inert stubs for its absent third-party imports, following SURVEY.md §8(c) /
Appendix A.  Nothing from the reference is copied; the fixtures hold inputs' seeds
and the reference's numeric outputs.

    python tools/make_golden.py [group ...]     # groups: feval traj nin hist host all

Thread count changes fp32 results, so everything runs with
torch.set_num_threads(1).
"""
import contextlib
import io
import json
import os
import sys
import tempfile
import types

sys.dont_write_bytecode = True
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"

for _name in ["gdown", "skvideo", "skvideo.io", "torchvision", "torchvision.transforms", "ffmpeg"]:
    sys.modules[_name] = types.ModuleType(_name)  # absent deps; none is on the hot path

import importlib.util

import numpy as np
import torch

_spec = importlib.util.spec_from_file_location("maua_synth", os.path.join(REPO, "maua-style_amd", "synth.py"))
synth = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(synth)

sys.path.insert(0, REF)
import config as ref_config  # noqa: E402  (reference modules, unmodified)
import loss as ref_loss  # noqa: E402
import models as ref_models  # noqa: E402
import optim as ref_optim  # noqa: E402
import utils as ref_utils  # noqa: E402

torch.set_num_threads(1)
TMP = tempfile.mkdtemp(prefix="maua_golden_")
VGG_PATH = os.path.join(TMP, "vgg19_synth.pth")  # path must contain "vgg19" (models.py:289)
VGG_PATH_ZB = os.path.join(TMP, "vgg19_synth_zerobias.pth")
NIN_PATH = os.path.join(TMP, "nin_synth.pth")
SCALING = os.path.join(TMP, "scaling-cpu.json")
torch.save(synth.vgg19_state_dict(), VGG_PATH)
torch.save(synth.vgg19_state_dict(bias_scale=0.0), VGG_PATH_ZB)
torch.save(synth.nin_state_dict(), NIN_PATH)
with open(SCALING, "w") as f:  # defeats optim.py:93-108 forcing gpu "0"
    json.dump({"100000": {"gpu": "c", "multidevice": False}}, f)


def get_args(extra=(), model=VGG_PATH, optimizer="lbfgs", S=64, N=10):
    argv = ["style.py", "--content", "c.png", "--style", "s.png", "--gpu", "c", "--backend", "mkl",
            "--model_file", model, "--disable_check", "--scaling_args", SCALING,
            "--ffmpeg_args", os.path.join(REF, "config", "ffmpeg-libx264.json"), "--optimizer", optimizer,
            "--image_sizes", str(S), "--num_iters", str(N), "--seed", "0", "--no_hist_match"] + list(extra)
    old = sys.argv
    sys.argv = argv
    try:
        return ref_config.get_args()
    finally:
        sys.argv = old


def quiet():
    return contextlib.redirect_stdout(io.StringIO())  # tqdm writes to stdout (optim.py:19)


def save(name, **arrays):
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"  wrote {os.path.relpath(path, REPO)}  ({os.path.getsize(path) / 1024:.0f} KB)")


def single_feval(args, content, styles, init, double=False, want_feats=()):
    """One forward+backward of the assembled loss network, reference code only."""
    ref_optim.set_model_args(args, max(*init.shape))
    with quiet():
        net, losses = ref_models.load_model(args)
    if double:
        net.double()
        args.dtype = torch.DoubleTensor
    feats = {}
    hooks = []
    for idx, mod in enumerate(net):
        if idx in want_feats:
            hooks.append(mod.register_forward_hook(
                lambda m, i, o, idx=idx: feats.__setitem__(idx, o.detach().clone())))
    with quiet():
        ref_optim.set_content_targets(net, content, args)
        ref_optim.set_style_targets(net, styles, args)
    for m in losses:
        m.mode = "loss"
    if args.normalize_weights:  # optim.py:176-178
        for m in net.content_losses + net.style_losses + net.temporal_losses:
            m.strength = m.strength / max(m.target.size())
    feats.clear()
    x = torch.nn.Parameter(init.clone().type(args.dtype))
    net(x)
    names, vals, total = [], [], 0
    for m in losses:
        names.append(m.name)
        if isinstance(m.loss, int):
            vals.append(0.0)
            continue
        vals.append(float(m.loss.detach()))
        total = total + m.loss
    total.backward()
    for h in hooks:
        h.remove()
    out = dict(
        loss_names=np.array(names), loss_values=np.array(vals, dtype=np.float64),
        total=np.float64(float(total.detach())), grad=x.grad.detach().numpy().copy(),
        module_types=np.array([type(m).__name__ for m in net]),
        module_names=np.array([getattr(m, "name", "") for m in net]),
    )
    for idx, t in feats.items():
        out[f"feat_{idx}"] = t.numpy().copy()
    return out, net


def gen_feval():
    print("[feval] single forward/backward fixtures")
    variants = {
        "default": [],
        "no_grad_norm": ["--no_grad_norm"],
        # with the default temporal_weight the reference itself dies here with ZeroDivisionError
        # (optim.py:178 divides by max(torch.Size([0])) of the empty temporal target)
        "normalize_weights": ["--normalize_weights", "--temporal_weight", "0"],
        "avgpool": ["--pooling", "avg"],
        "no_tv_no_vsf": ["--tv_weight", "0", "--video_style_factor", "0", "--temporal_weight", "0"],
        "covariance": ["--use_covariance"],
        "layers_alt": ["--content_layers", "relu3_2,relu4_2", "--style_layers", "relu1_2,relu2_2,relu3_3"],
        "weights_alt": ["--content_weight", "7.5", "--style_weight", "33", "--tv_weight", "0.02"],
    }
    for S in (32, 64):
        content, style, init = synth.images(S)
        for vname, extra in variants.items():
            if S == 64 and vname not in ("default", "no_grad_norm"):
                continue
            args = get_args(extra, S=S)
            want = (3, 7, 37) if (S == 32 and vname == "default") else ()
            out, net = single_feval(args, content, [style], init, want_feats=want)
            if S == 32 and vname in ("default", "covariance"):
                for k, m in enumerate(net.style_losses):
                    out[f"style_target_{k}"] = m.target.numpy().copy()
                out["content_target_0"] = net.content_losses[0].target.numpy().copy()
            out["input_checksums"] = np.array([synth.checksum(t) for t in (content, style, init)])
            out["flags"] = np.array(extra)
            save(f"feval_vgg19_S{S}_{vname}", **out)
    # zero-bias recipe of SURVEY Appendix A: sanity anchor (total loss 1.872610e+05 at S=64)
    content, style, init = synth.images(64)
    out, _ = single_feval(get_args(S=64, model=VGG_PATH_ZB), content, [style], init)
    print("    zero-bias S=64 total loss", out["total"], "(SURVEY Appendix A: 1.872610e+05)")
    save("feval_vgg19_S64_zerobias", **out)
    # fp64 arbiter for the default single feval
    content, style, init = synth.images(32)
    out, _ = single_feval(get_args(S=32), content, [style], init, double=True)
    save("feval_vgg19_S32_default_f64", **out)
    # two styles with blend weights, styles of other sizes than the content, non-square content
    g = torch.Generator().manual_seed(11)
    content = torch.rand(1, 3, 40, 56, generator=g) * 255 - 120
    s1 = torch.rand(1, 3, 48, 48, generator=g) * 255 - 120
    s2 = torch.rand(1, 3, 36, 60, generator=g) * 255 - 120
    init = torch.rand(1, 3, 40, 56, generator=g) * 255 - 120
    args = _get_args_two_styles(["--style_blend_weights", "0.3,0.9"], S=56)  # two names: len(style)==2
    out, net = single_feval(args, content, [s1, s2], init)
    out["blend"] = np.array(args.style_blend_weights)
    save("feval_vgg19_40x56_twostyles", **out)


def _get_args_two_styles(extra, S):
    argv = ["style.py", "--content", "c.png", "--style", "s1.png", "s2.png", "--gpu", "c", "--backend", "mkl",
            "--model_file", VGG_PATH, "--disable_check", "--scaling_args", SCALING,
            "--ffmpeg_args", os.path.join(REF, "config", "ffmpeg-libx264.json"),
            "--image_sizes", str(S), "--num_iters", "5", "--seed", "0", "--no_hist_match"] + list(extra)
    old = sys.argv
    sys.argv = argv
    try:
        return ref_config.get_args()
    finally:
        sys.argv = old


def run_traj(S, N, opt, double, extra=(), model=VGG_PATH, net_cache={}):
    args = get_args(extra, model=model, optimizer=opt, S=S, N=N)
    content, style, init = synth.images(S)
    key = (S, opt, double, tuple(extra), model)
    ref_optim.set_model_args(args, S)
    if key not in net_cache:
        with quiet():
            net, losses = ref_models.load_model(args)
        if double:
            net.double()
        net_cache[key] = (net, losses)
    net, losses = net_cache[key]
    if double:
        args.dtype = torch.DoubleTensor
    with quiet():
        out = ref_optim.optimize(content, [style], init.clone(), N, args, net, losses).detach()
    return out


def gen_traj():
    print("[traj] optimizer trajectories (fp32 1-thread and fp64 arbiter)")
    S = 64
    res = {}
    for opt, Ns in (("lbfgs", (1, 2, 3, 4, 5, 10, 20)), ("adam", (1, 5, 10, 20))):
        for N in Ns:
            for double in (False, True):
                out = run_traj(S, N, opt, double)
                key = f"{opt}_N{N}_{'f64' if double else 'f32'}"
                res[key] = out.numpy().astype(np.float64 if double else np.float32)
                init = synth.images(S)[2]
                print(f"    {key}: moved {float((out.double() - init.double()).norm() / init.double().norm()):.4e}")
    save(f"traj_vgg19_S{S}", **res)
    # small history to exercise the L-BFGS ring eviction, and Adam with another lr
    res = {}
    for double in (False, True):
        out = run_traj(32, 12, "lbfgs", double, extra=["--lbfgs_num_correction", "3"])
        res[f"lbfgs_m3_N12_{'f64' if double else 'f32'}"] = out.numpy()
        out = run_traj(32, 8, "adam", double, extra=["--learning_rate", "2.5"])
        res[f"adam_lr2.5_N8_{'f64' if double else 'f32'}"] = out.numpy()
    save("traj_vgg19_S32_variants", **res)
    # the same variants at 64^2: at 32^2 the deepest layers are 2x2, where one ReLU / max-pool decision flipped by a
    # last-bit difference moves the whole gradient by ~1e-3 (seen on the HIP path), so 32^2 is a poor yardstick
    res = {}
    for double in (False, True):
        out = run_traj(64, 12, "lbfgs", double, extra=["--lbfgs_num_correction", "3"])
        res[f"lbfgs_m3_N12_{'f64' if double else 'f32'}"] = out.numpy()
        out = run_traj(64, 8, "adam", double, extra=["--learning_rate", "2.5"])
        res[f"adam_lr2.5_N8_{'f64' if double else 'f32'}"] = out.numpy()
    save("traj_vgg19_S64_variants", **res)


def gen_nin():
    print("[nin] NIN + covariance (BASELINE config 5 at small size)")
    S = 128
    extra = ["--style_layers", "relu1,relu3,relu5,relu7,relu9,relu11", "--content_layers", "relu8", "--use_covariance"]
    content, style, init = synth.images(S)
    args = get_args(extra, model=NIN_PATH, S=S)
    out, net = single_feval(args, content, [style], init, want_feats=(3, 9))
    for k, m in enumerate(net.style_losses[:3]):  # 96^2, 96^2, 256^2; the 384^2/1024^2 ones are bulky
        out[f"style_target_{k}"] = m.target.numpy().copy()
    save(f"feval_nin_S{S}_covariance", **out)
    args = get_args(extra[:4], model=NIN_PATH, S=S)
    out, net = single_feval(args, content, [style], init)
    save(f"feval_nin_S{S}_gram", **out)
    # odd size: exercises ceil-mode pooling with a partial last window
    S2 = 99
    content, style, init = synth.images(S2)
    args = get_args(extra, model=NIN_PATH, S=S2)
    out, net = single_feval(args, content, [style], init)
    save(f"feval_nin_S{S2}_covariance", **out)
    res = {}
    for double in (False, True):
        out = run_traj(S, 5, "adam", double, extra=extra, model=NIN_PATH)
        res[f"adam_N5_{'f64' if double else 'f32'}"] = out.numpy()
    save(f"traj_nin_S{S}", **res)


def gen_hist():
    print("[hist] utils.match_histogram via the harness-side symeig shim (SURVEY §8c)")
    torch.symeig = lambda A, eigenvectors=True, upper=True: tuple(torch.linalg.eigh(A, UPLO="U" if upper else "L"))
    g = torch.Generator().manual_seed(21)
    target = torch.rand(1, 3, 24, 20, generator=g) * 255 - 120
    src1 = torch.rand(1, 3, 16, 28, generator=g) * 200 - 90
    src2 = torch.rand(1, 3, 18, 18, generator=g) * 120 - 30
    res = {"target": target.numpy(), "src1": src1.numpy(), "src2": src2.numpy()}
    for tag, srcs in (("one", [src1]), ("two", [src1, src2])):
        torch.manual_seed(1234)
        out = ref_utils.match_histogram(target.clone(), srcs, mode=True)
        res[f"out_{tag}"] = out.numpy()
        torch.manual_seed(1234)
        out = ref_utils.match_histogram(target.clone(), srcs, mode="avg")
        res[f"out_avg_{tag}"] = out.numpy()
    res["out_off"] = ref_utils.match_histogram(target.clone(), [src1], mode=False).numpy()
    # clips: a 3-frame target against a 4-frame and a 1-frame source ("avg": per target frame against the source's mean
    # frame; otherwise the whole clip at once against one random source frame - np.random, seeded here)
    clip = torch.rand(3, 3, 20, 24, generator=g) * 255 - 120
    vsrc = torch.rand(4, 3, 16, 16, generator=g) * 180 - 70
    res["clip"], res["vsrc"] = clip.numpy(), vsrc.numpy()
    for tag, mode in (("avg", "avg"), ("rand", True)):
        torch.manual_seed(77)
        np.random.seed(5)
        res[f"out_clip_{tag}"] = ref_utils.match_histogram(clip.clone(), [vsrc, src2], mode=mode).numpy()
    # a larger, strongly correlated case (natural-image-like colour statistics: cond(cov) ~ 1e3)
    base = torch.rand(1, 1, 96, 80, generator=g)
    big = torch.cat([base * 200 + torch.rand(1, 1, 96, 80, generator=g) * 8 * (k + 1) for k in range(3)], 1) - 100
    res["big"] = big.numpy()
    torch.manual_seed(99)
    res["out_big"] = ref_utils.match_histogram(big.clone(), [src1], mode=True).numpy()
    save("match_histogram", **res)
    # bilinear resizing exactly as style.img_img calls it (style.py:38-66): scale_factor form and size form
    import torch.nn.functional as F
    img = torch.rand(1, 3, 37, 53, generator=g) * 255 - 120
    rs = {"img": img.numpy()}
    for k, sf in enumerate((0.5, 0.73, 1.9, 2.0)):
        rs[f"sf_{k}"] = np.float64(sf)
        rs[f"out_sf_{k}"] = F.interpolate(img, scale_factor=sf, mode="bilinear", align_corners=False).numpy()
    for k, hw in enumerate(((74, 106), (20, 31), (37, 53), (111, 60))):
        rs[f"hw_{k}"] = np.array(hw)
        rs[f"out_hw_{k}"] = F.interpolate(img, hw, mode="bilinear", align_corners=False).numpy()
    save("resize_bilinear", **rs)


def gen_host():
    print("[host] config defaults, net assembly dumps, scaling decisions")
    host = {}

    def ns_dump(args):
        d = {}
        for k, v in vars(args).items():
            d[k] = v if isinstance(v, (int, float, str, bool, list, dict, type(None))) else repr(v)
        return d

    old_cwd = os.getcwd()
    os.chdir(REF)  # default --ffmpeg_args / --scaling_args are relative paths
    try:
        for tag, argv in {
            "defaults": ["--content", "a/b/cat.jpg", "--style", "x/s1.png", "y/s2.jpeg"],
            "lists": ["--content", "c.png", "--style", "s.png", "--image_sizes", "128,256", "--num_iters", "7,5",
                      "--style_blend_weights", "2,6", "--no_grad_norm", "--no_hist_match", "--gpu", "c"],
            "gpu_multi": ["--content", "c.png", "--style", "s.png", "--gpu", "0,1", "--backend", "nn"],
            "gpu_c_multi": ["--content", "c.png", "--style", "s.png", "--gpu", "c,0", "--backend", "mkl"],
            "load_args_vid": ["--content", "v.mp4", "--style", "s.png", "--load_args", "config/args-vid.json",
                              "--num_iters", "40,20,10,8,4"],
        }.items():
            sys.argv = ["style.py"] + argv
            try:
                host[f"args_{tag}"] = ns_dump(ref_config.get_args())
            except Exception as e:  # pragma: no cover - recorded so the mirror can match the failure mode
                host[f"args_{tag}"] = {"__error__": type(e).__name__, "__msg__": str(e)}
            host[f"argv_{tag}"] = argv
    finally:
        os.chdir(old_cwd)
    # set_model_args decisions with the stock scaling table
    decisions = {}
    for size in (256, 512, 1024, 1456, 1457, 2048, 2448, 2449, 3000, 3760, 5000, 6000):
        for gpus in ("0", "0,1"):
            a = types.SimpleNamespace(scaling_args=os.path.join(REF, "config", "scaling-img.json"), gpu=gpus,
                                      model_file="vgg19", optimizer="lbfgs", multidevice=False)
            ref_optim.set_model_args(a, size)
            decisions[f"{size}|{gpus}"] = {k: v for k, v in vars(a).items() if k != "scaling_args"}
    host["set_model_args"] = decisions
    # assembled loss-network dumps
    nets = {}
    for tag, extra, model in (
        ("default", [], VGG_PATH),
        ("no_tv_temporal", ["--tv_weight", "0", "--temporal_weight", "0"], VGG_PATH),
        ("conv_named", ["--content_layers", "conv2_2", "--style_layers", "conv1_1,relu3_1"], VGG_PATH),
        ("deep", ["--content_layers", "relu5_2", "--style_layers", "relu5_4"], VGG_PATH),
        ("nin", ["--style_layers", "relu1,relu3,relu5,relu7,relu9,relu11", "--content_layers", "relu8"], NIN_PATH),
    ):
        args = get_args(extra, model=model)
        with quiet():
            net, losses = ref_models.load_model(args)
        mods = []
        for m in net:
            d = {"type": type(m).__name__, "name": getattr(m, "name", None)}
            if isinstance(m, torch.nn.Conv2d):
                d.update(cin=m.in_channels, cout=m.out_channels, k=list(m.kernel_size), stride=list(m.stride),
                         pad=list(m.padding))
            if isinstance(m, (torch.nn.MaxPool2d, torch.nn.AvgPool2d)):
                d.update(k=m.kernel_size, stride=m.stride, pad=m.padding, ceil=m.ceil_mode)
            if hasattr(m, "strength"):
                d.update(strength=m.strength, normalize=getattr(m, "normalize", None))
            mods.append(d)
        nets[tag] = {"modules": mods, "losses": [m.name for m in losses],
                     "content": [m.name for m in net.content_losses], "style": [m.name for m in net.style_losses],
                     "tv": [m.name for m in net.tv_losses], "temporal": [m.name for m in net.temporal_losses]}
    host["nets"] = nets
    # L-BFGS eval counts: number of fevals for N iterations
    counts = {}
    for N in (1, 2, 3, 4, 5, 8):
        args = get_args(S=32, N=N)
        content, style, init = synth.images(32)
        calls = [0]
        ref_optim.set_model_args(args, 32)
        with quiet():
            net, losses = ref_models.load_model(args)
        fwd = net.forward

        def counting(x, fwd=fwd):
            calls[0] += 1
            return fwd(x)
        net.forward = counting
        with quiet():
            ref_optim.optimize(content, [style], init.clone(), N, args, net, losses)
        counts[str(N)] = calls[0] - 2  # minus the content and style capture passes
    host["lbfgs_fevals"] = counts
    counts = {}
    for N in (1, 3):
        args = get_args(S=32, N=N, optimizer="adam")
        content, style, init = synth.images(32)
        calls = [0]
        with quiet():
            net, losses = ref_models.load_model(args)
        fwd = net.forward

        def counting(x, fwd=fwd):
            calls[0] += 1
            return fwd(x)
        net.forward = counting
        with quiet():
            ref_optim.optimize(content, [style], init.clone(), N, args, net, losses)
        counts[str(N)] = calls[0] - 2
    host["adam_fevals"] = counts
    path = os.path.join(GOLD, "host_logic.json")
    with open(path, "w") as f:
        json.dump(host, f, indent=1, sort_keys=True, default=repr)
    print(f"  wrote {os.path.relpath(path, REPO)}")


def gen_traj_variants64():
    """Only the 64^2 variants block of gen_traj (added later; avoids regenerating the big trajectory file)."""
    res = {}
    for double in (False, True):
        out = run_traj(64, 12, "lbfgs", double, extra=["--lbfgs_num_correction", "3"])
        res[f"lbfgs_m3_N12_{'f64' if double else 'f32'}"] = out.numpy()
        out = run_traj(64, 8, "adam", double, extra=["--learning_rate", "2.5"])
        res[f"adam_lr2.5_N8_{'f64' if double else 'f32'}"] = out.numpy()
    save("traj_vgg19_S64_variants", **res)


def _install_torchvision_standins():
    """Functional stand-ins with torchvision's documented semantics for the four transforms reference load.py uses
    (SURVEY.md §8c): Lambda, Normalize, ToTensor (uint8 HWC -> float CHW / 255), ToPILImage (mul(255).byte())."""
    from PIL import Image
    T = sys.modules["torchvision.transforms"]

    class Lambda:
        def __init__(self, f):
            self.f = f

        def __call__(self, x):
            return self.f(x)

    class Normalize:
        def __init__(self, mean, std):
            self.mean, self.std = torch.tensor(mean), torch.tensor(std)

        def __call__(self, x):
            return (x - self.mean[:, None, None]) / self.std[:, None, None]

    class ToTensor:
        def __call__(self, img):
            if isinstance(img, np.ndarray):
                return torch.from_numpy(img.transpose(2, 0, 1).copy())
            arr = np.asarray(img, dtype=np.uint8)
            if arr.ndim == 2:  # mode "L": torchvision returns (1, H, W)
                arr = arr[:, :, None]
            return torch.from_numpy(arr.copy()).permute(2, 0, 1).float() / 255

    class ToPILImage:
        def __call__(self, t):
            return Image.fromarray(t.mul(255).byte().permute(1, 2, 0).numpy(), mode="RGB")

    T.Lambda, T.Normalize, T.ToTensor, T.ToPILImage = Lambda, Normalize, ToTensor, ToPILImage


def gen_cli():
    print("[cli] BASELINE config 1 through the reference's own style.img_img (256 px, 50 L-BFGS iterations, CPU)")
    import shutil
    _install_torchvision_standins()
    sys.modules["flow"] = types.ModuleType("flow")
    import load as ref_load  # noqa: F401
    import style as ref_style
    cpng, spng = os.path.join(REPO, "tests", "synth_content_256.png"), os.path.join(REPO, "tests", "synth_style_256.png")
    outdir = os.path.join(TMP, "out")
    os.makedirs(outdir, exist_ok=True)
    argv = ["style.py", "--content", cpng, "--style", spng, "--image_sizes", "256", "--num_iters", "50", "--gpu", "c",
            "--backend", "mkl", "--model_file", VGG_PATH, "--disable_check", "--scaling_args", SCALING, "--ffmpeg_args",
            os.path.join(REF, "config", "ffmpeg-libx264.json"), "--seed", "0", "--no_hist_match", "--init", "content",
            "--output_dir", outdir]
    old = sys.argv
    sys.argv = argv
    try:
        args = ref_config.get_args()
    finally:
        sys.argv = old
    res = {}
    pre_c, pre_s = ref_load.preprocess(cpng), ref_load.preprocess(spng)
    res["pre_content_checksum"] = np.array(synth.checksum(pre_c))
    res["pre_content_corner"] = pre_c[0, :, :4, :4].numpy()
    g = torch.Generator().manual_seed(31)
    probe = torch.rand(1, 3, 16, 16, generator=g) * 300 - 150  # exercises the clamp on both sides
    res["deprocess_probe_in"] = probe.numpy()
    res["deprocess_probe_out"] = np.asarray(ref_load.deprocess(probe.clone()))
    # --original_colors: luminance of the result, chroma of the content (load.py:236-240), different sizes on purpose
    from PIL import Image
    gc = torch.Generator().manual_seed(33)
    img_c = Image.fromarray((torch.rand(24, 40, 3, generator=gc) * 255).byte().numpy())
    img_g = Image.fromarray((torch.rand(32, 48, 3, generator=gc) * 255).byte().numpy())
    res["origcol_content"], res["origcol_generated"] = np.asarray(img_c), np.asarray(img_g)
    res["origcol_out"] = np.asarray(ref_load.original_colors(img_c, img_g))
    captured = {}
    real_opt = ref_optim.optimize

    def spy(*a, **k):
        out = real_opt(*a, **k)
        captured["out"] = out.detach().clone()
        return out
    ref_optim.optimize = spy
    torch.manual_seed(args.seed)
    with quiet():
        ref_style.img_img(args)
    ref_optim.optimize = real_opt
    png = os.path.join(outdir, "synth_content_256_synth_style_256_256.png")
    shutil.copy(png, os.path.join(GOLD, "cli_config1_ref.png"))
    res["out_f32"] = captured["out"].numpy()
    # fp64 arbiter on the same preprocessed inputs
    ref_optim.set_model_args(args, 256)
    with quiet():
        net, losses = ref_models.load_model(args)
    net.double()
    args.dtype = torch.DoubleTensor
    with quiet():
        out64 = ref_optim.optimize(pre_c, [pre_s], pre_c.clone(), 50, args, net, losses).detach()
    res["out_f64"] = out64.numpy().astype(np.float32)  # stored in fp32: the comparison tolerance is >= 1e-3
    print("    f32 vs f64 rel-L2 after 50 iterations:", float((captured["out"].double() - out64).norm() / out64.norm()))
    save("cli_config1", **res)


def write_video_fixture(root, write_flow, n_frames=3, S=64):
    """Frames + the flow cache the reference's vid_img expects under <output_dir>/<content>_<style>/flow/ (normally
    written by its flow networks): smooth synthetic fields and reliability masks for every ordered pair of frames.
    Same construction in tests/conftest.py::write_video_fixture (the product's writer is byte-identical, see
    tests/test_load_and_dist_cpu.py)."""
    from PIL import Image
    fdir = os.path.join(root, "clip")
    os.makedirs(fdir, exist_ok=True)
    names = []
    base = torch.rand(S + 8, S + 8, 3, generator=torch.Generator().manual_seed(21))
    for i in range(n_frames):  # a slowly translating random texture
        frame = (base[i * 2:i * 2 + S, i * 3:i * 3 + S] * 255).byte().numpy()
        names.append("%04d" % i)
        Image.fromarray(frame).save(os.path.join(fdir, names[-1] + ".png"))
    flow_dir = os.path.join(root, "out", "clip_synth_style_256", "flow")
    os.makedirs(flow_dir, exist_ok=True)
    g = torch.Generator().manual_seed(22)
    yy, xx = torch.meshgrid(torch.linspace(0, 3.14159, S), torch.linspace(0, 3.14159, S), indexing="ij")
    for a in names:
        for b in names:
            if a == b:
                continue
            for direction in ("forward", "backward"):
                amp = (torch.rand(2, generator=g) * 4 - 2)
                flow = torch.stack([amp[0] * torch.sin(yy) * torch.cos(xx), amp[1] * torch.cos(yy) * torch.sin(xx)], dim=2)
                write_flow(flow.numpy().astype(np.float32), os.path.join(flow_dir, f"{direction}_{a}_{b}.flo"))
                rel = ((torch.rand(S, S, generator=g) > 0.2).float() * 255).byte().numpy()
                Image.fromarray(rel, mode="L").save(os.path.join(flow_dir, f"{direction}_{a}_{b}.png"))
    return fdir, os.path.join(root, "out")


def gen_vid():
    """SURVEY 8(f)-3 end to end: the reference's own style.vid_img over precomputed flow files.  Only the flow
    ESTIMATION (flow.get_flow_model / load.process_content_video) and the final ffmpeg call are stubbed."""
    print("[vid] style.vid_img with a precomputed flow cache (3 frames, 64 px, 2 passes)")
    _install_torchvision_standins()
    flow_stub = types.ModuleType("flow")
    flow_stub.get_flow_model = lambda args: None
    sys.modules["flow"] = flow_stub

    class _Chain:
        def __getattr__(self, _):
            return lambda *a, **k: self
    sys.modules["ffmpeg"].input = lambda *a, **k: _Chain()
    import load as ref_load
    import style as ref_style
    root = os.path.join(TMP, "vid")
    fdir, outdir = write_video_fixture(root, ref_load.write_flow)
    ref_load.process_content_video = lambda flow_model, args: sorted(os.path.join(fdir, f) for f in os.listdir(fdir))
    spng = os.path.join(REPO, "tests", "synth_style_256.png")
    argv = ["style.py", "--transfer_type", "vid_img", "--content", fdir, "--style", spng, "--image_sizes", "64", "--num_iters",
            "8", "--passes_per_scale", "2", "--init", "prev_warp", "--gpu", "c", "--backend", "mkl", "--model_file", VGG_PATH,
            "--disable_check", "--scaling_args", SCALING, "--ffmpeg_args", os.path.join(REF, "config", "ffmpeg-libx264.json"),
            "--seed", "0", "--no_hist_match", "--output_dir", outdir]
    old = sys.argv
    sys.argv = argv
    try:
        args = ref_config.get_args()
    finally:
        sys.argv = old
    calls = []
    real_opt, real_tt = ref_optim.optimize, ref_optim.set_temporal_targets

    arb = {}

    def spy(content, styles, init, n, a, net=None, losses=None):
        init0 = init.detach().clone()  # the reference optimises `init` in place (nn.Parameter(init.type(same dtype)))
        tmod = net.temporal_losses[0]
        tstate = (tmod.target.detach().clone(), None if tmod.weights is None else tmod.weights.detach().clone())
        out = real_opt(content, styles, init, n, a, net, losses)
        # fp64 arbiter of THIS call: same inputs, same temporal state, the reference's optimize on a double network
        if "net" not in arb:
            arb["net"], arb["losses"] = ref_models.load_model(a)
            arb["net"].double()
        m64 = arb["net"].temporal_losses[0]
        m64.target = tstate[0].double()
        m64.weights = None if tstate[1] is None else tstate[1].double()
        a.dtype = torch.DoubleTensor
        out64 = real_opt(content, styles, init0.clone(), n, a, arb["net"], arb["losses"]).detach().clone()
        a.dtype = torch.FloatTensor
        calls.append((os.path.basename(a.output), init0, out.detach().clone(), out64, tstate))
        return out
    n_tt = [0]

    def spy_tt(*a, **k):
        n_tt[0] += 1
        return real_tt(*a, **k)
    ref_optim.optimize, ref_optim.set_temporal_targets = spy, spy_tt
    torch.manual_seed(args.seed)
    with quiet():
        ref_style.vid_img(args)
    ref_optim.optimize, ref_optim.set_temporal_targets = real_opt, real_tt
    from PIL import Image
    res = {"order": np.array([c[0] for c in calls]), "temporal_target_calls": np.array(n_tt[0])}
    for fname, init, out, out64, tstate in calls:
        res["init_" + fname] = init.numpy()
        res["out_" + fname] = out.numpy()
        res["out64_" + fname] = out64.numpy().astype(np.float32)  # compared at >= 1e-3
        if tstate[0].nelement():
            res["ttarget_" + fname] = tstate[0].numpy()
            res["tweights_" + fname] = tstate[1].numpy()
        png = os.path.join(outdir, "clip_synth_style_256", "64", fname)
        res["png_" + fname] = np.asarray(Image.open(png))
        print(f"    {fname}: f32 vs f64 of this call {float((out.double() - out64).norm() / out64.norm()):.3e}")
    print("    optimize calls:", [c[0] for c in calls], " set_temporal_targets calls:", n_tt[0])
    save("vid_flow_S64", **res)


def temporal_inputs(S):
    """Synthetic stand-ins for what vid_img feeds the temporal loss: the previous result warped by the flow (an image in
    the preprocessed range) and the flow-reliability mask in [0, 1] (style.py:272-281)."""
    g = torch.Generator().manual_seed(11)
    warp = torch.rand(1, 3, S, S, generator=g) * 255 - 120
    weights = (torch.rand(1, 1, S, S, generator=g) > 0.3).float() * torch.rand(1, 1, S, S, generator=g)
    return warp, weights


def gen_temporal():
    """pixel-level weighted ContentLoss ('temporal') set by optim.set_temporal_targets, and the flow-file
    reader load.flow_warp_map.  Reference code only; flow estimation itself is out of scope (files are synthetic)."""
    print("[temporal] weighted temporal ContentLoss + .flo warp maps")
    S = 64
    content, style, init = synth.images(S)
    warp, weights = temporal_inputs(S)
    for tag, extra in (("default", []), ("no_grad_norm", ["--no_grad_norm"])):
        for double in (False, True):
            args = get_args(extra, S=S)
            ref_optim.set_model_args(args, S)
            with quiet():
                net, losses = ref_models.load_model(args)
            if double:
                net.double()
                args.dtype = torch.DoubleTensor
            with quiet():
                ref_optim.set_content_targets(net, content, args)
                ref_optim.set_style_targets(net, [style], args)
                ref_optim.set_temporal_targets(net, warp, warp_weights=weights, args=args)
            for m in losses:
                m.mode = "loss"
            x = torch.nn.Parameter(init.clone().type(args.dtype))
            net(x)
            names, vals, total = [], [], 0
            for m in losses:
                names.append(m.name)
                vals.append(0.0 if isinstance(m.loss, int) else float(m.loss.detach()))
                if not isinstance(m.loss, int):
                    total = total + m.loss
            total.backward()
            suffix = "_f64" if double else ""
            save(f"feval_temporal_{tag}_S{S}{suffix}", loss_names=np.array(names), loss_values=np.array(vals, dtype=np.float64),
                 total=np.float64(float(total.detach())), grad=x.grad.detach().numpy().copy(),
                 temporal_target=net.temporal_losses[0].target.numpy().copy())
            print(f"    {tag}{suffix}: losses {dict(zip(names, vals))}")
    # trajectory with the temporal target in place (prebuilt net, as vid_img calls optimize)
    res = {}
    for opt, N in (("lbfgs", 6), ("adam", 6)):
        for double in (False, True):
            args = get_args([], optimizer=opt, S=S, N=N)
            ref_optim.set_model_args(args, S)
            with quiet():
                net, losses = ref_models.load_model(args)
            if double:
                net.double()
                args.dtype = torch.DoubleTensor
            with quiet():
                ref_optim.set_temporal_targets(net, warp, warp_weights=weights, args=args)
                out = ref_optim.optimize(content, [style], init.clone(), N, args, net, losses).detach()
            res[f"{opt}_N{N}_{'f64' if double else 'f32'}"] = out.numpy()
    save(f"traj_temporal_S{S}", **res)
    # .flo reader: synthetic smooth flow written with the reference's writer, read back by its reader at two sizes
    import tempfile
    import load as ref_load
    g = torch.Generator().manual_seed(12)
    h, w = 40, 56
    flow = (torch.rand(h, w, 2, generator=g) * 6 - 3).numpy().astype(np.float32)
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "a.flo")
        ref_load.write_flow(flow, path)
        raw = np.fromfile(path, dtype=np.uint8)
        maps = {f"warp_{hh}x{ww}": ref_load.flow_warp_map(path, (hh, ww)).numpy() for hh, ww in ((40, 56), (64, 80))}
    img = torch.rand(1, 3, 64, 80, generator=g) * 255 - 120
    warped = torch.nn.functional.grid_sample(img, torch.from_numpy(maps["warp_64x80"]), padding_mode="border")
    save("flow_warp_map", flow=flow, flo_file_bytes=raw, image=img.numpy(), warped=warped.numpy(), **maps)


def batch_inputs(B=2, C=16, H=12, W=10):
    g = torch.Generator().manual_seed(41)
    style_feats = torch.relu(torch.randn(B, C, H, W, generator=g))
    feats = torch.relu(torch.randn(B, C, H, W, generator=g))
    content_feats = torch.relu(torch.randn(1, C, H, W, generator=g))
    return style_feats, feats, content_feats


def gen_batch():
    """SURVEY 8(f)-4 at module level: StyleLoss static + dynamic terms and ContentLoss on a batch of B = 2 frames
    (loss.py:32-64, 141-181), the reference's modules driven directly."""
    print("[batch] loss modules on B = 2 frames")
    style_feats, feats, content_feats = batch_inputs()
    res = {}
    for cov in (False, True):
        for norm in (False, True):
            m = ref_loss.StyleLoss(100.0, use_covariance=cov, normalize=norm, video_style_factor=100)
            m.name, m.blend_weight = "style 4", 1.0
            m.mode = "capture"
            m(style_feats)
            m.mode = "loss"
            m.loss = 0
            x = feats.clone().requires_grad_(True)
            m(x)
            m.loss.backward()
            tag = f"cov{int(cov)}_norm{int(norm)}"
            res[f"style_loss_{tag}"] = np.float64(float(m.loss.detach()))
            res[f"style_grad_{tag}"] = x.grad.numpy().copy()
            res[f"style_target_{tag}"] = m.target.numpy().copy()
            res[f"style_video_target_{tag}"] = m.video_target.numpy().copy()
            print(f"    {tag}: loss {float(m.loss.detach()):.6e}  target {tuple(m.target.shape)}  video_target {tuple(m.video_target.shape)}")
    for norm in (False, True):
        c = ref_loss.ContentLoss(5.0, normalize=norm)
        c.name = "cont 29"
        c.mode = "capture"
        c(content_feats)
        c.mode = "loss"
        x = feats.clone().requires_grad_(True)
        c(x)
        c.loss.backward()
        res[f"content_loss_norm{int(norm)}"] = np.float64(float(c.loss.detach()))
        res[f"content_grad_norm{int(norm)}"] = x.grad.numpy().copy()
    gm = ref_loss.GramMatrix()
    res["gram_b2"] = gm(feats).numpy().copy()
    res["gram_b2_cov"] = gm(feats, use_covariance=True).numpy().copy()
    save("loss_modules_B2", **res)


def gen_traj_extra():
    """More trajectories: NIN + covariance under L-BFGS (config 5's optimiser), and VGG-19 with --pooling avg."""
    print("[traj_extra] NIN L-BFGS, average pooling")
    res = {}
    nin_extra = ["--style_layers", "relu1,relu3,relu5,relu7,relu9,relu11", "--content_layers", "relu8", "--use_covariance"]
    for double in (False, True):
        tag = "f64" if double else "f32"
        res[f"nin_lbfgs_N6_{tag}"] = run_traj(128, 6, "lbfgs", double, extra=nin_extra, model=NIN_PATH).numpy()
        res[f"avgpool_lbfgs_N6_{tag}"] = run_traj(64, 6, "lbfgs", double, extra=["--pooling", "avg"]).numpy()
        res[f"avgpool_adam_N6_{tag}"] = run_traj(64, 6, "adam", double, extra=["--pooling", "avg"]).numpy()
    for k in sorted(res):
        if k.endswith("f32"):
            a, b = torch.from_numpy(res[k]).double(), torch.from_numpy(res[k[:-3] + "f64"])
            print(f"    {k[:-4]}: f32 vs f64 {float((a - b).norm() / b.norm()):.3e}")
    save("traj_extra", **res)


def imgvid_inputs(S=64, T=5, TS=7):
    """Seeded stand-ins for img_vid's tensors: one content image, one style video of TS frames, a pastiche of T frames
    (uniform noise in the preprocessed range, like synth.images)."""
    g = torch.Generator().manual_seed(77)
    content = torch.rand(1, 3, S, S, generator=g) * 255 - 120
    style_video = torch.rand(TS, 3, S, S, generator=g) * 255 - 120
    init = torch.rand(T, 3, S, S, generator=g) * 255 - 120
    return content, style_video, init


def gen_imgvid():
    """SURVEY 8(f)-4 at workflow level: optim.optimize with transfer_type img_vid (optim.py:111-255 with the '_vid'
    branches: window schedule, per-window style-video targets, overlap-gradient masking, wrapped write-back) and
    optim.set_style_video_targets (optim.py:69-90), on B = gram_frame_window = 3 frames."""
    print("[imgvid] sliding-window video optimisation, B = 3")
    S, N = 64, 4
    layers = ["--style_layers", "relu1_1,relu2_1", "--content_layers", "relu2_2"]
    res = {}
    for opt, extra in (("lbfgs", []), ("adam", ["--avg_frame_window", "-1"])):
        for double in (False, True):
            args = get_args(["--transfer_type", "img_vid"] + layers + extra, optimizer=opt, S=S, N=N)
            args.gram_frame_window = 3  # style.img_vid sets the integer per scale (style.py:111)
            content, style_video, init = imgvid_inputs(S)
            ref_optim.set_model_args(args, S)
            with quiet():
                net, losses = ref_models.load_model(args)
            if double:
                net.double()
                args.dtype = torch.DoubleTensor
            with quiet():
                out = ref_optim.optimize(content, [style_video], init.double() if double else init.clone(), N, args, net, losses).detach()
            tag = f"{opt}_{'f64' if double else 'f32'}"
            res[f"out_{tag}"] = out.numpy()
            print(f"    {tag}: moved {float((out.double() - init.double()).norm() / init.double().norm()):.4e}")
            if opt == "adam" and double:  # targets as set_style_video_targets leaves them (captured once: avg window -1)
                for k, m in enumerate(net.style_losses):
                    res[f"target_{k}"] = m.target.numpy().copy()
                    vt = m.video_target
                    res[f"video_target_{k}_block"] = vt[:48, -48:].numpy().copy()  # a corner across two frames' channels
                    res[f"video_target_{k}_stats"] = np.array([vt.shape[0], float(vt.norm()), float(vt.trace()), float(vt.sum())])
    for opt in ("lbfgs", "adam"):
        a, b = torch.from_numpy(res[f"out_{opt}_f32"]).double(), torch.from_numpy(res[f"out_{opt}_f64"])
        print(f"    {opt}: f32 vs f64 {float((a - b).norm() / b.norm()):.3e}")
    save("imgvid_S64", **res)


def gen_feval_odd():
    """Odd planes behind the floor-mode pools (the reference's default --image_sizes 724 / 1448 give 181 -> 90): S = 90
    (90, 45, 22, 11, 5: odd planes pooled twice) and S = 130 (130, 65, 32, 16, 8: an odd plane LARGE enough for the fused conv + ReLU +
    pool / unpooling launches of the wide kernels), default flags, fp32 and the fp64 arbiter."""
    print("[feval_odd] single forward/backward fixtures at S = 90 and S = 130")
    for S in (90, 130):
        content, style, init = synth.images(S)
        for double in (False, True):
            out, _ = single_feval(get_args([], S=S), content, [style], init, double=double)
            out["input_checksums"] = np.array([synth.checksum(t) for t in (content, style, init)])
            save(f"feval_vgg19_S{S}_default" + ("_f64" if double else ""), **out)


def gen_vgg16():
    """The reference's other VGG stacks (models.py:134-137, selected by the checkpoint's NAME, models.py:248-327): VGG-16 and the
    channel-pruned VGG-16 ("prun": 24, 22, 41, 51, 108, 89, 111, 184, 276, 228 channels - multiples of nothing).  Single evaluations at
    S = 80 (planes 80 / 40 / 20 / 10 / 5), fp32 and the fp64 arbiter, and short trajectories of both optimisers at S = 64."""
    print("[vgg16] VGG-16 and pruned VGG-16 fixtures")
    paths = {"vgg16": (os.path.join(TMP, "vgg16_synth.pth"), synth.VGG16_CHANNELS),
             "vgg16prune": (os.path.join(TMP, "vgg16-prune_synth.pth"), synth.VGG16P_CHANNELS)}
    for tag, (path, channels) in paths.items():
        torch.save(synth.vgg19_state_dict(channels=channels), path)
        S = 80
        content, style, init = synth.images(S)
        for double in (False, True):
            out, net = single_feval(get_args([], model=path, S=S), content, [style], init, double=double)
            out["conv_channels"] = np.array([m.out_channels for m in net if isinstance(m, torch.nn.Conv2d)])
            save(f"feval_{tag}_S{S}_default" + ("_f64" if double else ""), **out)
        out, _ = single_feval(get_args(["--use_covariance", "--pooling", "avg", "--content_layers", "relu3_3,relu5_1", "--style_layers",
                                        "relu1_2,relu2_2,relu3_1,relu4_3"], model=path, S=72), *[[t] if i == 1 else t for i, t in enumerate(synth.images(72))])
        save(f"feval_{tag}_S72_covariance_avgpool_layers_alt", **out)
        res = {}
        for opt in ("lbfgs", "adam"):
            for double in (False, True):
                res[f"{opt}_N6_{'f64' if double else 'f32'}"] = run_traj(64, 6, opt, double, model=path).numpy()
        save(f"traj_{tag}_S64", **res)


GROUPS = {"vgg16": gen_vgg16, "feval_odd": gen_feval_odd, "imgvid": gen_imgvid, "traj_extra": gen_traj_extra, "batch": gen_batch, "vid": gen_vid, "temporal": gen_temporal, "cli": gen_cli, "traj64v": gen_traj_variants64, "feval": gen_feval, "traj": gen_traj, "nin": gen_nin, "hist": gen_hist, "host": gen_host}

if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    want = sys.argv[1:] or ["all"]
    if "all" in want:
        want = [g for g in GROUPS if g not in ("traj64v", "cli", "vid", "feval_odd", "vgg16")] + ["cli", "vid"]
    for gname in want:
        GROUPS[gname]()
    meta = {"torch": torch.__version__, "threads": 1, "numpy": np.__version__,
            "weights_checksum_vgg19": {k: synth.checksum(v) for k, v in list(synth.vgg19_state_dict().items())[:4]}}
    with open(os.path.join(GOLD, "META.json"), "w") as f:
        json.dump(meta, f, indent=1)
