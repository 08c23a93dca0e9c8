"""Pin the CPU oracle to fixtures produced by the unmodified reference (tools/make_golden.py).

Tolerances follow SURVEY.md §8(c): per-module losses <= 1e-5 relative, pixel gradient
<= 1e-5 rel-L2 for a CPU restatement; trajectories are judged against the fp64 arbiter with
the reference's own fp32 noise as the yardstick.
"""
import json
import os

import numpy as np
import pytest
import torch

import synth
from conftest import FEVAL_VARIANTS, GOLDEN, NIN_LAYERS, make_cfg, rel_l2
from oracle import OracleNet, build_spec, optimize
from oracle.style_oracle import loss_order


def gold(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def run_feval(cfg, content, styles, init, sd, dtype=torch.float32):
    spec = build_spec(cfg)
    net = OracleNet(spec, sd, dtype)
    net.capture_content(content)
    net.capture_style(styles, cfg.style_blend_weights)
    if cfg.normalize_weights:
        net.normalize_weights()
    total, losses, grad = net.feval(init)
    return spec, net, float(total), losses, grad


def check_feval(g, spec, total, losses, grad, tol_loss=1e-5, tol_grad=1e-5):
    order = loss_order(spec)
    names = [spec[i].name for i in order]
    assert names == list(g["loss_names"])
    for i, want in zip(order, g["loss_values"]):
        got = float(losses.get(i, 0.0))
        assert abs(got - want) <= tol_loss * max(abs(want), 1e-12), (spec[i].name, got, want)
    assert abs(total - float(g["total"])) <= tol_loss * abs(float(g["total"]))
    assert rel_l2(grad, g["grad"]) <= tol_grad


def test_synthetic_inputs_regenerate_identically():
    g = gold("feval_vgg19_S32_default")
    imgs = synth.images(32)
    np.testing.assert_allclose(np.array([synth.checksum(t) for t in imgs]), g["input_checksums"], rtol=0, atol=0)
    meta = json.load(open(os.path.join(GOLDEN, "META.json")))
    sd = synth.vgg19_state_dict()
    for k, want in meta["weights_checksum_vgg19"].items():
        assert synth.checksum(sd[k]) == want


@pytest.mark.parametrize("S,variant", [(32, v) for v in FEVAL_VARIANTS] + [(64, "default"), (64, "no_grad_norm"), (90, "default"), (130, "default")])
def test_feval_matches_reference(S, variant):
    g = gold(f"feval_vgg19_S{S}_{variant}")
    cfg = make_cfg(**FEVAL_VARIANTS[variant])
    content, style, init = synth.images(S)
    spec, net, total, losses, grad = run_feval(cfg, content, [style], init, synth.vgg19_state_dict())
    kinds = {"tv": "TVLoss", "temporal": "ContentLoss", "content": "ContentLoss", "style": "StyleLoss",
             "conv": "Conv2d", "relu": "ReLU", "pool": "MaxPool2d" if cfg.pooling == "max" else "AvgPool2d"}
    assert [kinds[l.kind] for l in spec] == list(g["module_types"])
    check_feval(g, spec, total, losses, grad)
    if "style_target_0" in g:
        sidx = [i for i, l in enumerate(spec) if l.kind == "style"]
        for k, i in enumerate(sidx):
            assert rel_l2(net.targets[i], g[f"style_target_{k}"]) <= 1e-5
        cidx = [i for i, l in enumerate(spec) if l.kind == "content"][0]
        assert rel_l2(net.targets[cidx], g["content_target_0"]) <= 1e-6


def test_feval_zero_bias_matches_survey_appendix():
    g = gold("feval_vgg19_S64_zerobias")
    content, style, init = synth.images(64)
    spec, net, total, losses, grad = run_feval(make_cfg(), content, [style], init,
                                               synth.vgg19_state_dict(bias_scale=0.0))
    check_feval(g, spec, total, losses, grad)
    assert abs(total - 1.872610e5) < 1.0  # SURVEY.md Appendix A anchor


def test_feval_fp64_arbiter():
    g = gold("feval_vgg19_S32_default_f64")
    content, style, init = synth.images(32)
    spec, net, total, losses, grad = run_feval(make_cfg(), content, [style], init, synth.vgg19_state_dict(),
                                               dtype=torch.float64)
    check_feval(g, spec, total, losses, grad, tol_loss=1e-10, tol_grad=1e-10)


def test_feval_two_styles_nonsquare():
    g = gold("feval_vgg19_40x56_twostyles")
    gen = torch.Generator().manual_seed(11)
    content = torch.rand(1, 3, 40, 56, generator=gen) * 255 - 120
    s1 = torch.rand(1, 3, 48, 48, generator=gen) * 255 - 120
    s2 = torch.rand(1, 3, 36, 60, generator=gen) * 255 - 120
    init = torch.rand(1, 3, 40, 56, generator=gen) * 255 - 120
    cfg = make_cfg(style_blend_weights=[0.25, 0.75])
    np.testing.assert_allclose(g["blend"], cfg.style_blend_weights)
    spec, net, total, losses, grad = run_feval(cfg, content, [s1, s2], init, synth.vgg19_state_dict())
    check_feval(g, spec, total, losses, grad)


def test_intermediate_features():
    g = gold("feval_vgg19_S32_default")
    content, style, init = synth.images(32)
    net = OracleNet(build_spec(make_cfg()), synth.vgg19_state_dict())
    acts, _ = net._forward(init)
    # the reference reuses ONE MaxPool2d instance for every pool (models.py:119-127), so the hook the
    # generator put on module 7 last fired for pool4 (module 34): "feat_7" holds pool4's output.
    for idx, key in ((3, "feat_3"), (34, "feat_7"), (37, "feat_37")):
        assert rel_l2(acts[idx], g[key]) <= 1e-6


# ------------------------------------------------------------------------------------------
# Trajectories.  fp32 L-BFGS is chaotic: the yardstick is the fp64 arbiter,
#   relL2(oracle_f32, ref_f64) <= max(1e-3, 2 * relL2(ref_f32, ref_f64))        
# and the oracle run in fp64 must reproduce the arbiter itself tightly.
# ------------------------------------------------------------------------------------------
TRAJ64 = [("lbfgs", n) for n in (1, 2, 3, 4, 5, 10, 20)] + [("adam", n) for n in (1, 5, 10, 20)]


@pytest.mark.parametrize("opt,N", TRAJ64)
def test_trajectory_vs_arbiter(opt, N):
    g = gold("traj_vgg19_S64")
    ref32, ref64 = g[f"{opt}_N{N}_f32"], g[f"{opt}_N{N}_f64"]
    content, style, init = synth.images(64)
    sd = synth.vgg19_state_dict()
    cfg = make_cfg(optimizer=opt)
    out64 = optimize(content, [style], init, N, cfg, sd, dtype=torch.float64)
    assert rel_l2(out64, ref64) <= 1e-7, "fp64 oracle must reproduce the fp64 arbiter"
    if N <= 10:
        out32 = optimize(content, [style], init, N, cfg, sd, dtype=torch.float32)
        floor = rel_l2(ref32, ref64)
        assert rel_l2(out32, ref64) <= max(1e-3, 2 * floor), (rel_l2(out32, ref64), floor)


def test_trajectory_variants_history_ring_and_adam_lr():
    g = gold("traj_vgg19_S32_variants")
    content, style, init = synth.images(32)
    sd = synth.vgg19_state_dict()
    out = optimize(content, [style], init, 12, make_cfg(lbfgs_num_correction=3), sd, dtype=torch.float64)
    assert rel_l2(out, g["lbfgs_m3_N12_f64"]) <= 1e-7
    out = optimize(content, [style], init, 8, make_cfg(optimizer="adam", learning_rate=2.5), sd, dtype=torch.float64)
    assert rel_l2(out, g["adam_lr2.5_N8_f64"]) <= 1e-9
    out = optimize(content, [style], init, 8, make_cfg(optimizer="adam", learning_rate=2.5), sd)
    assert rel_l2(out, g["adam_lr2.5_N8_f32"]) <= max(1e-3, 2 * rel_l2(g["adam_lr2.5_N8_f32"], g["adam_lr2.5_N8_f64"]))


def test_trajectory_variants_64():
    g = gold("traj_vgg19_S64_variants")
    content, style, init = synth.images(64)
    sd = synth.vgg19_state_dict()
    out = optimize(content, [style], init, 12, make_cfg(lbfgs_num_correction=3), sd, dtype=torch.float64)
    assert rel_l2(out, g["lbfgs_m3_N12_f64"]) <= 1e-7
    out = optimize(content, [style], init, 8, make_cfg(optimizer="adam", learning_rate=2.5), sd, dtype=torch.float64)
    assert rel_l2(out, g["adam_lr2.5_N8_f64"]) <= 1e-9


def test_lbfgs_eval_counts_match_reference():
    host = json.load(open(os.path.join(GOLDEN, "host_logic.json")))
    from oracle import lbfgs_run, adam_run
    x0 = torch.linspace(-1, 1, 50, dtype=torch.float64)

    def fg(x):
        return float(((x - 0.3) ** 4).sum() + (x * x).sum()), 4 * (x - 0.3) ** 3 + 2 * x
    for n, want in host["lbfgs_fevals"].items():
        assert lbfgs_run(fg, x0, int(n))[1] == want
    for n, want in host["adam_fevals"].items():
        assert adam_run(fg, x0, int(n))[1] == want


# ------------------------------------------------------------------------------------------
# NIN + covariance (BASELINE config 5)
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,S,cov", [("feval_nin_S128_covariance", 128, True), ("feval_nin_S128_gram", 128, False),
                                        ("feval_nin_S99_covariance", 99, True)])
def test_nin_feval(name, S, cov):
    g = gold(name)
    cfg = make_cfg(use_covariance=cov, **NIN_LAYERS)
    content, style, init = synth.images(S)
    spec, net, total, losses, grad = run_feval(cfg, content, [style], init, synth.nin_state_dict())
    check_feval(g, spec, total, losses, grad, tol_loss=2e-5, tol_grad=2e-5)
    if "style_target_0" in g:
        sidx = [i for i, l in enumerate(spec) if l.kind == "style"]
        for k in range(3):
            assert rel_l2(net.targets[sidx[k]], g[f"style_target_{k}"]) <= 1e-5


def test_nin_adam_trajectory():
    g = gold("traj_nin_S128")
    cfg = make_cfg(use_covariance=True, optimizer="adam", **NIN_LAYERS)
    content, style, init = synth.images(128)
    out = optimize(content, [style], init, 5, cfg, synth.nin_state_dict(), dtype=torch.float64)
    assert rel_l2(out, g["adam_N5_f64"]) <= 1e-9


# ---------------------------------------------------------------------------------------------------------
# the temporal path (weighted pixel-level ContentLoss, .flo warp maps)
# ---------------------------------------------------------------------------------------------------------
def temporal_inputs(S):
    """Same construction as tools/make_golden.py::temporal_inputs."""
    g = torch.Generator().manual_seed(11)
    warp = torch.rand(1, 3, S, S, generator=g) * 255 - 120
    weights = (torch.rand(1, 1, S, S, generator=g) > 0.3).float() * torch.rand(1, 1, S, S, generator=g)
    return warp, weights


@pytest.mark.parametrize("tag,over", [("default", {}), ("no_grad_norm", {"normalize_gradients": False})])
@pytest.mark.parametrize("double", [False, True])
def test_oracle_temporal_feval_matches_reference(tag, over, double):
    S = 64
    g = gold(f"feval_temporal_{tag}_S{S}" + ("_f64" if double else ""))
    cfg = make_cfg(**over)
    dtype = torch.float64 if double else torch.float32
    content, style, init = synth.images(S)
    warp, weights = temporal_inputs(S)
    net = OracleNet(build_spec(cfg), synth.vgg19_state_dict(), dtype)
    net.capture_content(content)
    net.capture_style([style], cfg.style_blend_weights)
    net.capture_temporal(warp, weights)
    total, losses, grad = net.feval(init)
    names = [l.name for l in net.spec if l.kind in ("content", "style", "tv", "temporal")]
    got = {net.spec[i].name: float(v) for i, v in losses.items()}
    tol = 1e-9 if double else 2e-5
    for name, want in zip(g["loss_names"], g["loss_values"]):
        assert abs(got[str(name)] - want) <= tol * max(1.0, abs(want)), name
    assert rel_l2(grad, torch.from_numpy(g["grad"])) <= (1e-9 if double else 2e-5)
    assert "temporal 1" in names


@pytest.mark.parametrize("opt", ["lbfgs", "adam"])
def test_oracle_temporal_trajectory_fp64(opt):
    S, N = 64, 6
    g = gold(f"traj_temporal_S{S}")
    cfg = make_cfg(optimizer=opt)
    content, style, init = synth.images(S)
    out = optimize(content, [style], init, N, cfg, synth.vgg19_state_dict(), dtype=torch.float64,
                          temporal=temporal_inputs(S))
    assert rel_l2(out, torch.from_numpy(g[f"{opt}_N{N}_f64"])) <= 1e-7


# ---------------------------------------------------------------------------------------------------------
# windows of B > 1 frames (img_vid) - static + dynamic style terms, '_vid' branches of optim.optimize
# ---------------------------------------------------------------------------------------------------------
def imgvid_inputs(S=64, T=5, TS=7):
    """Same construction as tools/make_golden.py::imgvid_inputs."""
    g = torch.Generator().manual_seed(77)
    content = torch.rand(1, 3, S, S, generator=g) * 255 - 120
    style_video = torch.rand(TS, 3, S, S, generator=g) * 255 - 120
    init = torch.rand(T, 3, S, S, generator=g) * 255 - 120
    return content, style_video, init


IMGVID_CFG = dict(style_layers="relu1_1,relu2_1", content_layers="relu2_2")


def test_oracle_style_video_targets_fp64():
    """optim.set_style_video_targets on a 7-frame clip, windows of 3: C x C and 3C x 3C targets of both style layers."""
    from oracle.style_oracle import OracleNet as Net
    g = gold("imgvid_S64")
    cfg = make_cfg(optimizer="adam", **IMGVID_CFG)
    _, style_video, _ = imgvid_inputs()
    net = Net(build_spec(cfg), synth.vgg19_state_dict(), torch.float64)
    net.capture_style_videos([style_video], cfg.style_blend_weights, 3)
    layers = [i for i, l in enumerate(net.spec) if l.kind == "style"]
    assert len(layers) == 2
    for k, i in enumerate(layers):
        assert rel_l2(net.targets[i], g[f"target_{k}"]) <= 1e-10
        vt = net.video_targets[i]
        rows, norm, trace, total = g[f"video_target_{k}_stats"]
        assert vt.shape == (int(rows), int(rows))
        assert rel_l2(vt[:48, -48:], g[f"video_target_{k}_block"]) <= 1e-10
        assert abs(float(vt.norm()) - norm) <= 1e-10 * norm and abs(float(vt.trace()) - trace) <= 1e-10 * abs(trace)
        assert abs(float(vt.sum()) - total) <= 1e-9 * norm


@pytest.mark.parametrize("opt,avg", [("lbfgs", 18), ("adam", -1)])
def test_oracle_img_vid_trajectory(opt, avg):
    """The '_vid' optimize loop (3 windows of B = 3 frames over a 5-frame clip, 4 iterations each): the fp64 oracle
    reproduces the reference's fp64 run, the fp32 oracle meets the trajectory rule."""
    from oracle.style_oracle import optimize_video
    g = gold("imgvid_S64")
    cfg = make_cfg(optimizer=opt, **IMGVID_CFG)
    content, style_video, init = imgvid_inputs()
    sd = synth.vgg19_state_dict()
    ref32, ref64 = g[f"out_{opt}_f32"], g[f"out_{opt}_f64"]
    out64 = optimize_video(content, [style_video], init, 4, cfg, sd, 3, avg_frame_window=avg, dtype=torch.float64)
    assert rel_l2(out64, ref64) <= 1e-7
    out32 = optimize_video(content, [style_video], init, 4, cfg, sd, 3, avg_frame_window=avg, dtype=torch.float32)
    floor = rel_l2(ref32, ref64)
    assert rel_l2(out32, ref64) <= max(1e-3, 2 * floor), (rel_l2(out32, ref64), floor)


# ---------------------------------------------------------------------------------------------------------
# The reference's other VGG stacks (models.py:134-137, chosen by the checkpoint's name, models.py:248-327): VGG-16 and the
# channel-pruned VGG-16 with its widths of 24, 22, 41, 51, 108, 89, 111, 184, 276, 228
# ---------------------------------------------------------------------------------------------------------
VGG16_MODELS = {"vgg16": ("vgg16_synth.pth", synth.VGG16_CHANNELS), "vgg16prune": ("vgg16-prune_synth.pth", synth.VGG16P_CHANNELS)}
VGG16_ALT = dict(use_covariance=True, pooling="avg", content_layers="relu3_3,relu5_1", style_layers="relu1_2,relu2_2,relu3_1,relu4_3")


@pytest.mark.parametrize("tag", list(VGG16_MODELS))
@pytest.mark.parametrize("name,S,over", [("S80_default", 80, {}), ("S72_covariance_avgpool_layers_alt", 72, VGG16_ALT)])
def test_vgg16_and_pruned_vgg16_feval_matches_reference(tag, name, S, over):
    g = gold(f"feval_{tag}_{name}")
    path, channels = VGG16_MODELS[tag]
    cfg = make_cfg(model_file=path, **over)
    content, style, init = synth.images(S)
    spec, net, total, losses, grad = run_feval(cfg, content, [style], init, synth.vgg19_state_dict(channels=channels))
    assert [l.cout for l in spec if l.kind == "conv"] == [c for c in channels if c != "P"][:sum(1 for l in spec if l.kind == "conv")]
    if "conv_channels" in g:
        assert [l.cout for l in spec if l.kind == "conv"] == list(g["conv_channels"])
    assert len(spec) == len(g["module_types"])
    check_feval(g, spec, total, losses, grad)


@pytest.mark.parametrize("tag", list(VGG16_MODELS))
def test_vgg16_and_pruned_vgg16_fp64_arbiter_and_trajectories(tag):
    path, channels = VGG16_MODELS[tag]
    sd = synth.vgg19_state_dict(channels=channels)
    g64 = gold(f"feval_{tag}_S80_default_f64")
    content, style, init = synth.images(80)
    spec, net, total, losses, grad = run_feval(make_cfg(model_file=path), content, [style], init, sd, dtype=torch.float64)
    check_feval(g64, spec, total, losses, grad, tol_loss=1e-10, tol_grad=1e-10)
    g = gold(f"traj_{tag}_S64")
    content, style, init = synth.images(64)
    for opt in ("lbfgs", "adam"):
        out = optimize(content, [style], init, 6, make_cfg(model_file=path, optimizer=opt), sd, dtype=torch.float64)
        assert rel_l2(out, g[f"{opt}_N6_f64"]) <= 1e-9, opt
        out32 = optimize(content, [style], init, 6, make_cfg(model_file=path, optimizer=opt), sd)
        ours, theirs = rel_l2(out32, g[f"{opt}_N6_f64"]), rel_l2(g[f"{opt}_N6_f32"], g[f"{opt}_N6_f64"])
        assert ours <= max(2.0 * theirs, 1e-3), (opt, ours, theirs)   # (the bar of test_trajectory_vs_arbiter)


def test_vgg_stack_is_chosen_by_the_checkpoint_name_like_the_reference():
    from oracle.style_oracle import VGG16_CHANNELS, VGG16P_CHANNELS, VGG19_CHANNELS, _vgg_channels
    for name, want in (("modelzoo/vgg16-prune.pth", VGG16P_CHANNELS), ("nyud-fcn32s-color-heavy.pth", VGG16_CHANNELS), ("fcn32s-heavy-pascal.pth", VGG16_CHANNELS),
                       ("vgg16-sod.pth", VGG16_CHANNELS), ("vgg19.pth", VGG19_CHANNELS), ("vgg16.pth", VGG16_CHANNELS), ("vgg19-prune.pth", VGG16P_CHANNELS),
                       ("nin.pth", None)):
        assert _vgg_channels(name) is want, name
    with pytest.raises(ValueError):
        _vgg_channels("vgg11.pth")
