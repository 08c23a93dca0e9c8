import argparse
import os
import sys

import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "maua-style_amd")
GOLDEN = os.path.join(REPO, "tests", "golden")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

torch.set_num_threads(1)  # thread count changes fp32 results; fixtures used 1


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(autouse=True)
def _library_state_is_per_test():
    """The split-K policy of the convolutions follows a process-wide hint (maua_set_split_batch_hint: frames the caller plans per
    launch); tests that plan frame batches must not change what later tests measure."""
    yield
    if torch.cuda.is_available():
        import hip
        hip.set_split_batch_hint(1)


def make_cfg(**over):
    """Namespace with the reference's defaults for the fields the hot path reads (config.py:15-89)."""
    d = dict(
        model_file="vgg19", pooling="max", content_layers="relu4_2",
        style_layers="relu1_1,relu2_1,relu3_1,relu4_1,relu5_1", tv_weight=1e-3, temporal_weight=50.0,
        content_weight=5.0, style_weight=100.0, use_covariance=False, normalize_gradients=True,
        video_style_factor=100.0, normalize_weights=False, optimizer="lbfgs", learning_rate=1.0,
        lbfgs_num_correction=100, lbfgs_tolerance_change=-1, lbfgs_tolerance_grad=-1,
        style_blend_weights=[1.0], shift_factor=0.0,
    )
    d.update(over)
    return argparse.Namespace(**d)


# flag sets of the golden single-feval variants (tools/make_golden.py gen_feval)
FEVAL_VARIANTS = {
    "default": {},
    "no_grad_norm": dict(normalize_gradients=False),
    "normalize_weights": dict(normalize_weights=True, temporal_weight=0.0),
    "avgpool": dict(pooling="avg"),
    "no_tv_no_vsf": dict(tv_weight=0.0, video_style_factor=0.0, temporal_weight=0.0),
    "covariance": dict(use_covariance=True),
    "layers_alt": dict(content_layers="relu3_2,relu4_2", style_layers="relu1_2,relu2_2,relu3_3"),
    "weights_alt": dict(content_weight=7.5, style_weight=33.0, tv_weight=0.02),
}
NIN_LAYERS = dict(model_file="nin", style_layers="relu1,relu3,relu5,relu7,relu9,relu11", content_layers="relu8")


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


VARIANT_FLAGS = {  # CLI spelling of FEVAL_VARIANTS (what tools/make_golden.py passed to the reference)
    "default": [],
    "no_grad_norm": ["--no_grad_norm"],
    "normalize_weights": ["--normalize_weights", "--temporal_weight", "0"],
    "avgpool": ["--pooling", "avg"],
    "no_tv_no_vsf": ["--tv_weight", "0", "--video_style_factor", "0", "--temporal_weight", "0"],
    "covariance": ["--use_covariance"],
    "layers_alt": ["--content_layers", "relu3_2,relu4_2", "--style_layers", "relu1_2,relu2_2,relu3_3"],
    "weights_alt": ["--content_weight", "7.5", "--style_weight", "33", "--tv_weight", "0.02"],
}
NIN_FLAGS = ["--style_layers", "relu1,relu3,relu5,relu7,relu9,relu11", "--content_layers", "relu8"]


@pytest.fixture(scope="session")
def weight_files(tmp_path_factory):
    """Seeded synthetic checkpoints saved under names that carry the architecture keyword."""
    import synth
    d = tmp_path_factory.mktemp("weights")
    paths = {"vgg19": str(d / "vgg19_synth.pth"), "nin": str(d / "nin_synth.pth")}
    torch.save(synth.vgg19_state_dict(), paths["vgg19"])
    torch.save(synth.nin_state_dict(), paths["nin"])
    return paths


def product_args(weight_files, extra=(), model="vgg19", optimizer="lbfgs", S=64, N=10, styles=("s.png",)):
    """Namespace built by the product's own config.get_args, the way a user would call style.py."""
    import json
    import config
    scaling = os.path.join(os.path.dirname(weight_files["vgg19"]), "scaling-test.json")
    if not os.path.exists(scaling):
        with open(scaling, "w") as f:
            json.dump({"100000": {"gpu": "0", "multidevice": False}}, f)
    argv = ["--content", "c.png", "--style", *styles, "--model_file", weight_files[model], "--disable_check",
            "--scaling_args", scaling, "--optimizer", optimizer, "--image_sizes", str(S), "--num_iters", str(N),
            "--seed", "0", "--no_hist_match"] + list(extra)
    return config.get_args(argv)


def write_video_fixture(root, n_frames=3, S=64):
    """Frames + flow cache for vid_img: same construction as tools/make_golden.py::write_video_fixture (whose files came
    from the reference's writer; the product's writer is byte-identical, tests/test_load_and_dist_cpu.py)."""
    import numpy as np
    from PIL import Image
    import load
    fdir = os.path.join(root, "clip")
    os.makedirs(fdir, exist_ok=True)
    names = []
    base = torch.rand(S + 8, S + 8, 3, generator=torch.Generator().manual_seed(21))
    for i in range(n_frames):
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
                load.write_flow(flow.numpy().astype(np.float32), os.path.join(flow_dir, f"{direction}_{a}_{b}.flo"))
                rel = ((torch.rand(S, S, generator=g) > 0.2).float() * 255).byte().numpy()
                Image.fromarray(rel, mode="L").save(os.path.join(flow_dir, f"{direction}_{a}_{b}.png"))
    return fdir, os.path.join(root, "out")


def planar_codes(codes):
    """The decision bytes of maua_pool2x2_fwd_codes / maua_conv3x3_x3w_relu_pool ([image][c / 8][pooled pixel][c % 8] in memory, held in
    a tensor of (n, c, h / 2, w / 2) bytes) as a per-channel (n, c, h / 2, w / 2) tensor."""
    n, c, ph, pw = codes.shape
    return codes.reshape(n, c // 8, ph, pw, 8).permute(0, 1, 4, 2, 3).reshape(n, c, ph, pw)
