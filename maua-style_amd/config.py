"""Command-line / JSON configuration with the reference's exact flag surface (reference config.py).

Precedence, as in the reference: defaults < --load_args file < explicit CLI values, later overridden per
image size by the --scaling_args table inside optim.optimize (optim.set_model_args).  Derived fields
(`normalize_gradients`, `match_histograms`, `cudnn_autotune`, integer lists, normalised blend weights,
`dtype`, `multidevice`, `backward_device`, `output`, `ffmpeg`) are produced by `postprocess`.

This build computes on MI355X only: `--gpu c` still parses (so saved argument files load), but the
loaders in models.py refuse to build a CPU network - there is no CPU compute path.
"""
import argparse
import json
import os
import uuid

import torch

from utils import name

HERE = os.path.dirname(os.path.abspath(__file__))

DEFAULT_FFMPEG_ARGS = "config/ffmpeg-libx264.json"

# (flag, kwargs) in the reference's order (config.py:15-89); types and defaults are part of the interface
_FLAGS = [
    # input options
    ("transfer_type", dict(default="img_img", choices=["img_img", "vid_img", "img_vid"])),
    ("output_dir", dict(default="./output")),
    ("content", dict(help="Content target image")),
    ("style", dict(help="Style target image", nargs="*")),
    ("init", dict(type=str, default="random")),
    ("seed", dict(type=int, default=-1)),
    # main parameters
    ("image_sizes", dict(default="256,512,724,1024,1448")),
    ("num_iters", dict(default="500,400,300,200,100")),
    ("content_weight", dict(type=float, default=5)),
    ("temporal_weight", dict(type=float, default=50)),
    ("style_weight", dict(type=float, default=100)),
    ("style_blend_weights", dict(default=None)),
    ("style_scale", dict(type=float, default=1.0)),
    ("tv_weight", dict(type=float, default=1e-3)),
    # model settings
    ("model_file", dict(type=str, default="vgg19",
                        help="Path to model .pth file or one of [prune, nyud, fcn32s, sod, vgg19, vgg16, nin]")),
    ("content_layers", dict(help="layers for content", default="relu4_2")),
    ("style_layers", dict(help="layers for style", default="relu1_1,relu2_1,relu3_1,relu4_1,relu5_1")),
    ("pooling", dict(choices=["avg", "max"], default="max")),
    ("disable_check", dict(action="store_true")),
    # switches
    ("original_colors", dict(action="store_true")),
    ("normalize_weights", dict(action="store_true")),
    ("no_grad_norm", dict(action="store_true")),
    ("no_hist_match", dict(action="store_true")),
    ("use_covariance", dict(action="store_true")),
    # optimizer
    ("optimizer", dict(choices=["lbfgs", "adam"], default="lbfgs")),
    ("learning_rate", dict(type=float, default=1)),
    ("lbfgs_num_correction", dict(type=int, default=100)),
    ("lbfgs_tolerance_change", dict(type=int, default=-1)),
    ("lbfgs_tolerance_grad", dict(type=int, default=-1)),
    # gpu
    ("gpu", dict(type=str, default="0", help="Zero-indexed ID of the GPU to use; for CPU mode set -gpu = c")),
    ("backend", dict(choices=["nn", "cudnn", "mkl", "mkldnn", "openmp", " mkl,cudnn ", " cudnn,mkl "], default="cudnn")),
    ("multidevice_strategy", dict(default="5")),
    ("no_cudnn_autotune", dict(action="store_true")),
    # video content settings
    ("flow_models", dict(type=str, default="unflow,pwc,spynet,liteflownet")),
    ("no_check_occlusion", dict(action="store_true")),
    ("passes_per_scale", dict(type=int, default=4)),
    ("loop", dict(action="store_true")),
    ("temporal_blend", dict(type=float, default=0.5)),
    ("fps", dict(type=float, default=24)),
    # video style settings
    ("num_frames", dict(type=int, default=48)),
    ("video_style_factor", dict(type=float, default=100)),
    ("gram_frame_window", dict(type=str, default="18,9,7")),
    ("avg_frame_window", dict(type=int, default=18)),
    ("shift_factor", dict(type=float, default=0)),
    # clip settings (parsed for compatibility; the CLIP/VQGAN path is outside this build)
    ("content_text", dict(type=str, default=None)),
    ("style_text", dict(type=str, default=None)),
    ("text_weight", dict(type=float, default=1)),
    ("vqgan_dir", dict(type=str, default="imagenet_16384")),
    ("clip_backbone", dict(type=str, default="ViT-B/32", choices=["RN50", "RN101", "RN50x4", "ViT-B/32"])),
    # logging
    ("verbose", dict(action="store_true")),
    ("print_iter", dict(type=int, default=0)),
    ("save_iter", dict(type=int, default=0)),
    ("save_args", dict(action="store_true")),
    ("load_args", dict(type=str, default=None)),
    ("ffmpeg_args", dict(type=str, default=DEFAULT_FFMPEG_ARGS)),
    ("scaling_args", dict(type=str, default="config/scaling-img.json",
                          help="multi-network multi-scale model-parallel configuration")),
    ("uniq", dict(action="store_true")),
]


def build_parser():
    parser = argparse.ArgumentParser()
    for flag, kw in _FLAGS:
        parser.add_argument("--" + flag, **kw)
    return parser


def _resolve(path):
    """Relative preset paths (config/...) resolve against the cwd first, then this package."""
    if os.path.exists(path) or os.path.isabs(path):
        return path
    alt = os.path.join(HERE, path)
    return alt if os.path.exists(alt) else path


def _output_stem(args):
    stem = f"{name(args.content)}_{'_'.join([name(s) for s in args.style])}"
    if args.uniq:
        stem += f"_{str(uuid.uuid4())[:6]}"
    return stem


def get_args(argv=None):
    """Parse the command line (reference config.py:10-131)."""
    parser = build_parser()
    args = parser.parse_args(argv)
    stem = _output_stem(args)

    if args.load_args is not None:
        # file values, overridden by every CLI value that differs from its default, and completed with the CLI value
        # of every key the file lacks (config.py:98-116)
        with open(_resolve(args.load_args), "r") as f:
            merged = json.load(f)
        for key, value in vars(args).items():
            if value != parser.get_default(key) or key not in merged:
                merged[key] = value
        args = argparse.Namespace(**merged)

    if args.save_args:
        with open(f"config/{stem}_args.json", "w") as f:
            json.dump(args.__dict__, f, indent=2)

    args.output = f"{args.output_dir}/{stem}"

    # Encoder settings are carried on `args.ffmpeg` as in the reference (config.py:129-132) although this build never
    # encodes video.  The stock preset is built in, so the default path needs no file; any other path is read.
    path = _resolve(args.ffmpeg_args)
    if os.path.exists(path):
        with open(path, "r") as f:
            ffargs = json.load(f)
    elif args.ffmpeg_args == DEFAULT_FFMPEG_ARGS:
        ffargs = {"c:v": "libx264", "preset": "slow", "pix_fmt": "yuv420p"}
    else:
        raise FileNotFoundError(path)
    ffargs["framerate"] = args.fps
    args.ffmpeg = ffargs
    return postprocess(args)


def postprocess(args):
    """Derived fields (reference config.py:134-168)."""
    args.normalize_gradients = not args.no_grad_norm
    args.match_histograms = not args.no_hist_match
    args.cudnn_autotune = not args.no_cudnn_autotune

    args.image_sizes = [int(s) for s in ("" + args.image_sizes).split(",")]
    args.num_iters = [int(s) for s in ("" + args.num_iters).split(",")]
    assert len(args.image_sizes) == len(
        args.num_iters), "-image_sizes and -num_iters must have the same number of elements!"

    if args.style_blend_weights is None:
        blend = [1.0] * (len(args.style) if args.style is not None else 1)
    else:
        blend = [float(x) for x in args.style_blend_weights.split(",")]
        assert len(blend) == len(
            args.style), "-style_blend_weights and -style_images must have the same number of elements!"
    total = sum(blend)
    args.style_blend_weights = [b / total for b in blend]  # normalised to sum to 1

    args.dtype, args.multidevice, args.backward_device = setup_gpu(args)
    return args


def setup_gpu(args):
    """(tensor type, multidevice?, device that holds the summed loss) from --gpu / --backend
    (reference config.py:171-207).  The cudnn/mkl switches are torch-backend knobs of the reference's
    numerics backend; they are accepted and ignored here because every hot op runs in libmaua_hip."""
    spec = str(args.gpu)
    first = spec.split(",")[0]
    cpu_first = "c" in (first.lower() if "," in spec else spec.lower())
    if cpu_first and "mkldnn" in args.backend:
        raise ValueError("MKL-DNN is not supported yet.")
    if "," in spec:
        if cpu_first:
            return torch.FloatTensor, True, "cpu"
        return torch.cuda.FloatTensor, True, "cuda:" + first
    if not cpu_first:
        return torch.cuda.FloatTensor, False, "cuda:" + spec
    return torch.FloatTensor, False, "cpu"


def load_args(filepath):
    """Namespace from a saved JSON argument file (reference config.py:210-224)."""
    with open(_resolve(filepath), "r") as f:
        args = argparse.Namespace(**json.load(f))
    if args.content is not None and args.style is not None:
        args.output = f"{args.output_dir}/{_output_stem(args)}"
    return postprocess(args)
