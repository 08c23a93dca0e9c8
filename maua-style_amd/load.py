"""Image pre/post-processing with the reference's conventions (reference load.py:21-137), without torchvision /
skvideo (absent here): Caffe-style VGG input = BGR, 0..255, mean-subtracted.

Video decoding and the optical-flow cache (load.py:141-231) need ffmpeg binaries and the flow networks of the
un-vendored submodules; they are outside this build.  `process_content_frames` covers the flow-less video path by
reading a directory of frame images.
"""
import os

import numpy as np
import torch as th
from PIL import Image

from utils import fetch, name  # noqa: F401

Image.MAX_IMAGE_PIXELS = 1000000000  # gigapixel inputs, as in the reference

_MEAN_BGR = th.tensor([103.939, 116.779, 123.68])
_EXT = [".png", ".jpeg", ".jpg", ".tiff"]


def preprocess(image_path):
    """File (or "random") -> (1,3,H,W) float32, BGR, 0..255 minus the ImageNet BGR mean (load.py:21-32)."""
    if image_path == "random":
        arr = np.random.normal(size=(256, 256, 3)).astype(np.float32)
        arr -= arr.min()
        arr /= arr.max()
        chw = th.from_numpy(arr).permute(2, 0, 1)  # ToTensor on a float array: HWC -> CHW, no rescale
    else:
        img = Image.open(fetch(image_path)).convert("RGB")
        chw = th.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float() / 255  # ToTensor
    bgr = (chw * 255)[th.LongTensor([2, 1, 0])]
    return (bgr - _MEAN_BGR[:, None, None]).unsqueeze(0)


def deprocess(output_tensor):
    """(1,3,H,W) network-space tensor -> PIL RGB image; values are clamped to [0,1] and truncated to 8 bits like
    torchvision's ToPILImage (load.py:47-52)."""
    t = output_tensor.squeeze(0).float().cpu() + _MEAN_BGR[:, None, None]
    rgb = (t[th.LongTensor([2, 1, 0])] / 255).clamp_(0, 1)
    arr = rgb.mul(255).byte().permute(1, 2, 0).numpy()
    return Image.fromarray(arr, mode="RGB")


def original_colors(content, generated):
    """Luminance of `generated` with the chroma of `content` (load.py:236-240)."""
    content_channels = list(content.resize(generated.size).convert("YCbCr").split())
    generated_channels = list(generated.convert("YCbCr").split())
    content_channels[0] = generated_channels[0]
    return Image.merge("YCbCr", content_channels).convert("RGB")


def save_tensor_to_file(tensor, args, iteration=None, size=None, filename=None):
    """Output naming of the reference (load.py:55-74): <output>[_<size>[_<iteration>]].png"""
    if filename is None:
        if size is None:
            filename = f"{args.output}"
        elif iteration is None:
            filename = f"{args.output}_{size}"
        else:
            filename = f"{args.output}_{size}_{iteration}"
    if tensor.size()[0] > 1:
        raise NotImplementedError("writing multi-frame tensors needs skvideo/ffmpeg, which this build does not have")
    img = deprocess(tensor.clone())
    if args.original_colors == 1:
        img = original_colors(deprocess(preprocess(args.content)), img)
    img.save(f"{filename}.png")


def process_style_images(args):
    """Expand directories in --style and preprocess every image (load.py:77-94)."""
    paths = []
    for entry in args.style:
        if os.path.isdir(entry):
            paths.extend(entry + "/" + f for f in os.listdir(entry) if os.path.splitext(f)[1].lower() in _EXT)
        else:
            paths.append(entry)
    return [preprocess(p) for p in paths]


def process_content_frames(content):
    """Sorted frame image paths of a directory (the flow-less stand-in for load.process_content_video)."""
    if not os.path.isdir(content):
        raise NotImplementedError("video files need ffmpeg; pass a directory of frame images")
    frames = sorted(content.rstrip("/") + "/" + f for f in os.listdir(content) if os.path.splitext(f)[1].lower() in _EXT)
    if not frames:
        raise FileNotFoundError(f"no frame images in {content}")
    return frames
