"""Image pre/post-processing with the reference's conventions (reference load.py:21-137), without torchvision /
skvideo (absent here): Caffe-style VGG input = BGR, 0..255, mean-subtracted.

Video decoding and flow ESTIMATION (load.py:141-188) need ffmpeg binaries and the flow networks of the un-vendored
submodules; they are outside this build.  `process_content_frames` reads a directory of frame images, and the flow
files the reference caches (`.flo` fields, reliability PNGs; load.py:191-231) are consumed by `flow_warp_map` /
`reliable_flow_weighting`.
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


def preprocess_u8(rgb_hwc):
    """`preprocess` of an image that is already in memory as (H,W,3) uint8 RGB (CPU or device tensor): the same fp32
    operations as reading it back from a (lossless) PNG, so the result is bit-identical to preprocess(path)."""
    chw = rgb_hwc.permute(2, 0, 1).float() / 255
    bgr = (chw * 255)[th.tensor([2, 1, 0], device=rgb_hwc.device)]
    return (bgr - _MEAN_BGR.to(rgb_hwc.device)[:, None, None]).unsqueeze(0)


def deprocess(output_tensor):
    """(1,3,H,W) network-space tensor -> PIL RGB image; values are clamped to [0,1] and truncated to 8 bits like
    torchvision's ToPILImage (load.py:47-52)."""
    if output_tensor.is_cuda:  # same arithmetic in one kernel: 3 bytes per pixel cross PCIe instead of 12
        import hip
        return Image.fromarray(hip.deprocess_u8(output_tensor.float().contiguous(), _MEAN_BGR).cpu().numpy(), mode="RGB")
    t = output_tensor.squeeze(0).float() + _MEAN_BGR[:, None, None]
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
        # The reference hands the clip to skvideo/ffmpeg as "<filename>.mp4" (load.py:65-69); codecs are out of scope here,
        # so the same frames (mean added back, BGR -> RGB, clamped, truncated to 8 bits) go to a directory of PNGs named
        # like the video would have been, which preprocess_video reads back.
        os.makedirs(filename, exist_ok=True)
        clip = tensor.detach().float().cpu() + _MEAN_BGR[None, :, None, None]
        clip = clip[:, th.LongTensor([2, 1, 0])].permute(0, 2, 3, 1).clamp_(0, 255).byte().numpy()
        for t, frame in enumerate(clip):
            Image.fromarray(frame, mode="RGB").save(f"{filename}/frame_{t:05d}.png")
        return
    save_image_to_file(deprocess(tensor), args, filename)


def save_image_to_file(img, args, filename):
    """The host half of save_tensor_to_file for one deprocessed image: optional colour transfer from the content image
    (reference load.py:71-73) and PNG encoding - no GPU call, so the workflow drivers run it on a background thread."""
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


def preprocess_video(video_path, fps=None):
    """A clip as (T,3,H,W) network-space tensor (reference load.py:35-43).  The reference decodes with skvideo at `fps`
    and falls back to a single image; here a clip is a directory of frame images (sorted by name), and a single image
    is a one-frame clip."""
    if os.path.isdir(video_path):
        return th.cat([preprocess(p) for p in process_content_frames(video_path)], dim=0)
    if os.path.splitext(video_path)[1].lower() in _EXT:
        return preprocess(video_path)
    raise NotImplementedError(f"{video_path}: decoding video files needs skvideo/ffmpeg; pass a directory of frame images")


def process_style_videos(args):
    """Every --style entry is one style clip (reference load.py:103-136: a list of videos, blend weights normalised to
    sum to one - config.get_args has already done that here)."""
    return [preprocess_video(entry, getattr(args, "fps", None)) for entry in args.style]


def process_content_frames(content):
    """Sorted frame image paths of a directory (the flow-less stand-in for load.process_content_video)."""
    if not os.path.isdir(content):
        raise NotImplementedError("video files need ffmpeg; pass a directory of frame images")
    frames = sorted(content.rstrip("/") + "/" + f for f in os.listdir(content) if os.path.splitext(f)[1].lower() in _EXT)
    if not frames:
        raise FileNotFoundError(f"no frame images in {content}")
    return frames


# ---------------------------------------------------------------------------------------------------------
# Precomputed optical flow.  Estimating flow (the reference's un-vendored flow networks) is out of
# scope; consuming the files it leaves in <output_dir>/flow/ is not: Middlebury .flo fields and reliability PNGs.
# ---------------------------------------------------------------------------------------------------------
FLO_MAGIC = 202021.25


def read_flo(filename):
    """Middlebury .flo: float32 magic 202021.25, int32 width, int32 height, then h*w (u, v) float32 pairs."""
    with open(filename, "rb") as f:
        magic = np.fromfile(f, np.float32, count=1)
        if magic.size != 1 or float(magic[0]) != FLO_MAGIC:
            raise ValueError(f"{filename}: not a .flo file (magic {magic})")  # the reference prints and then fails on `w`
        w = int(np.fromfile(f, np.int32, count=1)[0])
        h = int(np.fromfile(f, np.int32, count=1)[0])
        data = np.fromfile(f, np.float32, count=2 * w * h)
    if data.size != 2 * w * h:
        raise ValueError(f"{filename}: truncated flow field ({data.size} of {2 * w * h} values)")
    return data.reshape(h, w, 2)


def write_flow(flow, filename):
    """Inverse of read_flo (reference load.py:221-231)."""
    h, w = flow.shape[:2]
    with open(filename, "wb") as f:
        np.array([FLO_MAGIC], dtype=np.float32).tofile(f)
        np.array([w], dtype=np.int32).tofile(f)
        np.array([h], dtype=np.int32).tofile(f)
        np.asarray(flow, dtype=np.float32).tofile(f)


def flow_warp_map(filename, current_size):
    """Sampling grid for F.grid_sample that moves an image along a stored flow field (reference load.py:191-214):
    flow in pixels -> fractions of the image size, Gaussian-smoothed (sigma 5 px), added to the identity grid on
    [-1, 1]^2, bilinearly resized to `current_size`.  Returns (1, H, W, 2)."""
    import scipy.ndimage
    import torch.nn.functional as F
    flow = read_flo(filename).copy()
    h, w = flow.shape[:2]
    flow[:, :, 0] /= w
    flow[:, :, 1] /= h
    flow = scipy.ndimage.gaussian_filter(flow, [5, 5, 0])
    gx, gy = np.meshgrid(np.linspace(-1, 1, w), np.linspace(-1, 1, h))
    grid = th.from_numpy((np.stack([gx, gy], axis=2) + flow).astype(np.float32)).unsqueeze(0)
    return F.interpolate(grid.permute(0, 3, 1, 2), size=tuple(current_size), mode="bilinear",
                         align_corners=False).permute(0, 2, 3, 1)


def reliable_flow_weighting(filename):
    """Reliability mask PNG -> (1, C, H, W) float in [0, 1] (reference load.py:217-218, torchvision ToTensor)."""
    arr = np.asarray(Image.open(filename))
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = th.from_numpy(np.ascontiguousarray(arr)).permute(2, 0, 1)
    return (t.float() / 255.0 if t.dtype == th.uint8 else t.float()).unsqueeze(0)
