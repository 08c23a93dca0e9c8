"""Seeded synthetic weights and images for parity tests and benchmarks.

There is no network here, so real VGG-19 / NIN checkpoints cannot be fetched
(reference: models.py:257-337 downloads them).  Everything below is a pure
function of integer seeds, so the build container, the golden generator
(tools/make_golden.py) and the GPU box regenerate identical tensors from the
same torch build instead of shipping weight files.

Image recipe mirrors the reference's own synthetic driver, max-sizes.py:51
(`torch.rand(1,3,s,s)*255`), shifted to the mean-subtracted BGR range that
load.preprocess (load.py:21-32) produces.
"""
import math

import torch

# VGG-19 feature stack, reference models.py:138 (channel_list["VGG-19"]).
VGG19_CHANNELS = [64, 64, "P", 128, 128, "P", 256, 256, 256, 256, "P",
                  512, 512, 512, 512, "P", 512, 512, 512, 512, "P"]


# The other VGG feature stacks of reference models.py:134-137: VGG-16 (also behind the "nyud" / "fcn32s" / "sod" checkpoints, models.py:259-288)
# and the channel-pruned VGG-16 ("prun", models.py:249-258) whose widths are multiples of nothing.
VGG16_CHANNELS = [64, 64, "P", 128, 128, "P", 256, 256, 256, "P", 512, 512, 512, "P", 512, 512, 512, "P"]
VGG16P_CHANNELS = [24, 22, "P", 41, 51, "P", 108, 89, 111, "P", 184, 276, 228, "P", 512, 512, 512, "P"]


def vgg19_state_dict(bias_scale=0.05, dtype=torch.float32, channels=None):
    """He-normal conv weights keyed like torchvision-style `features.<idx>.*`.

    Seeds: weight of the conv at Sequential index idx uses Generator(1000+idx),
    its bias Generator(2000+idx).  `bias_scale=0` reproduces the zero-bias
    recipe of SURVEY.md Appendix A.  `channels`: another VGG stack (VGG16_CHANNELS, VGG16P_CHANNELS), same recipe.
    """
    sd, idx, cin = {}, 0, 3
    for c in (VGG19_CHANNELS if channels is None else channels):
        if c == "P":
            idx += 1
            continue
        g = torch.Generator().manual_seed(1000 + idx)
        sd[f"features.{idx}.weight"] = (
            torch.randn(c, cin, 3, 3, generator=g) * math.sqrt(2.0 / (9 * cin))
        ).to(dtype)
        if bias_scale:
            gb = torch.Generator().manual_seed(2000 + idx)
            sd[f"features.{idx}.bias"] = (torch.randn(c, generator=gb) * bias_scale).to(dtype)
        else:
            sd[f"features.{idx}.bias"] = torch.zeros(c, dtype=dtype)
        idx += 2
        cin = c
    return sd


# NIN feature stack, reference models.py:74-113: (cout, cin, k, stride, pad) per conv,
# "P" = MaxPool2d(3, 2, 0, ceil_mode=True), "D" = Dropout.
NIN_LAYERS = [(96, 3, 11, 4, 0), (96, 96, 1, 1, 0), (96, 96, 1, 1, 0), "P",
              (256, 96, 5, 1, 2), (256, 256, 1, 1, 0), (256, 256, 1, 1, 0), "P",
              (384, 256, 3, 1, 1), (384, 384, 1, 1, 0), (384, 384, 1, 1, 0), "P", "D",
              (1024, 384, 3, 1, 1), (1024, 1024, 1, 1, 0), (1000, 1024, 1, 1, 0)]


def nin_state_dict(bias_scale=0.05, dtype=torch.float32):
    """He-normal weights for every conv of NIN.features (Sequential indices as in
    reference models.py:83-112: conv, relu pairs; pools and dropout take one slot)."""
    sd, idx = {}, 0
    for spec in NIN_LAYERS:
        if spec in ("P", "D"):
            idx += 1
            continue
        c, cin, k, _, _ = spec
        g = torch.Generator().manual_seed(3000 + idx)
        sd[f"features.{idx}.weight"] = (
            torch.randn(c, cin, k, k, generator=g) * math.sqrt(2.0 / (k * k * cin))
        ).to(dtype)
        gb = torch.Generator().manual_seed(4000 + idx)
        sd[f"features.{idx}.bias"] = (torch.randn(c, generator=gb) * bias_scale).to(dtype)
        idx += 2
    return sd


def images(S, n=3, seed=7, H=None, W=None):
    """`n` images drawn in order from one generator: rand(1,3,H,W)*255 - 120."""
    g = torch.Generator().manual_seed(seed)
    H = S if H is None else H
    W = S if W is None else W
    return [torch.rand(1, 3, H, W, generator=g) * 255 - 120 for _ in range(n)]


def frames(n_frames, S, seed=9):
    """Synthetic video: rand(n,3,S,S)*255 - 120 (BASELINE config 4)."""
    g = torch.Generator().manual_seed(seed)
    return torch.rand(n_frames, 3, S, S, generator=g) * 255 - 120


def checksum(t):
    """Order-independent fingerprint used to check that both boxes regenerate the
    same tensors: (sum, sum of squares, weighted sum) in float64."""
    d = t.detach().double().flatten()
    w = torch.arange(1, d.numel() + 1, dtype=torch.float64) / d.numel()
    return [float(d.sum()), float((d * d).sum()), float((d * w).sum())]
