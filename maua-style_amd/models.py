"""Feature networks and loss-network assembly with the reference's names (reference models.py).

`load_model(args) -> (net, losses)` builds the same module sequence as the reference
(models.py:351-453): optional TVLoss and temporal ContentLoss on the pixels, then the conv / ReLU / pool
stack of VGG or NIN with ContentLoss / StyleLoss modules inserted after the named layers, truncated after
the last requested relu layer, parameters frozen, loss lists monkey-patched onto the net.

Differences that follow from being MI355X-native:
  * Conv2d / ReLU / MaxPool2d / AvgPool2d here are thin nn.Modules over libmaua_hip kernels (hip.py) with HIP
    backward passes; they only accept ROCm tensors (there is no CPU compute path);
  * only the `features` stack is materialised - the reference instantiates VGG's 25088x4096 classifier just
    to drop it (models.py:16-28, :357), which dominates its start-up time;
  * nothing is downloaded (no network): `--model_file` must name an existing .pth whose path contains the
    architecture keyword, exactly the rule the reference uses to pick the architecture (models.py:246-341);
  * the layer-split `ModelParallel` wrappers (models.py:456-566) exist to fit 11 GB cards and are not
    provided; `--gpu a,b` raises.
"""
import copy
from os import path

import torch
import torch.nn as nn

import hip
import plan
from loss import ContentLoss, GramMatrix, ScaleGradients, StyleLoss, TVLoss  # noqa: F401  (re-exported like `from loss import *`)

# --------------------------------------------------------------------------------------------------
# HIP-backed layers
# --------------------------------------------------------------------------------------------------

ROUTE_LOG = None  # a list while engine.StyleEngine.describe_routes records which kernel every convolution launch takes


def _route(kernel, x, produced, pad, backward, tile, ksplit=1, **fused):
    """One record per convolution launch (only while ROUTE_LOG is a list): kernel family, geometry, tile, K splits, what rides along."""
    if ROUTE_LOG is not None:
        rec = {"kernel": kernel, "pass": "bwd" if backward else "fwd", "consumed": int(x.shape[1]), "produced": int(produced),
               "plane": [int(x.shape[2]), int(x.shape[3])], "n": int(x.shape[0]), "pad": int(pad), "tile": tile, "ksplit": int(ksplit)}
        rec.update({k: v for k, v in fused.items() if v})
        ROUTE_LOG.append(rec)



def _x6_mode():
    """planner field conv_x6: "1" (default) = 3x3 stride-1 convs on the bf16 matrix cores with a 3-way operand split in both
    passes, "fwd" / "bwd" = only that pass, "0" = fp32 matrix cores everywhere."""
    mode = plan.get("conv_x6")
    return mode in ("1", "fwd"), mode in ("1", "bwd")


def _x3_enabled():
    """planner field conv_x3: "1" (default) = the fp32-accurate 3x3 convolution runs as fp16x3 (two-part fp16 split, three MFMAs per
    product block, conv_x3.hip; the 3-channel image layer keeps the exact bf16x6 products); "0" = bf16x6 everywhere (and NIN's 1x1 / 5x5 layers on the fp32 matrix cores)
    (three-part bf16 split, six MFMAs, conv_x6.hip).  Measured pixel-gradient error against the fp64 reference:
    2.8e-7 / 2.6e-7; the reference's own fp32 arithmetic: 4.5e-7 (4.6e-7 if the image layer ran on fp16x3 too)."""
    return plan.get("conv_x3") == "1"


def _x3w_enabled():
    """planner field conv_x3w: "1" (default) = layers whose consumed channel count is a multiple of 16 run the wide-tile fp16x3 kernel
    (conv_x3w.hip: 16-channel chunks, four accumulators per wave, two workgroups per CU); "0" = conv_x3.hip everywhere."""
    return plan.get("conv_x3w") == "1"


def _image_kernel_enabled():
    """planner field conv_image: "1" (default) = a 3x3 layer that consumes at most three channels (conv1_1) runs conv_img.hip in the forward pass;
    "0" = conv_x6.hip's general kernel with the image as one 8-channel chunk (the same bf16x6 products, other summation order)."""
    return plan.get("conv_image") == "1"


def _x3q_min_channels():
    """planner field conv_x3q: the smallest consumed channel count (a multiple of 32) from which a 3x3 layer runs conv_x3q.hip - 32-channel chunks on
    v_mfma_f32_16x16x32_f16, one workgroup of eight waves per CU (round 4) - instead of conv_x3w.hip; "0" = never.  Default 256: measured
    on one box (tools/bench_x3q.py, 1024 x 1024 layer shapes) 1.13-1.22 x conv_x3w at 512 channels, 1.04-1.13 x at 256, 1.02 x at 128;
    the 64- and 128-channel layers run two to four chunks per workgroup, too few to pay for the single workgroup's prologue."""
    v = plan.get("conv_x3q")
    return int(v) if v.isdigit() else 256


def _x3p_min_channels():
    """planner field conv_x3p: the smallest consumed channel count (a multiple of 32) from which a 3x3 layer runs conv_x3p.hip - conv_x3q's
    workgroup made persistent (round 5): one stream of chunks per CU, the next tile staged under the current one, the epilogue under the
    next tile's first chunk - instead of conv_x3q.hip / conv_x3w.hip; "0" = never.  Default 64: every VGG layer it supports."""
    v = plan.get("conv_x3p")
    return int(v) if v.isdigit() else 64


def conv3x3_is_x3p(consumed, h, w, pad, produced, n=1, accumulate=False):
    """Whether a 3x3 stride-1 pass that consumes `consumed` channels of an h x w plane and produces `produced` runs on conv_x3p.hip
    (forward: pad = the layer's padding; backward-data: 2 - padding)."""
    mc = _x3p_min_channels()
    return _x3_enabled() and _x3w_enabled() and mc > 0 and mc <= consumed <= plan.get_int("conv_x3p_max") and not accumulate and h * w >= _x3w_min_pixels() and \
        hip.conv_x3p_supported(consumed, h, w, produced, pad) and hip.conv_x3p_preferred(n, consumed, h, w, produced, pad)


def _x3p_gram_enabled(n, c_in, c_out, h, w):
    """Whether the Gram backward rides in conv_x3p's launch (its D . F chunks run between two items, latency-exposed) or in conv_x3w's
    (MAUA_X3P_GRAM_MIN_MB, default 700): measured in the network, conv_x3p's form wins 7-8 % where the launch's maps - gradient in,
    gradient out, F - are beyond what the 256 MB of Infinity Cache hold (2048 x 2048 images) and loses 2-3 % where they are not (1024)."""
    # (the frames the job PLANS per launch count, not this launch's batch: a frame's bits must not depend on how many others share its
    #  launches - a short last batch, or a frame optimised alone under the same plan, takes the same kernel)
    mb = hip.lib().maua_get_split_batch_hint() * (c_in + 2 * c_out) * h * w * 4 / 1e6
    return mb >= plan.get_float("x3p_gram_min_mb")


X3Q_UNPOOL_MIN_CHANNELS = 512  # conv_x3q's unpooling form pays from here: its corner vectors are cut in the exposed store phase
                               # (measured in the network: conv4_4's backward pass 181 -> 176 us, conv3_4's 186 -> 207)


def conv3x3_is_x3q(consumed, h, w, pad, at_least=0, produced=None, n=1):
    """Whether a 3x3 stride-1 pass that consumes `consumed` channels of an h x w plane runs on conv_x3q.hip (forward: consumed = the
    layer's input channels, pad = its padding; backward-data: its output channels, 2 - padding).  `at_least`: a higher channel bound
    for this pass (the backward pass of a conv + ReLU + pool group: the same kernel family whether or not the pool's backward pass is
    fused into it, so that the fusion changes no bit)."""
    mc = max(_x3q_min_channels(), at_least) if _x3q_min_channels() > 0 else 0
    return _x3_enabled() and _x3w_enabled() and mc > 0 and consumed >= mc and h * w >= _x3w_min_pixels() and \
        hip.conv_x3q_supported(consumed, h, w, pad) and (produced is None or hip.conv_x3q_preferred(n, consumed, h, w, produced, pad))


def _x3w_min_pixels():
    """Planes smaller than this run conv_x3.hip (4-row tiles, 1024 workgroup slots): on maps of a few tiles the wide kernel's
    8-row tiles and 512 slots leave the chip emptier (measured at 256 x 256: 726 vs 793 it/s).  Planner field x3w_min_pixels."""
    return plan.get_int("x3w_min_pixels")


def conv3x3_mfma(x, mod, backward, out=None, out_relu_mask=None, relu=False, accumulate=False, workspace=None, pool_group=False):
    """The fp32-accurate reduced-width matrix-core convolution of a 3x3 stride-1 layer (forward, or backward-data when
    `backward`): fp16x3 or bf16x6 according to MAUA_CONV_X3.  `pool_group`: the backward pass of a layer whose output feeds a 2x2 max
    pool with kept decisions, run here on the pool's own backward output (MAUA_FUSE_UNPOOL=0): same kernel family as the fused form."""
    pad = mod.padding[0]
    if backward:
        cout, p, bias = mod.in_channels, 2 - pad, None
    else:
        cout, p, bias = mod.out_channels, pad, mod.bias_device()
    consumed = mod.out_channels if backward else mod.in_channels
    n, h, w = x.shape[0], x.shape[2], x.shape[3]
    if conv3x3_is_x3p(consumed, x.shape[2], x.shape[3], p, cout, x.shape[0], accumulate):
        bf, bb, wsc = mod.banks3q()
        _route("conv_x3p", x, cout, p, backward, "64co x 16x32px, persistent", hip.conv_x3p_split(n, consumed, h, w, cout, p) if workspace is not None else 1,
               mask=out_relu_mask is not None, relu=relu)
        return hip.conv3x3_x3p(x, bb if backward else bf, wsc, bias, cout, p, relu, out=out, out_relu_mask=out_relu_mask, workspace=workspace)
    if conv3x3_is_x3q(consumed, x.shape[2], x.shape[3], p, X3Q_UNPOOL_MIN_CHANNELS if pool_group else 0, cout, x.shape[0]):
        bf, bb, wsc = mod.banks3q()
        _route("conv_x3q", x, cout, p, backward, "64co x 16x32px", hip.conv_x3q_split(n, consumed, h, w, cout, p) if workspace is not None else 1,
               mask=out_relu_mask is not None, relu=relu, accumulate=accumulate)
        return hip.conv3x3_x3q(x, bb if backward else bf, wsc, bias, cout, p, relu, out=out, out_relu_mask=out_relu_mask,
                               accumulate=accumulate, workspace=workspace)
    if _x3_enabled() and _x3w_enabled() and x.shape[2] * x.shape[3] >= _x3w_min_pixels() and \
            hip.conv_x3w_supported(consumed, x.shape[2], x.shape[3], p):
        bf, bb, wsc = mod.banks3w()
        _route("conv_x3w", x, cout, p, backward, "64co x 8x32px", hip.conv_x3w_split(n, consumed, h, w, cout, p) if workspace is not None else 1,
               mask=out_relu_mask is not None, relu=relu, accumulate=accumulate)
        return hip.conv3x3_x3w(x, bb if backward else bf, wsc, bias, cout, p, relu, out=out, out_relu_mask=out_relu_mask,
                               accumulate=accumulate, workspace=workspace)
    if _x3_enabled() and consumed > 4:
        # (the image layer, 3 input channels, stays on the exact bf16x6 products: it differences neighbouring pixels of large
        # common magnitude - the one place where the 2 bits fp16x3 drops could show - and costs one partly empty chunk)
        bf, bb, wsc = mod.banks3()
        _route("conv_x3", x, cout, p, backward, "64co x 4x32px", mask=out_relu_mask is not None, relu=relu, accumulate=accumulate)
        return hip.conv3x3_x3(x, bb if backward else bf, wsc, bias, cout, p, relu, out=out, out_relu_mask=out_relu_mask,
                              accumulate=accumulate, workspace=workspace)
    if not backward and consumed <= 3 and out_relu_mask is None and not accumulate and _image_kernel_enabled() and \
            hip.conv_image_supported(n, consumed, h, w, cout, p):  # (planes from 2^24 pixels on - 4096 x 4096 - take the general kernel below)
        # the image layer: the 27 (channel, tap) pairs as K, no LDS - bound by writing the activation (conv_img.hip; same bf16x6 products)
        _route("conv_image", x, cout, p, backward, "27 (channel, tap) pairs as K, bf16x6", relu=relu)
        return hip.conv3x3_image(x, mod.bank_image(), cout, p, relu, out=out)
    bf, bb = mod.banks6()
    _route("conv_x6" if cout > 4 or consumed < 16 else "conv3x3_few_out", x, cout, p, backward, "bf16x6", mask=out_relu_mask is not None, relu=relu, accumulate=accumulate)
    return hip.conv3x3_x6(x, bb if backward else bf, bias, cout, p, relu, out=out, out_relu_mask=out_relu_mask,
                          accumulate=accumulate, workspace=workspace)


def conv3x3_fwd_is_x3w(mod, h, w):
    """Whether the forward pass of a 3x3 layer on an h x w input plane runs on conv_x3w.hip."""
    return _x3_enabled() and _x3w_enabled() and h * w >= _x3w_min_pixels() and mod.kernel_size[0] == 3 and mod.stride[0] == 1 and \
        hip.conv_x3w_supported(mod.in_channels, h, w, mod.padding[0])


def conv3x3_fwd_family(mod, n, h, w):
    """(kernel family, K splits) the forward pass of a 3x3 layer on an n x h x w input takes with a workspace - what conv3x3_mfma /
    conv3x3_relu_pool will launch, so that the engine's eligibility rules ask about THAT kernel."""
    cin, cout, pad = mod.in_channels, mod.out_channels, mod.padding[0]
    if conv3x3_is_x3p(cin, h, w, pad, cout, n):
        return "conv_x3p", hip.conv_x3p_split(n, cin, h, w, cout, pad)
    if conv3x3_is_x3q(cin, h, w, pad, 0, cout, n):
        return "conv_x3q", hip.conv_x3q_split(n, cin, h, w, cout, pad)
    return "conv_x3w", hip.conv_x3w_split(n, cin, h, w, cout, pad)


def conv3x3_relu_pool(x, mod, pooled, codes, workspace=None):
    """conv + bias + ReLU + the 2x2 / 2 max pool behind it without the full-size activation (hip.conv3x3_x3w_relu_pool: one launch, or -
    small grids, with a workspace - a split channel loop whose adding pass pools)."""
    if conv3x3_is_x3p(mod.in_channels, x.shape[2], x.shape[3], mod.padding[0], mod.out_channels, x.shape[0]):
        bf, _, wsc = mod.banks3q()
        _route("conv_x3p", x, mod.out_channels, mod.padding[0], False, "64co x 16x32px, persistent",
               hip.conv_x3p_split(x.shape[0], mod.in_channels, x.shape[2], x.shape[3], mod.out_channels, mod.padding[0]) if workspace is not None else 1, relu=True, pool=True)
        return hip.conv3x3_x3p(x, bf, wsc, mod.bias_device(), mod.out_channels, mod.padding[0], True, out=pooled, pool_codes=codes,
                               workspace=workspace)
    if conv3x3_is_x3q(mod.in_channels, x.shape[2], x.shape[3], mod.padding[0], 0, mod.out_channels, x.shape[0]):
        bf, _, wsc = mod.banks3q()
        _route("conv_x3q", x, mod.out_channels, mod.padding[0], False, "64co x 16x32px",
               hip.conv_x3q_split(x.shape[0], mod.in_channels, x.shape[2], x.shape[3], mod.out_channels, mod.padding[0]) if workspace is not None else 1, relu=True, pool=True)
        return hip.conv3x3_x3q_relu_pool(x, bf, wsc, mod.bias_device(), mod.out_channels, mod.padding[0], pooled, codes, workspace=workspace)
    bf, _, wsc = mod.banks3w()
    _route("conv_x3w", x, mod.out_channels, mod.padding[0], False, "64co x 8x32px",
           hip.conv_x3w_split(x.shape[0], mod.in_channels, x.shape[2], x.shape[3], mod.out_channels, mod.padding[0]) if workspace is not None else 1, relu=True, pool=True)
    return hip.conv3x3_x3w_relu_pool(x, bf, wsc, mod.bias_device(), mod.out_channels, mod.padding[0], pooled, codes, workspace=workspace)


def conv3x3_bwd_is_x3w(mod, h, w):
    """Whether the backward-data pass of a 3x3 layer on an h x w plane runs on conv_x3w.hip (the kernel that can take the Gram
    backward of a style loss along, hip.conv3x3_x3w_gram)."""
    pad = mod.padding[0]
    return _x3_enabled() and _x3w_enabled() and h * w >= _x3w_min_pixels() and mod.kernel_size[0] == 3 and mod.stride[0] == 1 and \
        hip.conv_x3w_supported(mod.out_channels, h, w, 2 - pad)


def conv3x3_bwd_with_gram(gy, mod, feature_map, dmat_bank, dmat_inv_scale, out, workspace=None):
    """out = [F > 0] * (backward-data of the layer + D . F): the layer's input gradient and the Gram backward of the style loss on
    its input activation F in one launch."""
    if conv3x3_is_x3p(mod.out_channels, gy.shape[2], gy.shape[3], 2 - mod.padding[0], mod.in_channels, gy.shape[0]) and \
            _x3p_gram_enabled(gy.shape[0], mod.out_channels, mod.in_channels, gy.shape[2], gy.shape[3]):
        _, bb, wsc = mod.banks3q()
        _route("conv_x3p", gy, mod.in_channels, 2 - mod.padding[0], True, "64co x 16x32px, persistent", mask=True, gram=True)
        return hip.conv3x3_x3p(gy, bb, wsc, None, mod.in_channels, 2 - mod.padding[0], False, out=out, out_relu_mask=feature_map,
                               dmat_bank=dmat_bank, dmat_inv_scale=dmat_inv_scale, workspace=workspace)
    _, bb, wsc = mod.banks3w()
    _route("conv_x3w", gy, mod.in_channels, 2 - mod.padding[0], True, "64co x 8x32px",
           hip.conv_x3w_split(gy.shape[0], mod.out_channels, gy.shape[2], gy.shape[3], mod.in_channels, 2 - mod.padding[0]) if workspace is not None else 1, mask=True, gram=True)
    return hip.conv3x3_x3w_gram(gy, bb, wsc, feature_map, dmat_bank, dmat_inv_scale, mod.in_channels, 2 - mod.padding[0], out=out,
                                workspace=workspace)


def conv3x3_bwd_from_pooled(gy_pooled, codes, honour_relu_bit, mod, out, out_relu_mask=None, dmat_bank=None, dmat_inv_scale=None,
                            workspace=None):
    """Backward-data pass of a conv + ReLU + 2x2 max pool group from the gradient of the POOLED map and the pool's decision bytes
    (hip.conv3x3_x3w_unpool): the pool's backward pass happens while the kernel stages its input; with `dmat_bank`, the Gram backward
    of the style loss on the layer's input goes along (out_relu_mask = that activation)."""
    if conv3x3_is_x3p(mod.out_channels, out.shape[2], out.shape[3], 2 - mod.padding[0], mod.in_channels, gy_pooled.shape[0]) and \
            (dmat_bank is None or _x3p_gram_enabled(out.shape[0], mod.out_channels // 4, mod.in_channels, out.shape[2], out.shape[3])):
        _, bb, wsc = mod.banks3q()
        _route("conv_x3p", out, mod.in_channels, 2 - mod.padding[0], True, "64co x 16x32px, persistent", mask=out_relu_mask is not None, unpool=True, gram=dmat_bank is not None)
        return hip.conv3x3_x3p(gy_pooled, bb, wsc, None, mod.in_channels, 2 - mod.padding[0], False, out=out, out_relu_mask=out_relu_mask,
                               in_codes=codes, honour_relu_bit=honour_relu_bit, dmat_bank=dmat_bank, dmat_inv_scale=dmat_inv_scale,
                               workspace=workspace)
    if dmat_bank is None and conv3x3_is_x3q(mod.out_channels, out.shape[2], out.shape[3], 2 - mod.padding[0], X3Q_UNPOOL_MIN_CHANNELS,
                                            mod.in_channels, gy_pooled.shape[0]):
        _, bb, wsc = mod.banks3q()
        _route("conv_x3q", out, mod.in_channels, 2 - mod.padding[0], True, "64co x 16x32px", mask=out_relu_mask is not None, unpool=True)
        return hip.conv3x3_x3q_unpool(gy_pooled, codes, honour_relu_bit, bb, wsc, mod.in_channels, 2 - mod.padding[0], out=out,
                                      out_relu_mask=out_relu_mask, workspace=workspace)
    _, bb, wsc = mod.banks3w()
    _route("conv_x3w", out, mod.in_channels, 2 - mod.padding[0], True, "64co x 8x32px", mask=out_relu_mask is not None, unpool=True, gram=dmat_bank is not None)
    return hip.conv3x3_x3w_unpool(gy_pooled, codes, honour_relu_bit, bb, wsc, mod.in_channels, 2 - mod.padding[0], out=out,
                                  out_relu_mask=out_relu_mask, dmat_bank=dmat_bank, dmat_inv_scale=dmat_inv_scale, workspace=workspace)


def conv_strided_bwd_is_3x3(mod, h_out, w_out):
    """Whether the backward-data pass of a strided layer runs as a stride-1 3x3 convolution over the output sites + depth to space (planner
    field strided_bwd_3x3): kernel k <= 3 stride, no padding, stride^2 x in_channels <= 64 produced channels, out_channels a multiple of 16
    (conv_x3w's chunks) - NIN's conv1, `nn.Conv2d(3, 96, (11, 11), (4, 4))` (reference models.py:84): 4.5 GFLOP that the direct kernel takes
    140 us for at 1024 x 1024."""
    k, stride, pad = mod.kernel_size[0], mod.stride[0], mod.padding[0]
    return plan.on("strided_bwd_3x3") and _x6_mode()[1] and _x3_enabled() and _x3w_enabled() and stride > 1 and pad == 0 and k <= 3 * stride and \
        stride * stride * mod.in_channels <= 64 and mod.out_channels % 16 == 0 and (h_out + 2) * (w_out + 2) >= _x3w_min_pixels() and \
        hip.conv_x3w_supported(mod.out_channels, h_out, w_out, 2)


def conv_few_is_mfma(mod, n, h, w):
    """Whether the backward-data pass of the image layer (64 filters over 1-3 channels, 3x3, padding 1) runs on the matrix cores
    (conv_few_mfma.hip; planner field few_mfma): per gradient pixel the 27 (tap, channel) sums as one bf16x6 product block, then a
    nine-value gather.  Measured against conv3x3_few_out_kernel (vector ALU): 256 x 256 9.6 vs 16.8 us, 512 x 512 20.3 vs 30.9,
    724 x 724 31.8 vs 48.9, 1024 x 1024 68.1 vs 79.5, 2048 x 2048 289 vs 295."""
    k, stride, pad = mod.kernel_size[0], mod.stride[0], mod.padding[0]
    return plan.on("few_mfma") and _x6_mode()[1] and k == 3 and stride == 1 and pad == 1 and hip.conv_few_mfma_supported(n, mod.in_channels, h, w, mod.out_channels, pad)


def conv_few_mfma(gy, mod, out):
    """gx = conv_backward(gy) of the image layer on conv_few_mfma.hip (the library picks the tile height)."""
    _route("conv_few_mfma", gy, mod.in_channels, 1, True, "4 / 8 x 62 px, bf16x6")
    return hip.conv3x3_few_mfma(gy, mod.bank_few_mfma(), mod.in_channels, out=out)


def conv_strided_fwd_is_3x3(mod, h, w):
    """Whether the FORWARD pass of a strided layer runs as space to depth + a stride-1 3x3 convolution over sites (planner field
    strided_fwd_3x3): kernel k <= 3 stride, no padding, stride^2 x in_channels consumed channels a multiple of 16 (conv_x3w's chunks), more
    than 32 produced channels, and a plane conv_x3w takes.  fp16x3 products (22-bit operands, per-chunk scale): measured 1.5e-7 from fp64 on
    image-range input, the direct fp32 kernel's own distance (tests/test_strided_as_3x3_gpu.py)."""
    k, stride, pad = mod.kernel_size[0], mod.stride[0], mod.padding[0]
    if not (plan.on("strided_fwd_3x3") and _x6_mode()[0] and _x3_enabled() and _x3w_enabled() and stride > 1 and pad == 0 and k <= 3 * stride and
            (stride * stride * mod.in_channels) % 16 == 0 and mod.out_channels > 32 and h >= k and w >= k):
        return False
    qh, qw = (h - k) // stride + 3, (w - k) // stride + 3
    return qh * qw >= _x3w_min_pixels() and bool(hip.conv_x3w_supported(stride * stride * mod.in_channels, qh, qw, 0))


def conv_strided_fwd_as_3x3(x, mod, out, relu, sites, workspace=None):
    """y[co][Y][X] = sum x[c][s Y + ky][s X + kx] w[co][c][ky][kx] with ky = ry + s my: x regrouped as s^2 c channels over sites
    (hip.space_to_depth into `sites`, (n, s^2 c, OH + 2, OW + 2): the sites the OH x OW windows reach; pixels beyond the image read 0 and
    meet zero taps), then a 3x3 stride-1 unpadded convolution with F[co][(ry s + rx) c_in + c][my][mx] = w[co][c][ry + s my][rx + s mx]
    (conv_x3w.hip, fp16x3)."""
    s_ = mod.stride[0]
    hip.space_to_depth(x, s_, sites)
    bank, wsc = mod.bank_strided_fwd()
    _route("conv_x3w", sites, mod.out_channels, 0, False, "64co x 8x32px", 1, strided_as_3x3=True, relu=relu)
    return hip.conv3x3_x3w(sites, bank, wsc, mod.bias_device(), mod.out_channels, 0, relu, out=out, workspace=workspace)


def conv_strided_bwd_as_3x3(gy, mod, out, workspace=None, sites=None):
    """Backward-data of a stride-s, kernel-k (k <= 3 s), unpadded layer.  gx[i] = sum over windows Y and taps t with s Y + t = i of gy[Y] w[t]:
    for i = s q + r that is t = r + s m, Y = q - m, m = 0, 1, 2 - a 3-tap sum over the output SITES q per dimension, i.e. a stride-1 3x3
    convolution with padding 2 over gy whose s^2 x in_channels output channels are the s x s pixels of a site (filters = the k x k filter in
    steps of s, zero where r + s m >= k), then depth to space.  The convolution is conv_x3w's (fp16x3, like every other backward pass);
    `out` = the (n, in_channels, H, W) gradient, written whole (pixels no window covers: 0)."""
    s_, c_img = mod.stride[0], mod.in_channels
    bank, wsc = mod.bank_strided_bwd()
    sites = hip.conv3x3_x3w(gy, bank, wsc, None, s_ * s_ * c_img, 2, False, out=sites, workspace=workspace)  # (`sites`: the caller's (n, s^2 c, OH + 2, OW + 2) buffer)
    _route("conv_x3w", gy, s_ * s_ * c_img, 2, True, "64co x 8x32px", 1, strided_as_3x3=True)
    return hip.depth_to_space(sites, s_, out)


def conv1x1_is_mfma(mod, backward):
    """Whether a layer's pass runs on the fp16x3 1x1 kernel (conv1x1_x3.hip): 1x1, stride 1, no padding (NIN's cccp layers,
    reference models.py:84-110), the split-precision path enabled for that pass and fp16x3 selected."""
    k, stride, pad = mod.kernel_size[0], mod.stride[0], mod.padding[0]
    produced = mod.in_channels if backward else mod.out_channels
    return _x6_mode()[1 if backward else 0] and _x3_enabled() and k == 1 and stride == 1 and pad == 0 and produced > 32


def conv1x1_mfma(x, mod, backward, out=None, out_relu_mask=None, relu=False, workspace=None):
    """1x1 layer on the fp16 matrix cores in fp16x3 arithmetic; backward-data multiplies by the transposed weights."""
    w2d, w2d_t = mod.mats()
    _route("conv1x1_x3", x, mod.in_channels if backward else mod.out_channels, 0, backward, "64co x 128px", mask=out_relu_mask is not None, relu=relu)
    if backward:
        return hip.conv1x1_x3(x, w2d_t, None, False, out=out, out_relu_mask=out_relu_mask, workspace=workspace)
    return hip.conv1x1_x3(x, w2d, mod.bias_device(), relu, out=out, out_relu_mask=out_relu_mask, workspace=workspace)


def conv5x5_is_mfma(mod, backward):
    """Whether a layer's pass runs on the fp16x3 k x k kernel (conv_kxk_x3.hip): 5x5, stride 1 (NIN's conv2, reference
    models.py:86), the split-precision path enabled for that pass and fp16x3 selected."""
    k, stride, pad = mod.kernel_size[0], mod.stride[0], mod.padding[0]
    produced = mod.in_channels if backward else mod.out_channels
    return _x6_mode()[1 if backward else 0] and _x3_enabled() and k == 5 and stride == 1 and pad <= 4 and produced > 32


def conv5x5_mfma(x, mod, backward, out=None, out_relu_mask=None, relu=False, workspace=None):
    bf, bb, wsc = mod.banks_kxk()
    k, pad = mod.kernel_size[0], mod.padding[0]
    _route("conv_kxk_x3", x, mod.in_channels if backward else mod.out_channels, k - 1 - pad if backward else pad, backward, "k x k, fp16x3", mask=out_relu_mask is not None, relu=relu)
    if backward:
        return hip.conv_kxk_x3(x, bb, wsc, None, mod.in_channels, k, k - 1 - pad, False, out=out, out_relu_mask=out_relu_mask,
                               workspace=workspace)
    return hip.conv_kxk_x3(x, bf, wsc, mod.bias_device(), mod.out_channels, k, pad, relu, out=out, out_relu_mask=out_relu_mask,
                           workspace=workspace)


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod):
        ctx.mod, ctx.in_shape = mod, x.shape
        k, stride, pad = mod.kernel_size[0], mod.stride[0], mod.padding[0]
        if _x6_mode()[0] and k == 3 and stride == 1 and pad <= 2 and mod.out_channels >= plan.get_int("split_min_produced"):
            return conv3x3_mfma(x.contiguous(), mod, False)
        if conv1x1_is_mfma(mod, False):
            return conv1x1_mfma(x.contiguous(), mod, False)
        if conv5x5_is_mfma(mod, False):
            return conv5x5_mfma(x.contiguous(), mod, False)
        return hip.conv2d_fwd(x.contiguous(), mod.banks()[0], mod.bias_device(), k, stride, pad, False)

    @staticmethod
    def backward(ctx, gy):
        mod = ctx.mod
        k, stride, pad = mod.kernel_size[0], mod.stride[0], mod.padding[0]
        if _x6_mode()[1] and k == 3 and stride == 1 and pad <= 2 and mod.in_channels >= plan.get_int("split_min_produced"):
            return conv3x3_mfma(gy.contiguous(), mod, True), None
        if conv1x1_is_mfma(mod, True):
            return conv1x1_mfma(gy.contiguous(), mod, True), None
        if conv5x5_is_mfma(mod, True):
            return conv5x5_mfma(gy.contiguous(), mod, True), None
        gx = hip.conv2d_bwd_data(gy.contiguous(), None, mod.banks()[1], mod.weight.detach(), ctx.in_shape, k, stride, pad)
        return gx, None


class Conv2d(nn.Conv2d):
    """nn.Conv2d whose forward/backward-data run in libmaua_hip (square kernels, symmetric stride/padding)."""

    def reset_parameters(self):  # weights always come from a state dict; skip the random init
        pass

    def banks(self):
        """(forward bank [taps][cin][cout], backward bank [taps][cout][cin]) cached per weight version."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if getattr(self, "_bank_key", None) != key:
            self._banks = hip.conv_pack_filters(self.weight.detach().contiguous())
            self._bank_key = key
        return self._banks

    def bank_image(self):
        """The bank of hip.conv3x3_image (layers that consume 1-3 channels)."""
        b = self.bias_device()
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device, None if b is None else (b.data_ptr(), b._version))
        if getattr(self, "_bank_img_key", None) != key:
            self._bank_img = hip.conv_pack_filters_image(self.weight.detach().contiguous(), b)
            self._bank_img_key = key
        return self._bank_img

    def banks6(self):
        """bf16x3 pre-split banks (forward, backward-data) of the fp32-accurate bf16 path, 3x3 filters only."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if getattr(self, "_bank6_key", None) != key:
            self._banks6 = hip.conv_pack_filters_x6(self.weight.detach().contiguous())
            self._bank6_key = key
        return self._banks6

    def banks3(self):
        """fp16x2 pre-split, pre-scaled banks (forward, backward-data, filter scale) of the fp16 three-product path."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if getattr(self, "_bank3_key", None) != key:
            self._banks3 = hip.conv_pack_filters_x3(self.weight.detach().contiguous())
            self._bank3_key = key
        return self._banks3

    def banks3w(self):
        """Banks of the wide-tile fp16x3 kernel (conv_x3w.hip): the same pre-split, pre-scaled fp16 pairs as banks3 in
        16-channel chunks, [chunk][cout tile][tap][part][octet][co][8 ch]."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if getattr(self, "_bank3w_key", None) != key:
            self._banks3w = hip.conv_pack_filters_x3w(self.weight.detach().contiguous())
            self._bank3w_key = key
        return self._banks3w

    def banks3q(self):
        """Banks of the 32-channel-chunk fp16x3 kernel (conv_x3q.hip): the same pre-split, pre-scaled fp16 pairs as banks3,
        [chunk of 32][cout tile][tap][part][octet][co][8 ch]."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if getattr(self, "_bank3q_key", None) != key:
            self._banks3q = hip.conv_pack_filters_x3q(self.weight.detach().contiguous())
            self._bank3q_key = key
        return self._banks3q

    def bank_few_mfma(self):
        """Bank of conv_few_mfma (the image layer's backward-data pass on the matrix cores)."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if getattr(self, "_bankfm_key", None) != key:
            self._bankfm = hip.conv_pack_filters_few_mfma(self.weight.detach().contiguous())
            self._bankfm_key = key
        return self._bankfm

    def bank_strided_fwd(self):
        """(bank, filter scale) of conv_strided_fwd_as_3x3 (conv_x3w's forward bank): F[co][(ry s + rx) c_in + c][my][mx] =
        w[co][c][ry + s my][rx + s mx], 0 beyond the k x k filter."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if getattr(self, "_banksf_key", None) != key:
            w = self.weight.detach()
            co, ci, k, _ = w.shape
            s_ = self.stride[0]
            wp = torch.zeros(co, ci, 3 * s_, 3 * s_, device=w.device, dtype=w.dtype)
            wp[:, :, :k, :k] = w
            f = wp.view(co, ci, 3, s_, 3, s_).permute(0, 3, 5, 1, 2, 4).reshape(co, s_ * s_ * ci, 3, 3).contiguous()
            bank, _, wsc = hip.conv_pack_filters_x3w(f)
            self._banksf = (bank, wsc)
            self._banksf_key = key
        return self._banksf

    def bank_strided_bwd(self):
        """(bank, filter scale) of conv_strided_bwd_as_3x3: the k x k filter regrouped as 3x3 taps over output sites,
        F[(ry s + rx) c_in + c][co][ty][tx] = w[co][c][ry + s (2 - ty)][rx + s (2 - tx)] (0 beyond the filter), as conv_x3w's FORWARD bank
        of a layer that consumes out_channels and produces s^2 c_in channels with padding 2."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if getattr(self, "_banksb_key", None) != key:
            w = self.weight.detach()
            co, ci, k, _ = w.shape
            s_ = self.stride[0]
            wp = torch.zeros(co, ci, 3 * s_, 3 * s_, device=w.device, dtype=w.dtype)
            wp[:, :, :k, :k] = w
            # wp[co][c][ry + s m_y][rx + s m_x] -> [ry][rx][c][co][m_y][m_x], taps t = 2 - m
            f = wp.view(co, ci, 3, s_, 3, s_).permute(3, 5, 1, 0, 2, 4).flip(4, 5).reshape(s_ * s_ * ci, co, 3, 3).contiguous()
            bank, _, wsc = hip.conv_pack_filters_x3w(f)
            self._banksb = (bank, wsc)
            self._banksb_key = key
        return self._banksb

    def banks_kxk(self):
        """fp16x2 pre-split, pre-scaled banks (forward, backward-data, filter scale) of the k x k fp16x3 kernel."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if getattr(self, "_bankk_key", None) != key:
            self._banksk = hip.conv_pack_filters_kxk_x3(self.weight.detach().contiguous())
            self._bankk_key = key
        return self._banksk

    def mats(self):
        """1x1 layers: the weights as plain matrices ([cout][cin], [cin][cout]) for the fp16x3 1x1 kernel."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if getattr(self, "_mats_key", None) != key:
            w2d = self.weight.detach().reshape(self.out_channels, self.in_channels).contiguous()
            self._mats = (w2d, w2d.t().contiguous())
            self._mats_key = key
        return self._mats

    def bias_device(self):
        return None if self.bias is None else self.bias.detach()

    def forward(self, x):
        return _ConvFn.apply(x, self)


class _ReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        hip.relu_(x)
        ctx.mark_dirty(x)
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        return hip.relu_bwd(gy.contiguous(), y)


class ReLU(nn.Module):
    """In-place ReLU, like the reference's nn.ReLU(inplace=True) (models.py:130)."""

    def __init__(self, inplace=True):
        super().__init__()
        self.inplace = inplace

    def forward(self, x):
        if not self.inplace:
            x = x.clone()
        elif x.requires_grad and x.is_leaf:
            x = x.clone()
        return _ReluFn.apply(x)


class _PoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k, stride, ceil, mode):
        xc = x.contiguous()
        ctx.save_for_backward(xc)
        ctx.cfg = (k, stride, ceil, mode)
        return hip.pool2d_fwd(xc, k, stride, ceil, mode)

    @staticmethod
    def backward(ctx, gy):
        (xc,) = ctx.saved_tensors
        k, stride, ceil, mode = ctx.cfg
        return hip.pool2d_bwd(gy.contiguous(), xc, k, stride, ceil, mode), None, None, None, None


def _as_int(v):
    return int(v[0]) if isinstance(v, (tuple, list)) else int(v)


class _Pool2d(nn.Module):
    mode = "max"

    def __init__(self, kernel_size, stride=None, padding=0, ceil_mode=False):
        super().__init__()
        if _as_int(padding) != 0:
            raise ValueError("pooling with padding is not used by any supported network")
        self.kernel_size, self.stride = kernel_size, (stride if stride is not None else kernel_size)
        self.padding, self.ceil_mode = padding, ceil_mode

    def forward(self, x):
        return _PoolFn.apply(x, _as_int(self.kernel_size), _as_int(self.stride), bool(self.ceil_mode), self.mode)

    def extra_repr(self):
        return f"kernel_size={self.kernel_size}, stride={self.stride}, ceil_mode={self.ceil_mode}"


class MaxPool2d(_Pool2d):
    mode = "max"


class AvgPool2d(_Pool2d):
    mode = "avg"


# --------------------------------------------------------------------------------------------------
# Architectures (reference models.py:16-139)
# --------------------------------------------------------------------------------------------------
channel_list = {
    "VGG-16p": [24, 22, "P", 41, 51, "P", 108, 89, 111, "P", 184, 276, 228, "P", 512, 512, 512, "P"],
    "VGG-16": [64, 64, "P", 128, 128, "P", 256, 256, 256, "P", 512, 512, 512, "P", 512, 512, 512, "P"],
    "VGG-19": [64, 64, "P", 128, 128, "P", 256, 256, 256, 256, "P", 512, 512, 512, 512, "P", 512, 512, 512, 512, "P"],
}


def _pool_layer(pooling, *a, **kw):
    if pooling == "max":
        return MaxPool2d(*a, **kw)
    if pooling == "avg":
        return AvgPool2d(*a, **kw)
    raise ValueError("Unrecognized pooling argseter")


def build_sequential(channels, pooling):
    """conv3x3(pad 1)+ReLU stacks separated by ONE shared 2x2 pooling module, as in the reference
    (models.py:116-132)."""
    pool2d = _pool_layer(pooling, kernel_size=2, stride=2)
    layers, cin = [], 3
    for c in channels:
        if c == "P":
            layers.append(pool2d)
        else:
            layers += [Conv2d(cin, c, kernel_size=3, padding=1), ReLU(inplace=True)]
            cin = c
    return nn.Sequential(*layers)


class _FeatureNet(nn.Module):
    def __init__(self, features):
        super().__init__()
        self.features = features


class VGG(_FeatureNet):
    pass


class VGG_SOD(_FeatureNet):
    pass


class VGG_FCN32S(_FeatureNet):
    pass


class VGG_PRUNED(_FeatureNet):
    pass


class NIN(nn.Module):
    """Network-in-Network feature stack (reference models.py:74-113); the trailing 1000-way conv, global
    pooling and softmax are kept so that state-dict indices line up, load_model never reaches them."""

    def __init__(self, pooling):
        super().__init__()
        pool2d = _pool_layer(pooling, (3, 3), (2, 2), (0, 0), ceil_mode=True)
        spec = [(3, 96, 11, 4, 0), (96, 96, 1, 1, 0), (96, 96, 1, 1, 0), "P",
                (96, 256, 5, 1, 2), (256, 256, 1, 1, 0), (256, 256, 1, 1, 0), "P",
                (256, 384, 3, 1, 1), (384, 384, 1, 1, 0), (384, 384, 1, 1, 0), "P", "D",
                (384, 1024, 3, 1, 1), (1024, 1024, 1, 1, 0), (1024, 1000, 1, 1, 0)]
        layers = []
        for s in spec:
            if s == "P":
                layers.append(pool2d)
            elif s == "D":
                layers.append(nn.Dropout(0.5))
            else:
                cin, cout, k, stride, pad = s
                layers += [Conv2d(cin, cout, (k, k), (stride, stride), (pad, pad)), ReLU(inplace=True)]
        layers += [nn.AvgPool2d((6, 6), (1, 1), (0, 0), ceil_mode=True), nn.Softmax()]
        self.features = nn.Sequential(*layers)


def _vgg_layer_names(channels):
    conv, relu, pool, block, i = [], [], [], 1, 1
    for c in channels:
        if c == "P":
            pool.append(f"pool{block}")
            block, i = block + 1, 1
        else:
            conv.append(f"conv{block}_{i}")
            relu.append(f"relu{block}_{i}")
            i += 1
    return {"C": conv, "R": relu, "P": pool}


vgg16_dict = _vgg_layer_names(channel_list["VGG-16"])
vgg19_dict = _vgg_layer_names(channel_list["VGG-19"])
nin_dict = {
    "C": ["conv1", "cccp1", "cccp2", "conv2", "cccp3", "cccp4", "conv3", "cccp5", "cccp6", "conv4-1024", "cccp7-1024",
          "cccp8-1024"],
    "R": [f"relu{i}" for i in range(1, 13)],
    "P": [f"pool{i}" for i in range(1, 5)],
    "D": ["drop"],
}

# keyword in --model_file -> (wrapper class, channel list, layer-name table, default file); checked in this order
_VGG_VARIANTS = [
    ("prun", VGG_PRUNED, "VGG-16p", vgg16_dict, "modelzoo/vgg16-prune.pth"),
    ("nyud", VGG_FCN32S, "VGG-16", vgg16_dict, "modelzoo/nyud-fcn32s-color-heavy.pth"),
    ("fcn32s", VGG_FCN32S, "VGG-16", vgg16_dict, "modelzoo/fcn32s-heavy-pascal.pth"),
    ("sod", VGG_SOD, "VGG-16", vgg16_dict, "modelzoo/vgg16-sod.pth"),
    ("vgg19", VGG, "VGG-19", vgg19_dict, "modelzoo/vgg19.pth"),
    ("vgg16", VGG, "VGG-16", vgg16_dict, "modelzoo/vgg16.pth"),
]


def _load_features(cnn, model_file, strict):
    """load_state_dict for the `features.*` keys only (this build has no classifier to receive the rest)."""
    sd = torch.load(model_file, map_location="cpu")
    feats = {k: v for k, v in sd.items() if k.startswith("features.")}
    own = cnn.state_dict()
    if strict:
        missing = [k for k in own if k not in feats]
        unexpected = [k for k in feats if k not in own]
        if missing or unexpected:
            raise RuntimeError(f"Error(s) in loading state_dict: missing {missing}, unexpected {unexpected}")
    cnn.load_state_dict({k: v for k, v in feats.items() if k in own}, strict=False)


def select_model(model_file, pooling, verbose, disable_check):
    """Pick the architecture from keywords in `model_file` and load its weights (reference models.py:246-347).
    Returns (cnn with a `.features` Sequential, layer-name table)."""
    vgg_keywords = ["fcn32s", "prun", "sod", "vgg", "nyud"]
    if any(k in model_file for k in vgg_keywords):
        for key, cls, chans, table, default in _VGG_VARIANTS:
            if key in model_file:
                if verbose:
                    print(("VGG-19" if chans == "VGG-19" else "VGG-16") + " Architecture Detected")
                cnn, layer_list = cls(build_sequential(channel_list[chans], pooling)), table
                break
        else:
            raise ValueError("VGG architecture not recognized.")
    elif "nin" in model_file:
        if verbose:
            print("NIN Architecture Detected")
        if pooling not in ("max", "avg"):
            raise ValueError("Unrecognized pooling argseter")
        cnn, layer_list, default = NIN(pooling), nin_dict, "modelzoo/nin.pth"
    else:
        raise ValueError("Model architecture not recognized.")
    if not path.exists(model_file):
        model_file = default
        if not path.exists(model_file):
            raise FileNotFoundError(
                f"model weights not found at {model_file}: this build has no network access and never downloads; "
                "pass --model_file <path to a .pth whose name contains the architecture keyword>")
    _load_features(cnn, model_file, strict=(not disable_check))
    if verbose:
        print("Successfully loaded " + str(model_file))
    return cnn, layer_list


def assemble(features, layer_list, args):
    """Insert the loss modules into the feature stack and truncate it (reference models.py:359-451).
    Pure module plumbing - no tensor is touched, so it also runs without a GPU."""
    content_layers = args.content_layers.split(",")
    style_layers = args.style_layers.split(",")
    features = copy.deepcopy(features)
    content_losses, style_losses, tv_losses, temporal_losses = [], [], [], []
    next_content_idx, next_style_idx = 1, 1
    net = nn.Sequential()
    c, r = 0, 0

    def add(mod, prefix, bucket):
        mod.name = f"{prefix} {len(net)}"
        net.add_module(str(len(net)), mod)
        bucket.append(mod)

    if args.tv_weight > 0:
        add(TVLoss(args.tv_weight), "tv", tv_losses)
    if args.temporal_weight > 0:
        add(ContentLoss(args.temporal_weight, args.normalize_gradients), "temporal", temporal_losses)

    def add_named(layer_name, index):
        found_content = found_style = False
        if layer_name in content_layers:
            if args.verbose:
                print("Setting up content layer " + str(index) + ": " + str(layer_name))
            add(ContentLoss(args.content_weight, args.normalize_gradients), "cont", content_losses)
            found_content = True
        if layer_name in style_layers:
            if args.verbose:
                print("Setting up style layer " + str(index) + ": " + str(layer_name))
            add(StyleLoss(args.style_weight, args.use_covariance, args.normalize_gradients,
                          video_style_factor=args.video_style_factor, shift_factor=args.shift_factor),
                "style", style_losses)
            found_style = True
        return found_content, found_style

    for i, layer in enumerate(list(features), 1):
        if not (next_content_idx <= len(content_layers) or next_style_idx <= len(style_layers)):
            continue
        if isinstance(layer, nn.Conv2d):
            net.add_module(str(len(net)), layer)
            add_named(layer_list["C"][c], i)
            c += 1
        if isinstance(layer, ReLU):
            net.add_module(str(len(net)), layer)
            fc, fs = add_named(layer_list["R"][r], i)
            next_content_idx += int(fc)  # only relu-named layers count towards "done"
            next_style_idx += int(fs)
            r += 1
        if isinstance(layer, _Pool2d):
            net.add_module(str(len(net)), layer)

    for param in net.parameters():
        param.requires_grad = False
    net.content_losses = content_losses
    net.style_losses = style_losses
    net.tv_losses = tv_losses
    net.temporal_losses = temporal_losses
    return net, content_losses + style_losses + tv_losses + temporal_losses


def load_model(args):
    """Build the loss network for `args` on the current ROCm device (reference models.py:351-453)."""
    cnn, layer_list = select_model(str(args.model_file).lower(), args.pooling, args.verbose, args.disable_check)
    if getattr(args, "multidevice", False):
        raise NotImplementedError(
            "layer-split multi-GPU (--gpu a,b / --multidevice_strategy) is a memory workaround for 11 GB cards and is "
            "not part of the MI355X build; shard frames/images across GPUs instead (dist.py)")
    if "c" in str(args.gpu).lower():
        raise RuntimeError("--gpu c: this build computes on MI355X through libmaua_hip only; there is no CPU path")
    hip.lib()  # fail here, loudly, when the HIP library is missing
    cnn = cnn.cuda()
    return assemble(cnn.features, layer_list, args)
