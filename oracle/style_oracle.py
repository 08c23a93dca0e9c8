"""CPU restatement of the reference hot path (test infrastructure; see oracle/__init__.py).

Every function cites the reference lines it follows.  Dense contractions use torch CPU
ops (the reference's own arithmetic backend); the backward pass, loss algebra and the
optimizers are written out explicitly - no autograd, no torch.optim.
"""
import math
from dataclasses import dataclass, field
from typing import List, Optional

import torch
import torch.nn.functional as F

# ----------------------------------------------------------------------------------------
# Layer tables (reference models.py:138-243).  Names are generated, not copied.
# ----------------------------------------------------------------------------------------
VGG19_CHANNELS = [64, 64, "P", 128, 128, "P", 256, 256, 256, 256, "P",
                  512, 512, 512, 512, "P", 512, 512, 512, 512, "P"]
VGG16_CHANNELS = [64, 64, "P", 128, 128, "P", 256, 256, 256, "P", 512, 512, 512, "P", 512, 512, 512, "P"]   # models.py:136
VGG16P_CHANNELS = [24, 22, "P", 41, 51, "P", 108, 89, 111, "P", 184, 276, 228, "P", 512, 512, 512, "P"]    # models.py:135 (channel-pruned)


def _vgg_channels(m):
    """Which VGG feature stack a checkpoint name selects (models.py:248-327: any of the VGG keywords, then in this order "prun" -> VGG-16p,
    "nyud" / "fcn32s" / "sod" -> VGG-16 (their classifiers differ, and are dropped), "vgg19", "vgg16"; otherwise a ValueError)."""
    if not any(k in m for k in ("fcn32s", "prun", "sod", "vgg", "nyud")):
        return None
    if "prun" in m:
        return VGG16P_CHANNELS
    if "nyud" in m or "fcn32s" in m or "sod" in m:
        return VGG16_CHANNELS
    if "vgg19" in m:
        return VGG19_CHANNELS
    if "vgg16" in m:
        return VGG16_CHANNELS
    raise ValueError("VGG architecture not recognized.")  # models.py:327


def _vgg_names(channels):
    conv, relu, block, i = [], [], 1, 1
    for c in channels:
        if c == "P":
            block, i = block + 1, 1
        else:
            conv.append(f"conv{block}_{i}")
            relu.append(f"relu{block}_{i}")
            i += 1
    return conv, relu


NIN_CONV_NAMES = ["conv1", "cccp1", "cccp2", "conv2", "cccp3", "cccp4", "conv3", "cccp5", "cccp6",
                  "conv4-1024", "cccp7-1024", "cccp8-1024"]
NIN_RELU_NAMES = [f"relu{i}" for i in range(1, 13)]
# (cout, cin, k, stride, pad) | "P" | "D"   (reference models.py:83-112)
NIN_FEATURES = [(96, 3, 11, 4, 0), (96, 96, 1, 1, 0), (96, 96, 1, 1, 0), "P",
                (256, 96, 5, 1, 2), (256, 256, 1, 1, 0), (256, 256, 1, 1, 0), "P",
                (384, 256, 3, 1, 1), (384, 384, 1, 1, 0), (384, 384, 1, 1, 0), "P", "D",
                (1024, 384, 3, 1, 1), (1024, 1024, 1, 1, 0), (1000, 1024, 1, 1, 0)]


@dataclass
class LayerSpec:
    kind: str  # tv | temporal | conv | relu | pool | style | content
    name: str = ""
    feat_idx: int = -1  # index into <model>.features (state-dict key) for convs
    cin: int = 0
    cout: int = 0
    k: int = 0
    stride: int = 1
    pad: int = 0
    pool_mode: str = "max"
    ceil: bool = False
    strength: float = 0.0
    normalize: bool = False
    use_covariance: bool = False
    video_style_factor: float = 0.0
    extra: dict = field(default_factory=dict)


def _feature_layers(model_file, pooling):
    """Flat list of the feature stack: ('conv', feat_idx, cout, cin, k, s, p) / ('relu',) / ('pool', k, s, ceil).
    VGG: models.py:116-132; NIN: models.py:74-113 (Dropout is dropped by load_model, models.py:381-438)."""
    m = str(model_file).lower()
    if pooling not in ("max", "avg"):
        raise ValueError("Unrecognized pooling argseter")  # models.py:124 (sic)
    out, idx = [], 0
    channels = _vgg_channels(m)
    if channels is not None:
        cin = 3
        for c in channels:
            if c == "P":
                out.append(("pool", 2, 2, False))
                idx += 1
            else:
                out.append(("conv", idx, c, cin, 3, 1, 1))
                out.append(("relu",))
                idx += 2
                cin = c
        return out, _vgg_names(channels)
    if "nin" in m:
        for spec in NIN_FEATURES:
            if spec == "P":
                out.append(("pool", 3, 2, True))
                idx += 1
            elif spec == "D":
                idx += 1
            else:
                c, cin, k, s, p = spec
                out.append(("conv", idx, c, cin, k, s, p))
                out.append(("relu",))
                idx += 2
        return out, (NIN_CONV_NAMES, NIN_RELU_NAMES)
    raise ValueError("Model architecture not recognized.")  # models.py:341


def build_spec(cfg) -> List[LayerSpec]:
    """Assemble the loss network the way models.load_model does (models.py:351-453):
    optional TV and temporal modules first, then feature layers with Content/Style modules
    inserted after the named layers, stopping once every requested relu layer was placed."""
    layers, (conv_names, relu_names) = _feature_layers(cfg.model_file, cfg.pooling)
    content_layers = cfg.content_layers.split(",")
    style_layers = cfg.style_layers.split(",")
    spec: List[LayerSpec] = []
    if cfg.tv_weight > 0:
        spec.append(LayerSpec("tv", name=f"tv {len(spec)}", strength=cfg.tv_weight))
    if cfg.temporal_weight > 0:
        spec.append(LayerSpec("temporal", name=f"temporal {len(spec)}", strength=cfg.temporal_weight,
                              normalize=cfg.normalize_gradients))
    next_c, next_s, c, r = 1, 1, 0, 0

    def add_losses(lname, is_relu):
        nonlocal next_c, next_s
        if lname in content_layers:
            spec.append(LayerSpec("content", name=f"cont {len(spec)}", strength=cfg.content_weight,
                                  normalize=cfg.normalize_gradients))
            if is_relu:
                next_c += 1
        if lname in style_layers:
            spec.append(LayerSpec("style", name=f"style {len(spec)}", strength=cfg.style_weight,
                                  normalize=cfg.normalize_gradients, use_covariance=cfg.use_covariance,
                                  video_style_factor=cfg.video_style_factor))
            if is_relu:
                next_s += 1

    for lay in layers:
        if not (next_c <= len(content_layers) or next_s <= len(style_layers)):
            break
        if lay[0] == "conv":
            _, fidx, cout, cin, k, s, p = lay
            spec.append(LayerSpec("conv", name=conv_names[c], feat_idx=fidx, cin=cin, cout=cout, k=k, stride=s, pad=p))
            add_losses(conv_names[c], False)
            c += 1
        elif lay[0] == "relu":
            spec.append(LayerSpec("relu", name=relu_names[r]))
            add_losses(relu_names[r], True)
            r += 1
        else:
            _, k, s, ceil = lay
            spec.append(LayerSpec("pool", k=k, stride=s, ceil=ceil, pool_mode=cfg.pooling))
    return spec


def loss_order(spec):
    """`losses` list order returned by load_model: content, style, tv, temporal (models.py:453)."""
    out = []
    for kind in ("content", "style", "tv", "temporal"):
        out += [i for i, l in enumerate(spec) if l.kind == kind]
    return out


# ----------------------------------------------------------------------------------------
# Loss algebra
# ----------------------------------------------------------------------------------------
def gram_matrix(x, use_covariance=False):
    """loss.GramMatrix.forward (loss.py:67-91) for the live code path (shift/flip args are dead):
    reshape to (B*C, H*W), optionally subtract each row's mean, return X X^T (un-normalised)."""
    B, C, H, W = x.shape
    xf = x.reshape(B * C, H * W)
    if use_covariance:
        xf = xf - xf.mean(1).unsqueeze(1)
    return xf @ xf.t()


def _scale_grad_coeff(incoming, strength):
    """Backward of loss.ScaleGradients (loss.py:17-20) for a scalar: g/(|g|+1e-8) * strength^2."""
    g = torch.as_tensor(float(incoming), dtype=torch.float64)
    return float(g / (g.abs() + 1e-8)) * strength * strength


class OracleNet:
    """Explicit forward/backward of the assembled loss network: B = 1 (img_img / vid_img paths) and windows of B > 1
    frames (img_vid: per-frame static style terms plus the cross-frame dynamic Gram term)."""

    def __init__(self, spec, state_dict, dtype=torch.float32):
        self.spec = spec
        self.dtype = dtype
        self.w, self.b = {}, {}
        for l in spec:
            if l.kind == "conv":
                self.w[l.feat_idx] = state_dict[f"features.{l.feat_idx}.weight"].to(dtype)
                self.b[l.feat_idx] = state_dict[f"features.{l.feat_idx}.bias"].to(dtype)
                prev = None
        for i, l in enumerate(spec):
            if l.kind in ("content", "style") and spec[i - 1].kind == "conv":
                # the reference's in-place ReLU rewrites the tensor these modules saved; autograd then
                # refuses the backward pass (version-counter error), so this layout never trains there.
                raise NotImplementedError("loss modules on conv-named layers are unusable in the reference")
        self.targets = {}  # layer index -> target tensor (content: feature map, style: CxC)
        self.video_targets = {}  # style layer index -> (B*C) x (B*C) target of StyleLoss.dynamic_loss
        self.strength = {i: l.strength for i, l in enumerate(spec) if l.kind in ("content", "style", "tv", "temporal")}

    # -- forward ---------------------------------------------------------------------------
    def _forward(self, x, keep=True):
        acts, aux = [], []
        h = x
        for l in self.spec:
            a = None
            if l.kind == "conv":
                h = F.conv2d(h, self.w[l.feat_idx], self.b[l.feat_idx], stride=l.stride, padding=l.pad)
            elif l.kind == "relu":
                h = torch.relu(h)
            elif l.kind == "pool":
                if l.pool_mode == "max":
                    h, a = F.max_pool2d(h, l.k, l.stride, 0, ceil_mode=l.ceil, return_indices=True)
                else:
                    h = F.avg_pool2d(h, l.k, l.stride, 0, ceil_mode=l.ceil)
            acts.append(h)
            aux.append(a)
        return acts, aux

    def _gram_normed(self, f, l):
        # StyleLoss.static_loss: gram / input[idx].nelement()  (loss.py:144)
        return gram_matrix(f, l.use_covariance) / f[0].nelement()

    def capture_content(self, image):
        """optim.set_content_targets (optim.py:22-32) -> ContentLoss 'capture' (loss.py:61-62)."""
        acts, _ = self._forward(image.to(self.dtype))
        for i, l in enumerate(self.spec):
            if l.kind == "content":
                self.targets[i] = acts[i].clone()

    def capture_temporal(self, warp_image, warp_weights=None):
        """optim.set_temporal_targets (optim.py:35-47): the pixel-level ContentLoss captures the warped previous frame as
        it is (loss.py:61-62) and keeps the reliability mask as `.weights` (applied to the INPUT only, loss.py:52-53)."""
        for i, l in enumerate(self.spec):
            if l.kind == "temporal":
                self.targets[i] = warp_image.to(self.dtype).clone()
                self.temporal_weights = None if warp_weights is None else warp_weights.to(self.dtype).clone()

    def capture_style(self, images, blend_weights):
        """optim.set_style_targets (optim.py:50-66) -> StyleLoss.static_loss 'capture' (loss.py:146-151):
        target = sum_i blend_i * Gram_i / (C*H*W) / B.  (The dynamic target, loss.py:170-175, equals it for B=1.)"""
        for i, l in enumerate(self.spec):
            if l.kind == "style":
                self.targets.pop(i, None)
        for img, bw in zip(images, blend_weights):
            acts, _ = self._forward(img.to(self.dtype))
            for i, l in enumerate(self.spec):
                if l.kind == "style":
                    g = bw * self._gram_normed(acts[i], l) / img.shape[0]
                    self.targets[i] = g if i not in self.targets else self.targets[i] + g

    def capture_style_videos(self, videos, blend_weights, window):
        """optim.set_style_video_targets (optim.py:69-90): every window of `window` consecutive frames of every style clip
        goes through the net in 'capture' mode with blend weight / number of windows.  StyleLoss.static_loss adds
        blend * Gram(frame) / (C*H*W) / B for every frame of the window (loss.py:141-151), StyleLoss.dynamic_loss adds
        blend * Gram(window as (B*C) rows) / (B*C*H*W) to the video target when video_style_factor > 0 - unless a video
        target of another size already exists (loss.py:164-175)."""
        for i, l in enumerate(self.spec):
            if l.kind == "style":
                self.targets.pop(i, None)
                self.video_targets.pop(i, None)
        for video, bw in zip(videos, blend_weights):
            n_windows = max(len(video) - window + 1, 1)
            w_blend = bw / n_windows
            for start in range(n_windows):
                frames = video[start:start + window].to(self.dtype)
                acts, _ = self._forward(frames)
                for i, l in enumerate(self.spec):
                    if l.kind != "style":
                        continue
                    f = acts[i]
                    for b in range(f.shape[0]):
                        g = w_blend * self._gram_normed(f[b:b + 1], l) / f.shape[0]
                        self.targets[i] = g if i not in self.targets else self.targets[i] + g
                    if l.video_style_factor > 0:
                        rows = f.shape[0] * f.shape[1]
                        if i in self.video_targets and self.video_targets[i].shape[0] != rows:
                            continue
                        gd = w_blend * gram_matrix(f, l.use_covariance) / f.nelement()
                        self.video_targets[i] = gd if i not in self.video_targets else self.video_targets[i] + gd

    def normalize_weights(self):
        """optim.py:176-178: strength /= max(target.size()); raises like the reference when a
        temporal module with an empty target is present (ZeroDivisionError)."""
        for i, l in enumerate(self.spec):
            if l.kind in ("content", "style", "temporal"):
                size = self.targets[i].shape if i in self.targets else torch.Size([0])
                self.strength[i] = self.strength[i] / max(size)

    # -- one function evaluation -----------------------------------------------------------
    def feval(self, x):
        """feval closure (optim.py:201-238) for B=1: returns (total, {layer idx: loss}, grad wrt x).

        Closed form (SURVEY §8 a-spec).  Reported loss weights vs gradient weights differ when
        normalize_gradients is on because ScaleGradients is applied to the scalar MSE."""
        x = x.to(self.dtype)
        if x.shape[0] > 1:
            return self._feval_window(x)
        acts, aux = self._forward(x)
        losses, inject = {}, {}
        for i, l in enumerate(self.spec):
            s = self.strength.get(i, 0.0)
            if l.kind == "tv":
                # loss.TVLoss (loss.py:224-233)
                dv = x[:, :, 1:, :] - x[:, :, :-1, :]
                dh = x[:, :, :, 1:] - x[:, :, :, :-1]
                losses[i] = s * (dv.abs().sum() + dh.abs().sum())
                g = torch.zeros_like(x)
                sv, sh = torch.sign(dv), torch.sign(dh)
                g[:, :, 1:, :] += sv
                g[:, :, :-1, :] -= sv
                g[:, :, :, 1:] += sh
                g[:, :, :, :-1] -= sh
                inject[i] = s * g
            elif l.kind == "temporal":
                if i not in self.targets:
                    continue  # empty target in "loss" mode -> early return (loss.py:46-47)
                # ContentLoss.forward on the pixels (loss.py:49-59): MSE(x * weights, target), weights on the input only
                t, w = self.targets[i], getattr(self, "temporal_weights", None)
                if x.shape[1:] != t.shape[1:]:
                    continue  # loss.py:44
                diff = (x * w if w is not None else x) - t
                mse = (diff * diff).mean()
                losses[i] = mse * s
                coeff = _scale_grad_coeff(s, s) if l.normalize else s
                gi = diff * (coeff * 2.0 / diff.numel())
                inject[i] = gi * w if w is not None else gi
            elif l.kind == "content":
                # ContentLoss.forward (loss.py:49-59), B = 1
                f, t = acts[i], self.targets[i]
                if f.shape[1:] != t.shape[1:]:
                    continue  # loss.py:44
                diff = f - t
                mse = (diff * diff).mean()
                losses[i] = mse * s
                coeff = _scale_grad_coeff(s, s) if l.normalize else s
                inject[i] = diff * (coeff * 2.0 / diff.numel())
            elif l.kind == "style":
                # StyleLoss.static_loss + dynamic_loss (loss.py:141-181), B = 1: both use the same Gram
                f, t = acts[i], self.targets[i]
                C = f.shape[1]
                n = f[0].nelement()
                G = self._gram_normed(f, l)
                d = G - t
                mse = (d * d).mean()
                vsf = l.video_style_factor
                losses[i] = mse * s + (vsf * mse * s if vsf > 0 else 0.0)
                if l.normalize:
                    coeff = _scale_grad_coeff(s, s) + (_scale_grad_coeff(vsf * s, s) if vsf > 0 else 0.0)
                else:
                    coeff = s + (vsf * s if vsf > 0 else 0.0)
                # d mse / dF = (2/C^2) * (1/n) * (D + D^T) F_c ; D symmetric -> 2 D F_c
                ff = f.reshape(C, -1)
                if l.use_covariance:
                    ff = ff - ff.mean(1).unsqueeze(1)
                gf = (coeff * 2.0 / (C * C) / n) * ((d + d.t()) @ ff)
                if l.use_covariance:
                    gf = gf - gf.mean(1).unsqueeze(1)  # backward of the row-mean subtraction
                inject[i] = gf.reshape(f.shape)
        g = self._backward(x, acts, aux, inject)
        total = sum(losses.values())
        return total, losses, g

    def _backward(self, x, acts, aux, inject):
        """Backward to the pixels: the injected loss gradients flow down through relu / conv / pool."""
        g = None
        for i in range(len(self.spec) - 1, -1, -1):
            l = self.spec[i]
            if i in inject:
                g = inject[i] if g is None else g + inject[i]
            if g is None:
                continue
            inp = acts[i - 1] if i > 0 else x
            if l.kind == "relu":
                g = g * (acts[i] > 0).to(g.dtype)
            elif l.kind == "conv":
                g = torch.nn.grad.conv2d_input(inp.shape, self.w[l.feat_idx], g, stride=l.stride, padding=l.pad)
            elif l.kind == "pool":
                if l.pool_mode == "max":
                    gi = torch.zeros(inp.shape[0], inp.shape[1], inp.shape[2] * inp.shape[3], dtype=g.dtype)
                    gi.scatter_add_(2, aux[i].reshape(*aux[i].shape[:2], -1), g.reshape(*g.shape[:2], -1))
                    g = gi.reshape(inp.shape)
                else:
                    g = _avg_pool_backward(g, inp.shape, l.k, l.stride, l.ceil)
        return g

    def _feval_window(self, x):
        """feval for a window of B > 1 frames (img_vid).  TV over the whole batch (loss.py:224-233); ContentLoss loops over the
        frames against the single-frame target, each entering with strength / B (loss.py:48-58); StyleLoss: static term
        per frame against the C x C target with strength / B (loss.py:141-158), dynamic term on the (B*C) x (B*C) Gram of
        the window against the video target with video_style_factor * strength / B, skipped when the sizes differ
        (loss.py:164-181).  With normalize_gradients every scalar MSE passes ScaleGradients: its gradient weight becomes
        sign(incoming) * strength^2, whatever the 1 / B."""
        B = x.shape[0]
        acts, aux = self._forward(x)
        losses, inject = {}, {}
        for i, l in enumerate(self.spec):
            s = self.strength.get(i, 0.0)
            if l.kind == "tv":
                dv = x[:, :, 1:, :] - x[:, :, :-1, :]
                dh = x[:, :, :, 1:] - x[:, :, :, :-1]
                losses[i] = s * (dv.abs().sum() + dh.abs().sum())
                g = torch.zeros_like(x)
                sv, sh = torch.sign(dv), torch.sign(dh)
                g[:, :, 1:, :] += sv
                g[:, :, :-1, :] -= sv
                g[:, :, :, 1:] += sh
                g[:, :, :, :-1] -= sh
                inject[i] = s * g
            elif l.kind == "temporal":
                if i in self.targets:
                    raise NotImplementedError("temporal targets on a window of frames are not part of img_vid")
            elif l.kind == "content":
                f, t = acts[i], self.targets[i]
                if f.shape[1:] != t.shape[1:]:
                    continue
                assert t.shape[0] == 1
                coeff = _scale_grad_coeff(s / B, s) if l.normalize else s / B
                diff = f - t  # every frame against the same target
                n = f[0].nelement()
                losses[i] = sum(((diff[b] * diff[b]).mean() * s / B for b in range(B)))
                inject[i] = diff * (coeff * 2.0 / n)
            elif l.kind == "style":
                f, t = acts[i], self.targets[i]
                C, n = f.shape[1], f[0].nelement()
                coeff = _scale_grad_coeff(s / B, s) if l.normalize else s / B
                loss = 0.0
                gf_all = torch.zeros_like(f)
                for b in range(B):
                    fb = f[b:b + 1]
                    d = self._gram_normed(fb, l) - t
                    loss = loss + (d * d).mean() * s / B
                    ff = fb.reshape(C, -1)
                    if l.use_covariance:
                        ff = ff - ff.mean(1).unsqueeze(1)
                    gfb = (coeff * 2.0 / (C * C) / n) * ((d + d.t()) @ ff)
                    if l.use_covariance:
                        gfb = gfb - gfb.mean(1).unsqueeze(1)
                    gf_all[b] = gfb.reshape(f.shape[1:])
                vsf = l.video_style_factor
                if vsf > 0:
                    vt = self.video_targets[i]
                    if vt.shape[0] == B * C:
                        nall = f.nelement()
                        d = gram_matrix(f, l.use_covariance) / nall - vt
                        loss = loss + vsf * (d * d).mean() * s / B
                        cd = _scale_grad_coeff(vsf * s / B, s) if l.normalize else vsf * s / B
                        ff = f.reshape(B * C, -1)
                        if l.use_covariance:
                            ff = ff - ff.mean(1).unsqueeze(1)
                        gd = (cd * 2.0 / (B * C) ** 2 / nall) * ((d + d.t()) @ ff)
                        if l.use_covariance:
                            gd = gd - gd.mean(1).unsqueeze(1)
                        gf_all = gf_all + gd.reshape(f.shape)
                losses[i] = loss
                inject[i] = gf_all
        g = self._backward(x, acts, aux, inject)
        total = sum(losses.values())
        return total, losses, g


def _avg_pool_backward(g, in_shape, k, stride, ceil):
    """Adjoint of F.avg_pool2d(pad=0, count_include_pad default) used by --pooling avg (models.py:122)."""
    H, W = in_shape[2:]
    out = torch.zeros(in_shape, dtype=g.dtype)
    OH, OW = g.shape[2:]
    for oy in range(OH):
        y0, y1 = oy * stride, min(oy * stride + k, H)
        for ox in range(OW):
            x0, x1 = ox * stride, min(ox * stride + k, W)
            out[:, :, y0:y1, x0:x1] += (g[:, :, oy, ox] / ((y1 - y0) * (x1 - x0)))[:, :, None, None]
    return out


# ----------------------------------------------------------------------------------------
# Optimizers (torch.optim.LBFGS / Adam as configured at optim.py:180-196), restated
# ----------------------------------------------------------------------------------------
class _LbfgsState:
    def __init__(self):
        self.n_iter = 0
        self.func_evals = 0
        self.d = None
        self.t = None
        self.Y: List[torch.Tensor] = []  # gradient differences ("old_dirs" in torch)
        self.S: List[torch.Tensor] = []  # steps ("old_stps")
        self.rho: List[float] = []
        self.h_diag = 1.0
        self.g_prev: Optional[torch.Tensor] = None


def _lbfgs_step(x, closure, st, max_iter, history, lr=1.0, tol_grad=-1.0, tol_change=-1.0, trace=None):
    """One call of LBFGS.step without line search: the iteration structure of torch/optim/lbfgs.py
    (first direction -g with t=min(1,1/|g|_1); afterwards curvature-pair update guarded by y.s>1e-10,
    two-loop recursion, unit step, re-evaluation except on the last iteration, max_eval = 5*max_iter//4)."""
    max_eval = max_iter * 5 // 4
    loss, g = closure(x)
    evals = 1
    st.func_evals += 1
    if float(g.abs().max()) <= tol_grad:
        return x
    n_iter = 0
    while n_iter < max_iter:
        n_iter += 1
        st.n_iter += 1
        if st.n_iter == 1:
            d = -g
            st.Y, st.S, st.rho, st.h_diag = [], [], [], 1.0
        else:
            # scalars stay 0-dim tensors of the working dtype, as in torch (fp32 scalar arithmetic)
            y = g - st.g_prev
            s = st.d * st.t
            ys = y.dot(s)
            if float(ys) > 1e-10:
                if len(st.Y) == history:
                    st.Y.pop(0), st.S.pop(0), st.rho.pop(0)
                st.Y.append(y)
                st.S.append(s)
                st.rho.append(1.0 / ys)
                st.h_diag = ys / y.dot(y)
            m = len(st.Y)
            alpha = [None] * m
            q = -g
            for i in range(m - 1, -1, -1):
                alpha[i] = st.S[i].dot(q) * st.rho[i]
                q = q - float(alpha[i]) * st.Y[i]
            d = q * st.h_diag
            for i in range(m):
                beta = st.Y[i].dot(d) * st.rho[i]
                d = d + float(alpha[i] - beta) * st.S[i]
        st.g_prev = g.clone()
        prev_loss = loss
        t = min(1.0, 1.0 / g.abs().sum()) * lr if st.n_iter == 1 else lr  # tensor when < 1
        gtd = float(g.dot(d))
        st.d, st.t = d, t
        if gtd > -tol_change:
            break
        x = x + float(t) * d
        if trace is not None:
            trace.append(x.clone())
        if n_iter != max_iter:
            loss, g = closure(x)
            evals += 1
            st.func_evals += 1
        if n_iter == max_iter or evals >= max_eval:
            break
        if float(g.abs().max()) <= tol_grad:
            break
        if float((d * t).abs().max()) <= tol_change:
            break
        if abs(loss - prev_loss) < tol_change:
            break
    return x


def lbfgs_run(fg, x0, num_iters, history=100, tol_grad=-1.0, tol_change=-1.0, trace=None, stats=None):
    """The reference's L-BFGS driver: `LBFGS(max_iter=num_iters)` and `while i[0] <= 1: step(feval)`
    (optim.py:180-191, 240-241) where i[0] counts fevals - so num_iters == 1 runs step() twice.
    fg(x_flat) -> (loss: float, grad_flat)."""
    st = _LbfgsState()
    calls = [0]

    def closure(x):
        calls[0] += 1
        loss, g = fg(x)
        return float(loss), g

    x = x0.clone().flatten()
    while calls[0] <= 1:
        x = _lbfgs_step(x, closure, st, num_iters, history, 1.0, tol_grad, tol_change, trace)
    if stats is not None:
        stats.update(history_len=len(st.Y), n_iter=st.n_iter, func_evals=st.func_evals)
    return x.reshape(x0.shape), calls[0]


def adam_run(fg, x0, num_iters, lr=1.0, betas=(0.9, 0.999), eps=1e-8):
    """torch.optim.Adam([x], lr) single-tensor update, driven by `while i[0] <= num_iters`
    (optim.py:192-196, 240-241) -> num_iters + 1 steps."""
    x = x0.clone().flatten()
    m = torch.zeros_like(x)
    v = torch.zeros_like(x)
    b1, b2 = betas
    calls = 0
    step = 0
    while calls <= num_iters:
        _, g = fg(x)
        calls += 1
        step += 1
        m = m + (g - m) * (1 - b1)  # exp_avg.lerp_(grad, 1-beta1)
        v = v * b2 + (g * g) * (1 - b2)
        bc1 = 1 - b1 ** step
        bc2 = 1 - b2 ** step
        denom = v.sqrt() / math.sqrt(bc2) + eps
        x = x - (lr / bc1) * (m / denom)
    return x.reshape(x0.shape), calls


def optimize(content, styles, init, num_iters, cfg, state_dict, dtype=torch.float32, trace=None, temporal=None):
    """optim.optimize (optim.py:111-255) for transfer types without '_vid' and B = 1.  `temporal` = (warp_image,
    warp_weights) when the caller had run optim.set_temporal_targets on the prebuilt net (style.py:281)."""
    spec = build_spec(cfg)
    net = OracleNet(spec, state_dict, dtype)
    if temporal is not None:
        net.capture_temporal(*temporal)
    net.capture_content(content)
    net.capture_style(styles, cfg.style_blend_weights)
    if getattr(cfg, "normalize_weights", False):
        net.normalize_weights()
    shape = init.shape

    def fg(xf):
        total, _, g = net.feval(xf.reshape(shape))
        return float(total), g.flatten()

    x0 = init.to(dtype)
    if cfg.optimizer == "lbfgs":
        out, _ = lbfgs_run(fg, x0, num_iters, history=cfg.lbfgs_num_correction,
                           tol_grad=float(cfg.lbfgs_tolerance_grad), tol_change=float(cfg.lbfgs_tolerance_change),
                           trace=trace)
    elif cfg.optimizer == "adam":
        out, _ = adam_run(fg, x0, num_iters, lr=cfg.learning_rate)
    else:
        raise ValueError(cfg.optimizer)
    return out


# ----------------------------------------------------------------------------------------
# img_vid: sliding windows of frames (optim.py:111-255, the '_vid' branches)
# ----------------------------------------------------------------------------------------
def wrapping_indices(n_frames, start, length):
    """utils.wrapping_slice(..., return_indices=True) (utils.py:76-85)."""
    if n_frames == 1:
        return torch.zeros(1, dtype=torch.int64)
    if start + length <= n_frames:
        return torch.arange(start, start + length)
    return torch.cat((torch.arange(start, n_frames), torch.arange(0, (start + length) % n_frames)))


def video_windows(lengths, window):
    """optim.py:113-123: window starts over the pastiche (lengths[0]) and over every style clip."""
    num_windows = math.ceil(lengths[0] / window)
    step = [(n - window / 2) / num_windows for n in lengths]
    return [[math.ceil(step[k] * j) for j in range(num_windows + 1)] if n != 1 else [0] * (num_windows + 1)
            for k, n in enumerate(lengths)]


def optimize_video(content, styles, init, num_iters, cfg, state_dict, window, avg_frame_window=18, dtype=torch.float32):
    """optim.optimize for transfer types with '_vid': the clip `init` (T frames) is optimised `window` frames at a time; the
    style-video targets come from the whole clips once (avg_frame_window == -1, optim.py:140-144) or, per window, from the
    `avg_frame_window` frames of every clip that start at that clip's own window start (optim.py:158-165); frames already
    styled by the previous window (front overlap) and, past the wrap-around, by the first (end overlap) get a zero
    gradient (optim.py:217-221); the window is written back through the wrapping index (optim.py:243-245)."""
    spec = build_spec(cfg)
    net = OracleNet(spec, state_dict, dtype)
    net.capture_content(content)
    windows = video_windows([init.shape[0]] + [v.shape[0] for v in styles], window)
    if avg_frame_window == -1:
        net.capture_style_videos(styles, cfg.style_blend_weights, window)
    output = init.clone().to(dtype)
    T = output.shape[0]
    for w, start in enumerate(windows[0]):
        front = windows[0][w - 1] + window - start
        end = (start + window) % T if start + window >= T else 0
        idx = wrapping_indices(T, start, window)
        if avg_frame_window != -1:
            current = [v[wrapping_indices(v.shape[0], windows[k + 1][w], avg_frame_window)] for k, v in enumerate(styles)]
            net.capture_style_videos(current, cfg.style_blend_weights, window)
        if w == 0 and getattr(cfg, "normalize_weights", False):
            net.normalize_weights()
        x0 = output[idx]
        shape = x0.shape

        def fg(xf, w=w, front=front, end=end):
            total, _, g = net.feval(xf.reshape(shape))
            g = g.clone()
            if w != 0:
                g[:front] = 0
                if end > 0:
                    g[-end:] = 0
            return float(total), g.flatten()

        if cfg.optimizer == "lbfgs":
            out, _ = lbfgs_run(fg, x0, num_iters, history=cfg.lbfgs_num_correction, tol_grad=float(cfg.lbfgs_tolerance_grad),
                               tol_change=float(cfg.lbfgs_tolerance_change))
        elif cfg.optimizer == "adam":
            out, _ = adam_run(fg, x0, num_iters, lr=cfg.learning_rate)
        else:
            raise ValueError(cfg.optimizer)
        output[idx] = out.reshape(shape).to(output.dtype)
    return output


# ---------------------------------------------------------------------------------------------------------
# Image-space steps around the loop, CPU restatements
# ---------------------------------------------------------------------------------------------------------
def _colour_stats(t_bwhc, eps):
    """reference utils.py:88-93: mean over every dim but the last, the centred tensor reshaped to (C, -1) exactly as written
    there (`h.permute(0, 3, 1, 2).reshape(C, -1)`: for a batch of more than one frame the rows are NOT the channels - the
    B*C planes in (frame, channel) order are cut into C runs of B planes), and its covariance + eps I."""
    mu = t_bwhc.mean(list(range(t_bwhc.dim() - 1)))
    rows = (t_bwhc - mu).permute(0, 3, 1, 2).reshape(t_bwhc.size(3), -1)
    return mu, rows, rows @ rows.T / rows.shape[1] + eps * torch.eye(rows.shape[0], dtype=t_bwhc.dtype)


def _psd_sqrt(cov):
    """utils.py:127-131 with torch.linalg.eigh(UPLO="U") standing in for the removed torch.symeig(upper=True)."""
    vals, vecs = torch.linalg.eigh(cov, UPLO="U")
    root = torch.sqrt(torch.diagflat(vals))
    root[root != root] = 0
    return vecs @ root @ vecs.T


def match_histogram(target, sources, eps=1e-2, mode="avg", dtype=None):
    """reference utils.py:96-151 (PCA colour transfer; jitter from torch's global generator and, for the random-frame
    mode, numpy's, in the reference's order).  `dtype=torch.float64` evaluates the same formula in double (the jitter is
    still drawn in fp32, so both precisions see the same numbers)."""
    import numpy as np
    if not mode:
        return target
    backup = target.clone()
    work = dtype or target.dtype
    try:
        per_frame = mode == "avg"
        sources = sources if isinstance(sources, list) else [sources]
        out = torch.zeros_like(target, dtype=work)
        for source in sources:
            tgt = target.permute(0, 3, 2, 1).to(work)
            src = source.permute(0, 3, 2, 1).to(work)
            src = src.mean(0).unsqueeze(0) if per_frame else src[np.random.randint(0, src.shape[0])].unsqueeze(0)
            matched = torch.zeros_like(tgt)
            for idx in range(tgt.shape[0] if per_frame else 1):
                frame = tgt[idx].unsqueeze(0) if per_frame else tgt
                _, t, cov_t = _colour_stats(frame + (1e-3 * torch.randn(size=frame.shape)).to(work), eps)
                mu_s, _, cov_s = _colour_stats(src + (1e-3 * torch.randn(size=src.shape)).to(work), eps)
                ts = _psd_sqrt(cov_s) @ torch.inverse(_psd_sqrt(cov_t)) @ t
                m = ts.reshape(*frame.permute(0, 3, 1, 2).shape).permute(0, 2, 3, 1) + mu_s
                if per_frame:
                    matched[idx] = m
                else:
                    matched = m
            out += matched.permute(0, 3, 2, 1) / len(sources)
        return out
    except RuntimeError:
        return backup


def resize_bilinear(x, size=None, scale_factor=None):
    """style.py:38-66: F.interpolate(mode="bilinear", align_corners=False) in the two forms the reference uses."""
    if scale_factor is not None:
        return F.interpolate(x, scale_factor=scale_factor, mode="bilinear", align_corners=False)
    return F.interpolate(x, size, mode="bilinear", align_corners=False)


def deprocess_u8(x):
    """load.py:47-52 (+ torchvision ToPILImage's `mul(255).byte()`): (1,3,H,W) BGR mean-subtracted -> (H,W,3) uint8 RGB."""
    mean = torch.tensor([103.939, 116.779, 123.68])
    t = x.squeeze(0).float() + mean[:, None, None]
    rgb = (t[torch.LongTensor([2, 1, 0])] / 255).clamp_(0, 1)
    return rgb.mul(255).byte().permute(1, 2, 0).contiguous()
