"""Loss modules with the reference's names and attributes (reference loss.py), computing on MI355X.

Every tensor-sized operation goes through libmaua_hip (hip.py): Gram / covariance matrices on the fp32
matrix cores, fused MSE forward+backward, total variation.  The modules keep the reference's protocol -
they sit inside the feature network, return their input unchanged and leave the scalar in `.loss`
(`mode` in {"none", "capture", "loss"}) - so code written against the reference's loss.py keeps working,
including autograd: each op is a torch.autograd.Function whose backward is again a HIP kernel.

The iteration loop itself does not run these modules one by one: optim.optimize hands the assembled
network to engine.StyleEngine, which reads `.target`, `.strength`, `.normalize` ... from them and
chains the kernels without autograd.  Both paths produce the same numbers (tests/test_engine_gpu.py).
"""
import numpy as np  # noqa: F401  (kept importable like the reference module)
import torch
import torch.nn as nn

import hip
from utils import info  # noqa: F401


class ScaleGradients(torch.autograd.Function):
    """Identity in the forward pass; the backward pass replaces the incoming gradient g by
    g / (||g|| + 1e-8) * strength^2  (reference loss.py:10-20).  It is applied to the *scalar* MSE, so the
    gradient weight of a loss becomes +-strength^2 whatever the reported weight is."""

    @staticmethod
    def forward(ctx, input_tensor, strength):
        ctx.strength = strength
        return input_tensor

    @staticmethod
    def backward(ctx, grad_output):
        unit = grad_output / (torch.norm(grad_output, keepdim=True) + 1e-8)
        return unit * ctx.strength * ctx.strength, None


def normalize_weights(content_losses, style_losses):
    """Divide every strength by the largest dimension of its target (reference loss.py:24-28)."""
    for mod in list(content_losses) + list(style_losses):
        mod.strength = mod.strength / max(mod.target.size())


class _MseFn(torch.autograd.Function):
    """mean((x - target)^2) with the gradient 2 (x - target) / n produced by the same kernel pass."""

    @staticmethod
    def forward(ctx, x, target):
        xc = x.contiguous()
        tc = target.expand_as(x).contiguous()
        n = xc.numel()
        loss = torch.zeros(1, device=x.device, dtype=torch.float32)
        grad = torch.empty_like(xc)
        hip.mse_fwd_bwd(xc, tc, grad, 1.0 / n, 2.0 / n, False, loss)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g, None


class _GramFn(torch.autograd.Function):
    """X X^T of X = x.reshape(B*C, H*W) (rows optionally centred); backward (dG + dG^T) X."""

    @staticmethod
    def forward(ctx, x, use_covariance):
        B, C, H, W = x.shape
        xc = x.contiguous()
        gram, mean = hip.gram_fwd(xc.reshape(1, B * C, H, W), 1.0, bool(use_covariance))
        ctx.save_for_backward(xc)
        ctx.mean = mean
        return gram

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        d = (g + g.t()).contiguous()
        gf = torch.empty_like(xc)
        hip.gram_bwd(d, xc, ctx.mean, gf, False)
        return gf, None


class _TVFn(torch.autograd.Function):
    """sum |x[:,:,1:,:]-x[:,:,:-1,:]| + sum |x[:,:,:,1:]-x[:,:,:,:-1]| and its sign() gradient, one kernel."""

    @staticmethod
    def forward(ctx, x):
        xc = x.contiguous()
        loss = torch.zeros(1, device=x.device, dtype=torch.float32)
        grad = torch.empty_like(xc)
        hip.tv_fwd_bwd(xc, grad, 1.0, False, loss)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g


class ContentLoss(nn.Module):
    """Feature-space MSE against a captured target (reference loss.py:32-64).  Also used on the pixels as the
    'temporal' loss, with optional per-pixel `weights`."""

    def __init__(self, strength, normalize=False):
        super().__init__()
        self.strength = strength
        self.crit = _MseFn.apply
        self.mode = "none"
        self.weights = None
        self.normalize = normalize
        self.loss = 0
        self.target = torch.Tensor()

    def forward(self, input):
        if self.mode == "none" or (input.shape[1:] != self.target.shape[1:] and self.target.nelement() != 0):
            return input
        if "temporal" in self.name and self.target.shape[0] == 0 and self.mode == "loss":
            return input
        self.loss = 0
        frames = input.shape[0]
        for idx in range(frames):
            if self.mode == "loss":
                frame = input[[idx]]
                if self.weights is not None:
                    frame = frame * self.weights
                loss = self.crit(frame, self.target)
                if self.normalize:
                    loss = ScaleGradients.apply(loss, self.strength)
                self.loss += loss * self.strength / frames
            if self.mode == "capture":
                self.target = input.detach()
        return input


class GramMatrix(nn.Module):
    """(B*C) x (B*C) Gram matrix of a feature batch, or its covariance form (reference loss.py:67-91).  The
    shift/flip arguments exist in the reference's signature but its code always ends with y = x (and its
    `::-1` slices are illegal in torch), so they are accepted and have no effect here either."""

    def forward(self, x, shift_x=0, shift_y=0, shift_t=0, flip_h=False, flip_v=False, use_covariance=False):
        return _GramFn.apply(x, use_covariance)


class StyleLoss(nn.Module):
    """Gram-matrix MSE against blended style targets (reference loss.py:94-186): a per-frame 'static' term
    and, when video_style_factor > 0 (the default, 100), a whole-batch 'dynamic' term."""

    def __init__(self, strength, use_covariance=False, normalize=False, video_style_factor=0, shift_factor=0,
                 flip_factor=0, rotation_factor=0):
        super().__init__()
        self.reset_targets()
        self.strength = strength
        self.blend_weight = None
        self.video_style_factor = video_style_factor
        self.shift_factor = shift_factor
        self.flip_factor = flip_factor
        self.rotation_factor = rotation_factor
        self.gram = GramMatrix()
        self.crit = _MseFn.apply
        self.loss = 0
        self.mode = "none"
        self.use_covariance = use_covariance
        self.normalize = normalize

    def reset_targets(self):
        self.target = torch.Tensor()
        self.video_target = torch.Tensor()
        self.shift_targets_x = []
        self.shift_targets_y = []

    def forward(self, input):
        if self.mode == "none":
            return input
        self.static_loss(input)
        if self.video_style_factor > 0:
            self.dynamic_loss(input)
        return input

    def _accumulate(self, loss):
        if self.normalize:
            loss = ScaleGradients.apply(loss, self.strength)
        return loss

    def static_loss(self, input):
        frames = input.shape[0]
        for idx in range(frames):
            frame = input[idx].unsqueeze(0)
            gram = self.gram(frame, use_covariance=self.use_covariance) / frame.nelement()
            if self.mode == "capture":
                self.loss = 0
                contrib = self.blend_weight * gram.detach() / frames
                self.target = contrib if self.target.nelement() == 0 else self.target + contrib
            if self.mode == "loss":
                self.loss += self._accumulate(self.crit(gram, self.target)) * self.strength / frames

    def dynamic_loss(self, input):
        rows = input.shape[0] * input.shape[1]
        if self.video_target.nelement() != 0 and rows != self.video_target.shape[0]:
            return  # image styles are ignored by the dynamic term when shapes differ
        gram = self.gram(input, use_covariance=self.use_covariance) / input.nelement()
        if self.mode == "capture":
            self.loss = 0
            contrib = self.blend_weight * gram.detach()
            self.video_target = contrib if self.video_target.nelement() == 0 else self.video_target + contrib
        if self.mode == "loss":
            loss = self._accumulate(self.crit(gram, self.video_target))
            self.loss += self.video_style_factor * loss * self.strength / input.shape[0]


class TVLoss(nn.Module):
    """Anisotropic (L1) total variation of the image itself (reference loss.py:224-233); always active."""

    def __init__(self, strength):
        super().__init__()
        self.strength = strength

    def forward(self, input):
        self.loss = self.strength * _TVFn.apply(input)
        return input
