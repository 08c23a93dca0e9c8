"""Fused function evaluation for the image-optimisation loop.

`StyleEngine` compiles the loss network that models.load_model assembled into a flat plan and evaluates
`feval(x) -> loss slots, d loss / d x` (reference optim.py:201-238: zero_grad, net(pastiche), sum of
module losses, backward) as a fixed chain of libmaua_hip kernels on torch's current stream:

  forward   conv+bias+ReLU fused (one MFMA kernel per conv layer), pools, Gram matrices of the style layers;
  losses    one fused MSE kernel per loss gives the scalar and the gradient seed (Gram difference D / feature
            difference) - the 6-7 `.item()` syncs per iteration of the reference are gone, scalars stay on
            the device in `slots`;
  backward  hand-derived (no autograd): the Gram backward D*F accumulates into the feature gradient, every
            conv's backward-data applies the ReLU mask of its own output while it stages the gradient tile,
            pool backward recomputes the argmax, TV adds its sign gradient at the pixels.

Gradient weights follow the reference's ScaleGradients quirk (loss.py:10-20): with
`normalize` the *gradient* of a loss term is weighted by strength^2 while the *reported* loss uses strength.
Activations and gradient buffers are allocated once per image size and reused by every iteration, so an
iteration allocates nothing and can be captured into a hipGraph (`capture=True`).
"""
import contextlib
import os

import torch

import hip
import loss as loss_mod
import models as models_mod
import plan


class UnsupportedNet(RuntimeError):
    """The module sequence contains something the fused plan does not cover (callers fall back to running the
    modules one by one, which still executes on the GPU through the same kernels)."""


def _scale_grad_coeff(incoming, strength):
    # backward of ScaleGradients for a scalar: g / (|g| + 1e-8) * strength^2
    g = float(incoming)
    return g / (abs(g) + 1e-8) * strength * strength


class _Step:
    __slots__ = ("kind", "mod", "relu", "src", "dst", "slot", "k", "stride", "pad", "ceil", "mode")

    def __init__(self, kind, mod=None):
        self.kind, self.mod, self.relu = kind, mod, False
        self.src = self.dst = self.slot = None


class StyleEngine:
    def __init__(self, net, losses):
        self.net, self.losses = net, list(losses)
        self.slot_of = {id(m): i for i, m in enumerate(self.losses)}
        self.steps = self._plan(list(net))
        self.shape = None
        self.graph, self.graph_key = None, None
        # B > 1 frames are either ONE problem (img_vid's window: per-frame terms over B plus the cross-frame dynamic Gram term)
        # or, with `independent`, B separate single-frame problems evaluated together (vid_img's frames without optical
        # flow): every frame gets exactly the B = 1 arithmetic - its own loss slots, its own total - while the convolutions
        # and pools run on the whole batch so that small images still fill the chip.
        self.independent = False
        self.batch_hint = 1  # frames per launch the job plans with: fixes the convolutions' split-K policy (hip.set_split_batch_hint)
        # 3x3 stride-1 convs run on the bf16 matrix cores with a 3-way operand split (fp32 accuracy, conv_x6.hip) unless
        # MAUA_CONV_X6=0 asks for the fp32-MFMA kernels (A/B comparisons)
        mode = plan.get("conv_x6")  # "1" both passes, "fwd" / "bwd" one of them, "0" off
        self.use_x6 = mode != "0"
        self.x6_fwd, self.x6_bwd = mode in ("1", "fwd"), mode in ("1", "bwd")
        self.timer = None  # bench.py: list receiving (tag, algorithmic flops, bytes, start event, end event) per launch

    # -- planning --------------------------------------------------------------------------------------
    def _plan(self, mods):
        steps, act = [], 0  # act = index of the current activation (0 = the image)
        i = 0
        while i < len(mods):
            m = mods[i]
            if isinstance(m, loss_mod.TVLoss):
                if act != 0:
                    raise UnsupportedNet("TVLoss away from the pixels")
                s = _Step("tv", m)
                s.src = act
            elif isinstance(m, loss_mod.ContentLoss):
                s = _Step("content", m)
                s.src = act
            elif isinstance(m, loss_mod.StyleLoss):
                s = _Step("style", m)
                s.src = act
            elif isinstance(m, models_mod.Conv2d):
                s = _Step("conv", m)
                s.k, s.stride, s.pad = m.kernel_size[0], m.stride[0], m.padding[0]
                if m.kernel_size[0] != m.kernel_size[1] or m.stride[0] != m.stride[1] or m.padding[0] != m.padding[1]:
                    raise UnsupportedNet("non-square convolution")
                if i + 1 < len(mods) and isinstance(mods[i + 1], models_mod.ReLU):
                    s.relu = True
                    i += 1
                elif i + 1 < len(mods) and isinstance(mods[i + 1], (loss_mod.ContentLoss, loss_mod.StyleLoss)):
                    # the reference's in-place ReLU makes this layout fail in autograd; nothing to be faithful to
                    raise UnsupportedNet("loss module on a conv output (pre-ReLU)")
                s.src, act = act, act + 1
                s.dst = act
            elif isinstance(m, models_mod.ReLU):
                s = _Step("relu", m)
                s.src = s.dst = act
                # (a stand-alone ReLU rewrites its activation in place.  Loss terms read their source activation AFTER the whole forward
                #  pass - deferred Gram partial launches, the backward pass's D . F - so a loss module in front of it on the same
                #  activation would see the rewritten values; the reference's autograd refuses that layout too)
                if any(t.kind in ("style", "content") and t.src == act for t in steps):
                    raise UnsupportedNet("in-place ReLU over an activation a loss module reads")
            elif isinstance(m, models_mod._Pool2d):
                s = _Step("pool", m)
                s.k, s.stride = models_mod._as_int(m.kernel_size), models_mod._as_int(m.stride)
                s.ceil, s.mode = bool(m.ceil_mode), m.mode
                s.src, act = act, act + 1
                s.dst = act
            else:
                raise UnsupportedNet(f"module {type(m).__name__}")
            if s.kind in ("tv", "content", "style"):
                s.slot = self.slot_of.get(id(m))
                if s.slot is None:
                    raise UnsupportedNet("loss module missing from the losses list")
            steps.append(s)
            i += 1
        return steps

    # -- buffers ---------------------------------------------------------------------------------------
    def _prepare(self, x):
        hip.set_split_batch_hint(self.batch_hint)
        if self.shape == tuple(x.shape) and getattr(self, "prepared_for", None) == (self.independent, self.batch_hint):
            return
        dev = x.device
        B = x.shape[0]
        self.shape = tuple(x.shape)
        self.prepared_for = (self.independent, self.batch_hint)
        self.graph = None
        self.alloc_epoch = getattr(self, "alloc_epoch", 0) + 1  # every buffer below is new: graphs captured over the old ones are void
        self.iter_graphs = {}
        shapes = {0: tuple(x.shape)}
        for s in self.steps:
            n, c, h, w = shapes[s.src]
            if s.kind == "conv":
                oh, ow = hip.conv_out_hw(h, w, s.k, s.stride, s.pad)
                shapes[s.dst] = (n, s.mod.out_channels, oh, ow)
            elif s.kind == "pool":
                shapes[s.dst] = (n, c, hip.pool_out_size(h, s.k, s.stride, s.ceil), hip.pool_out_size(w, s.k, s.stride, s.ceil))
        self.act = {k: (None if k == 0 else torch.empty(v, device=dev)) for k, v in shapes.items()}
        self.gbuf = {k: torch.empty(v, device=dev) for k, v in shapes.items()}
        # 2x2/2 max pools on even planes keep their decisions (one byte per window) for the backward pass
        self.pool_codes = {}
        if plan.on("pool_codes"):
            for s in self.steps:
                if s.kind == "pool" and s.k == 2 and s.stride == 2 and s.mode == "max" and not s.ceil and \
                        hip.pool2x2_codes_supported(*shapes[s.src]):  # (floor mode: the kernels' grids and code planes are (h // 2, w // 2))
                    self.pool_codes[id(s)] = torch.empty(shapes[s.dst], dtype=torch.uint8, device=dev)
        # B > 1 (img_vid's windows of frames): every loss module has several terms - one per frame, plus the cross-frame
        # dynamic Gram term of a StyleLoss - each with its own slot behind the per-module ones
        self.terms = {}
        n_slots = max(len(self.losses), 1)
        if B > 1 and not self.independent:
            for s in self.steps:
                if s.kind in ("style", "content"):
                    count = B + (1 if s.kind == "style" else 0)
                    self.terms[id(s)] = list(range(n_slots, n_slots + count))
                    n_slots += count
        if B > 1 and self.independent:  # one row of slots and one total per frame
            self.slots_all = torch.zeros(B, n_slots, device=dev)
            self.slots = self.slots_all
            self.total = torch.zeros(B, device=dev)
        else:
            self.slots_all = torch.zeros(n_slots, device=dev)
            self.slots = self.slots_all[:max(len(self.losses), 1)]
            self.total = torch.zeros(1, device=dev)
        # Single images and independent frames: the losses leave their partial sums in a ledger (one record per frame and slot)
        # and ONE launch at the end of the evaluation forms every loss value and the totals (hip.loss_ledger_sum)
        self.ledger = None
        self.slots_f64 = None  # tests set this (zeros_like(slots_all, dtype=float64)): the losses before their rounding to fp32
        if (B == 1 or self.independent) and plan.on("loss_ledger"):
            self.ledger = hip.loss_ledger(B, n_slots, dev)
        # Gram / loss chains of a single image: split-K slabs per layer, ONE finishing launch per evaluation (MAUA_GRAM_BATCH=0: a
        # finishing launch per layer, the round-2 form; same bits)
        self.gram_batch_on = plan.on("gram_batch")
        self.gram_partial_batch_on = plan.on("gram_partial_batch")  # (... and one partial launch for the Gram-form layers)
        self._gram_wsp, self._gram_batches = getattr(self, "_gram_wsp", {}), {}
        # conv + ReLU whose only consumer is a 2x2 / 2 max pool with kept decisions, on conv_x3w.hip: the pool runs in the convolution's
        # epilogue (or, where a small grid splits the channel loop, in the pass that adds the slabs) and the full-size activation is
        # never written (nothing reads it: the backward pass routes by the decision bytes).  fused_pool[conv step] = pool step.
        self.fused_pool = {}
        if self.x6_fwd and plan.on("fuse_pool"):
            for s in self.steps:
                if s.kind != "conv" or not s.relu or s.k != 3 or s.stride != 1:
                    continue
                users = [t for t in self.steps if t.src == s.dst and t is not s]
                if len(users) == 1 and users[0].kind == "pool" and id(users[0]) in self.pool_codes and \
                        self._x6_ok(s, s.mod.out_channels) and models_mod.conv3x3_fwd_is_x3w(s.mod, *shapes[s.src][2:]) and \
                        s.mod.out_channels % 8 == 0 and (plan.on("fuse_pool_split") or models_mod.conv3x3_fwd_family(
                            s.mod, B, shapes[s.src][2], shapes[s.src][3])[1] == 1):
                    self.fused_pool[id(s)] = users[0]
        self.pooled_by_conv = {id(v) for v in self.fused_pool.values()}
        for s in self.steps:  # (those activations exist as shapes only: 0.5 GB less at 1024x1024)
            if id(s) in self.fused_pool:
                self.act[s.dst] = torch.empty(shapes[s.dst], device="meta")
        # Such groups on the way back (whether or not the forward pass fused the pool: the decision bytes are the same): where the
        # backward-data pass of the convolution runs on conv_x3w.hip, it stages its input straight from the pooled map's gradient and
        # the decision bytes (hip.conv3x3_x3w_unpool) - the pool's backward launch and the full-size gradient it would write and the
        # convolution read again do not exist.  fused_unpool[conv step] = pool step.
        self.fused_unpool, self.pool_groups = {}, set()  # (pool_groups: the candidates, whether or not the fusion is switched on)
        if self.x6_bwd:
            for s in self.steps:
                if s.kind != "conv" or not s.relu or s.k != 3 or s.stride != 1 or s.pad != 1:
                    continue
                users = [t for t in self.steps if t.src == s.dst and t is not s]
                if len(users) == 1 and users[0].kind == "pool" and id(users[0]) in self.pool_codes and \
                        self._x6_ok(s, s.mod.in_channels) and models_mod.conv3x3_bwd_is_x3w(s.mod, *shapes[s.dst][2:]):
                    self.pool_groups.add(id(s))
                    if plan.on("fuse_unpool"):
                        self.fused_unpool[id(s)] = users[0]
                        self.gbuf[s.dst] = torch.empty(shapes[s.dst], device="meta")
        self.unpooled_by_conv = {id(v) for v in self.fused_unpool.values()}
        # Single images: where a style loss is the only loss on the input activation of a 3x3 layer whose backward-data pass runs
        # on conv_x3w.hip, that pass takes the Gram backward along (D . F as extra one-tap chunks of its K loop) instead of a
        # separate read-modify-write pass over the gradient map: fused_gram[conv step] = (style step, bank of D, 1 / scale).
        self.fused_gram = {}
        max_c = plan.get_int("fuse_gram_max_c")  # pays on the shallow, bandwidth-bound layers (relu4_1: break-even)
        if (B == 1 or self.independent) and self.x6_bwd and max_c > 0:
            relu_out = {s.dst for s in self.steps if s.kind == "conv" and s.relu}
            for s in self.steps:
                if s.kind != "conv" or s.k != 3 or s.stride != 1 or s.pad != 1 or s.src not in relu_out:
                    continue
                on_src = [t for t in self.steps if t.kind in ("style", "content") and t.src == s.src]
                c, h, w = shapes[s.src][1:]
                if len(on_src) == 1 and on_src[0].kind == "style" and not on_src[0].mod.use_covariance and c % 16 == 0 and \
                        c <= max_c and self._x6_ok(s, s.mod.in_channels) and models_mod.conv3x3_bwd_is_x3w(s.mod, *shapes[s.dst][2:]):
                    bank = hip.conv_x3w_dmat_bank(c, dev, B)
                    if bank is not None:
                        self.fused_gram[id(s)] = (on_src[0], bank[0], bank[1])
        self.fused_style = {id(v[0]): v for v in self.fused_gram.values()}
        # The image layer (conv1_1: 3 -> 64 channels on conv_img.hip) can leave the Gram slabs of its own output (relu1_1) next to the
        # activation (hip.conv3x3_image_gram): that layer's partial kernel - a second pass over the largest activation of the network -
        # disappears.  Single images whose Gram chains are batched behind the forward pass.  image_gram[conv step] = style step.
        self.image_gram = {}
        # (not for frames of a planned batch - batch_hint > 1: a frame's bits must not depend on how many others share its launches)
        if B == 1 and self.batch_hint == 1 and self.ledger is not None and self.x6_fwd and plan.on("image_gram") and \
                plan.on("gram_x3") and models_mod._image_kernel_enabled():  # (ready slabs are folded by the fp16x3 route only)
            for s in self.steps:
                if s.kind != "conv" or not s.relu or s.k != 3 or s.stride != 1 or s.mod.in_channels > 3 or s.mod.out_channels != 64 or \
                        not self._x6_ok(s, 64):
                    continue
                on_dst = [t for t in self.steps if t.kind == "style" and t.src == s.dst]
                if len(on_dst) == 1 and not on_dst[0].mod.use_covariance and hip.gram_mse_ledger_supported(64):
                    h, w = shapes[s.src][2:]
                    slabs = hip.conv_image_gram_slabs(h, w, s.pad)
                    if 0 < slabs * 64 * 64 * 4 <= hip.gram_workspace_bytes(64, shapes[s.dst][2] * shapes[s.dst][3]):
                        self.image_gram[id(s)] = (on_dst[0], slabs)
        self.gram, self.dmat, self.mean = {}, {}, {}
        self.gram_d, self.dmat_d, self.mean_d = {}, {}, {}
        ws = hip.reduce_workspace_bytes(max(t.numel() for t in self.gbuf.values()))
        for s in self.steps:
            if s.kind == "style":
                c = shapes[s.src][1]
                hw = shapes[s.src][2] * shapes[s.src][3]
                self.gram[id(s)] = torch.empty(B, c, c, device=dev) if B > 1 else torch.empty(c, c, device=dev)
                self.dmat[id(s)] = torch.empty(B, c, c, device=dev) if B > 1 else torch.empty(c, c, device=dev)
                self.mean[id(s)] = (torch.empty(B, c, device=dev) if B > 1 else torch.empty(c, device=dev)) \
                    if s.mod.use_covariance else None
                ws = max(ws, hip.gram_workspace_bytes(c, hw), 4 * c + 256)
                if B > 1 and not self.independent:  # the dynamic term's (B C) x (B C) Gram over the whole window
                    self.gram_d[id(s)] = torch.empty(B * c, B * c, device=dev)
                    self.dmat_d[id(s)] = torch.empty(B * c, B * c, device=dev)
                    self.mean_d[id(s)] = torch.empty(B * c, device=dev) if s.mod.use_covariance else None
                    ws = max(ws, hip.gram_workspace_bytes(B * c, hw), 4 * B * c + 256)
        for s in self.steps:  # split-K workspaces of the bf16x6 convs (forward and backward-data geometry)
            if s.kind == "conv" and s.stride == 1:
                n, cin, h, w = shapes[s.src]
                _, cout, oh, ow = shapes[s.dst]
                ws = max(ws, hip.conv_workspace_bytes(n, cin, h, w, cout, s.k, 1, s.pad),
                         hip.conv_workspace_bytes(n, cout, oh, ow, cin, s.k, 1, s.k - 1 - s.pad))
                if s.k == 5:
                    ws = max(ws, hip.conv_kxk_x3_workspace_bytes(n, cin, h, w, cout, 5, s.pad),
                             hip.conv_kxk_x3_workspace_bytes(n, cout, oh, ow, cin, 5, 4 - s.pad))
                if s.k == 1 and s.pad == 0:
                    ws = max(ws, hip.conv1x1_x3_workspace_bytes(n, cin, h * w, cout), hip.conv1x1_x3_workspace_bytes(n, cout, h * w, cin))
                if s.k == 3:
                    ws = max(ws, hip.conv_x6_workspace_bytes(n, cin, h, w, cout, s.pad),
                             hip.conv_x6_workspace_bytes(n, cout, oh, ow, cin, 2 - s.pad),
                             hip.conv_x3_workspace_bytes(n, cin, h, w, cout, s.pad),
                             hip.conv_x3_workspace_bytes(n, cout, oh, ow, cin, 2 - s.pad),
                             hip.conv_x3w_workspace_bytes(n, cin, h, w, cout, s.pad),
                             hip.conv_x3w_workspace_bytes(n, cout, oh, ow, cin, 2 - s.pad),
                             hip.conv_x3q_workspace_bytes(n, cin, h, w, cout, s.pad),
                             hip.conv_x3q_workspace_bytes(n, cout, oh, ow, cin, 2 - s.pad),
                             hip.conv_x3p_workspace_bytes(n, cin, h, w, cout, s.pad),
                             hip.conv_x3p_workspace_bytes(n, cout, oh, ow, cin, 2 - s.pad))
        # strided layers whose backward pass runs as a 3x3 convolution over the output sites (models.conv_strided_bwd_as_3x3): the sites' buffer
        self.strided_sites, self.strided_fwd_sites = {}, {}
        for s in self.steps:
            if s.kind == "conv" and s.stride > 1 and self.x6_fwd and models_mod.conv_strided_fwd_is_3x3(s.mod, *shapes[s.src][2:]):
                n, cin, h, w = shapes[s.src]
                qh, qw = shapes[s.dst][2] + 2, shapes[s.dst][3] + 2
                self.strided_fwd_sites[id(s)] = torch.empty(n, s.stride * s.stride * cin, qh, qw, device=dev)
                ws = max(ws, hip.conv_x3w_workspace_bytes(n, s.stride * s.stride * cin, qh, qw, s.mod.out_channels, 0))
        for s in self.steps:
            if s.kind == "conv" and s.stride > 1 and self.x6_bwd and models_mod.conv_strided_bwd_is_3x3(s.mod, *shapes[s.dst][2:]):
                n, cin, _, _ = shapes[s.src]
                _, cout, oh, ow = shapes[s.dst]
                self.strided_sites[id(s)] = torch.empty(n, s.stride * s.stride * cin, oh + 2, ow + 2, device=dev)
                ws = max(ws, hip.conv_x3w_workspace_bytes(n, cout, oh, ow, s.stride * s.stride * cin, 2))
        self.ws = torch.empty(ws, dtype=torch.uint8, device=dev)
        self.x_static = torch.empty(self.shape, device=dev)
        if plan.get("debug_poison") == "1":  # tests: every buffer starts as NaN, so a read-before-write shows up
            for t in list(self.act.values()) + list(self.gbuf.values()) + list(self.gram.values()) + list(self.dmat.values()) + \
                    list(self.gram_d.values()) + list(self.dmat_d.values()):
                if t is not None and not t.is_meta:
                    t.fill_(float("nan"))
            self.ws[:self.ws.numel() // 4 * 4].view(torch.float32)[:] = float("nan")
        # planner field finish_in_launch: split channel loops of the convolutions finished inside the producing launch
        # (hip.conv_arm_workspace; measured neutral, off by default): the arrival counters belong to this workspace
        self.ws_counters = torch.empty(hip.ARRIVAL_COUNTER_BYTES, dtype=torch.uint8, device=dev) if plan.on("finish_in_launch") and ws > 0 else None
        self.ws_armed_epoch = None
        # Independent frames: the per-frame kernels (Gram, losses, Gram backward, optimiser) of different frames share
        # nothing, and most of them are too small to fill the chip or are latency-bound chains - they run on a few side
        # streams so that the GPU overlaps them (fork after the kernel that produced their input, join before the next
        # kernel that reads their output).  Every stream has its own reduction / split-K workspace.
        # A single image: the Gram / loss chain of a style layer only has to be done when the backward pass starts, so it
        # can run on ONE side stream next to the following convolutions (MAUA_STYLE_STREAM=0 / 1 forces it off / on).  Round 3: with the
        # partial kernels of all layers in one launch behind the forward pass the serial form wins up to 1448 x 1448 (724: 263.7 vs
        # 261.8 it/s, 1024: 178.9 vs 177.8, 1448: 80.3 vs 80.9, 2048: 44.07 vs 44.35) - the side stream is for the largest images.
        self.side, self.side_ws, self.ev_main, self.ev_side = [], [], None, []
        aside = plan.get("style_stream")
        self.style_aside = B == 1 and self.ledger is not None and \
            (aside == "1" or (aside == "auto" and x.shape[2] * x.shape[3] >= 1536 * 1536))
        if (B > 1 and self.independent and plan.get_int("side_streams") > 0) or self.style_aside:
            ns = min(B, plan.get_int("side_streams")) if B > 1 else 1
            small = hip.reduce_workspace_bytes(max(t.numel() for t in self.gbuf.values()) // B)
            for s in self.steps:
                if s.kind == "style":
                    c, hw = shapes[s.src][1], shapes[s.src][2] * shapes[s.src][3]
                    small = max(small, hip.gram_workspace_bytes(c, hw), 4 * c + 256)
            self.side = [torch.cuda.Stream(device=dev) for _ in range(ns)]
            self.side_ws = [torch.empty(small, dtype=torch.uint8, device=dev) for _ in range(ns)]
            self.ev_main = torch.cuda.Event()
            self.ev_side = [torch.cuda.Event() for _ in range(ns)]

    # -- one evaluation --------------------------------------------------------------------------------
    def _style_terms(self, s, B):
        """B > 1: ((loss weight, gradient weight) of one frame's static term, the same for the dynamic term or None).
        loss.py:141-181: each frame's MSE enters with strength / B, the cross-frame one with video_style_factor * strength / B;
        with `normalize` the gradient passes ScaleGradients (sign * strength^2) instead."""
        m = s.mod
        st, vsf = m.strength, m.video_style_factor
        static = (st / B, _scale_grad_coeff(st / B, st) if m.normalize else st / B)
        dynamic = None
        if vsf > 0:
            if m.video_target.nelement() == 0:
                raise RuntimeError(f"{m.name}: dynamic style target not captured")
            if m.video_target.shape[0] == B * m.target.shape[0]:  # otherwise the reference skips the term (loss.py:165-166)
                dynamic = (vsf * st / B, _scale_grad_coeff(vsf * st / B, st) if m.normalize else vsf * st / B)
        return static, dynamic

    def _coefficients(self, s):
        """(reported-loss weight, gradient weight) of a loss step for B = 1."""
        m = s.mod
        st = m.strength
        if s.kind == "content":
            return st, (_scale_grad_coeff(st, st) if m.normalize else st)
        vsf = m.video_style_factor
        dyn = vsf > 0
        lw = st + (vsf * st if dyn else 0.0)
        if m.normalize:
            gw = _scale_grad_coeff(st, st) + (_scale_grad_coeff(vsf * st, st) if dyn else 0.0)
        else:
            gw = lw
        return lw, gw

    def _active(self, s, shapes_src):
        m = s.mod
        if s.kind == "tv":
            return True
        if m.mode != "loss":
            return False
        if s.kind == "content":
            if m.target.nelement() == 0:
                return False  # temporal module without a target (loss.py:46-47) or never captured
            if tuple(m.target.shape[1:]) != tuple(shapes_src[1:]):
                return False  # loss.py:44
            if shapes_src[0] > 1 and self.independent:
                if m.weights is not None or m.target.shape[0] != shapes_src[0]:
                    raise UnsupportedNet("independent frames need one unweighted content target per frame")
            elif shapes_src[0] > 1 and (m.weights is not None or m.target.shape[0] != 1):
                raise UnsupportedNet("weighted / multi-frame-target ContentLoss on B > 1 frames runs on the module path")
            if m.weights is not None:
                w = m.weights
                if s.src != 0 or w.dim() != 4 or w.shape[0] != 1 or tuple(w.shape[2:]) != tuple(shapes_src[2:]) \
                        or w.shape[1] not in (1, shapes_src[1]):
                    raise UnsupportedNet("weighted ContentLoss of this shape runs on the module path")
            return True
        if m.target.nelement() == 0:
            raise RuntimeError(f"{m.name}: style target not captured")
        if shapes_src[0] == 1 and m.video_style_factor > 0 and m.video_target.nelement() != 0 \
                and m.video_target.shape != m.target.shape:
            raise UnsupportedNet("dynamic style target of another shape")
        return True

    def _arm(self):
        """This host thread's convolution launches on self.ws finish small splits in the launch (maua_conv_arm_workspace) when the planner
        field finish_in_launch is on.  The counters are zeroed ONCE per workspace (the memset must not be part of a captured iteration:
        every launch leaves them zeroed); later calls only point the thread-local setting back at this engine's workspace."""
        if self.ws_counters is None:
            hip.conv_arm_workspace(None)
            return
        if self.ws_armed_epoch != (self.ws.data_ptr(), self.ws_counters.data_ptr()):
            hip.conv_arm_workspace(self.ws, self.ws_counters)
            self.ws_armed_epoch = (self.ws.data_ptr(), self.ws_counters.data_ptr())
        else:
            hip.conv_arm_workspace(self.ws, self.ws_counters, zero=False)

    def fork(self):
        """Side streams wait for everything enqueued on the current stream so far."""
        if self.side:
            self.ev_main.record()
            for st in self.side:
                st.wait_event(self.ev_main)

    def join(self):
        """The current stream waits for everything enqueued on the side streams."""
        cur = torch.cuda.current_stream()
        for st, ev in zip(self.side, self.ev_side):
            ev.record(st)
            cur.wait_event(ev)

    def _gram_ws(self, s, c, hw, dev):
        """The style layer's own split-K workspace (its slabs wait for the batched finishing launch)."""
        need = hip.gram_workspace_bytes(c, hw)
        t = self._gram_wsp.get(id(s))
        if t is None or t.numel() < need or t.device != dev:
            t = torch.empty(need, dtype=torch.uint8, device=dev)
            self._gram_wsp[id(s)] = t
        return t

    def frame_stream(self, b):
        """(context manager selecting frame b's stream, that stream's workspace): the current stream and the shared workspace
        when side streams are off."""
        if not self.side:
            import contextlib
            return contextlib.nullcontext(), self.ws
        k = b % len(self.side)
        return torch.cuda.stream(self.side[k]), self.side_ws[k]

    def _timed(self, tag, flops, nbytes, fn):
        """Run one kernel launch; when a timer list is attached, bracket it with events on the launch stream."""
        if self.timer is None:
            return fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        self.timer.append((tag, flops, nbytes, e0, e1))
        return out

    @staticmethod
    def _conv_work(s, in_shape, out_shape, backward):
        """Algorithmic FLOPs and bytes of one conv launch (SURVEY.md Appendix B accounting)."""
        _, cin, h, w = in_shape
        _, cout, oh, ow = out_shape
        flops = 2 * cin * cout * s.k * s.k * oh * ow
        wbytes = (cin * cout * s.k * s.k + cout) * 4
        ib, ob = cin * h * w * 4, cout * oh * ow * 4
        nbytes = (ob + ob + wbytes + ib) if backward else (ib + wbytes + ob)
        return flops, nbytes

    def _image_gram_now(self, s, shape):
        """Whether this evaluation's image-layer launch carries the Gram slabs of its style layer: the layer is active (its fold and
        finishing launches then run with the other layers' behind the forward pass, or - largest images - on the side stream)."""
        st, _ = self.image_gram[id(s)]
        return self.gram_batch_on and self.gram_partial_batch_on and self._active(st, self.act[s.dst].shape)

    def image_gram_slabs(self, style_step):
        return next(n for st, n in self.image_gram.values() if st is style_step)

    def _x6_ok(self, s, produced_channels):
        """bf16x6 kernel: 3x3, stride 1, and enough produced channels to fill its 64-channel tile."""
        return self.use_x6 and s.k == 3 and s.stride == 1 and s.pad <= 2 and produced_channels >= plan.get_int("split_min_produced")

    def _run(self, x):
        hip.set_split_batch_hint(self.batch_hint)  # host-side setting read when a launch picks its split: nothing is enqueued
        self._arm()
        a, g = self.act, self.gbuf
        a[0] = x
        hip.fill_(self.slots_all, 0.0)
        forked = False
        batch = []
        emitted = set()  # style steps whose Gram slabs came out of the image layer's launch
        # ---------------- forward
        for s in self.steps:
            if s.kind == "conv":
                fl, nb = self._conv_work(s, a[s.src].shape, a[s.dst].shape, False)
                if id(s) in self.fused_pool:
                    ps = self.fused_pool[id(s)]
                    self._timed("conv3x3_split_fwd", fl, nb, lambda: models_mod.conv3x3_relu_pool(
                        a[s.src], s.mod, a[ps.dst], self.pool_codes[id(ps)], workspace=self.ws))
                elif id(s) in self.image_gram and self._image_gram_now(s, a[s.src].shape):
                    st, _ = self.image_gram[id(s)]
                    gws = self._gram_ws(st, 64, a[s.dst].shape[2] * a[s.dst].shape[3], x.device)
                    models_mod._route("conv_image", a[s.src], s.mod.out_channels, s.pad, False, "27 (channel, tap) pairs as K, bf16x6", relu=True, gram_slabs=True)
                    self._timed("conv3x3_split_fwd", fl, nb, lambda: hip.conv3x3_image_gram(a[s.src], s.mod.bank_image(), s.pad, a[s.dst], gws))
                    emitted.add(id(st))
                elif self.x6_fwd and self._x6_ok(s, s.mod.out_channels):
                    self._timed("conv3x3_split_fwd", fl, nb, lambda: models_mod.conv3x3_mfma(
                        a[s.src], s.mod, False, out=a[s.dst], relu=s.relu, workspace=self.ws))
                elif self.x6_fwd and models_mod.conv1x1_is_mfma(s.mod, False):
                    self._timed("conv_1x1_fwd", fl, nb, lambda: models_mod.conv1x1_mfma(
                        a[s.src], s.mod, False, out=a[s.dst], relu=s.relu, workspace=self.ws))
                elif self.x6_fwd and models_mod.conv5x5_is_mfma(s.mod, False):
                    self._timed("conv_5x5_fwd", fl, nb, lambda: models_mod.conv5x5_mfma(
                        a[s.src], s.mod, False, out=a[s.dst], relu=s.relu, workspace=self.ws))
                elif id(s) in self.strided_fwd_sites:
                    self._timed("conv_other_fwd", fl, nb, lambda: models_mod.conv_strided_fwd_as_3x3(
                        a[s.src], s.mod, a[s.dst], s.relu, self.strided_fwd_sites[id(s)], workspace=self.ws))
                else:
                    wf, _ = s.mod.banks()
                    models_mod._route("conv2d_fwd", a[s.src], s.mod.out_channels, s.pad, False, "direct / fp32 MFMA", relu=s.relu)
                    self._timed("conv_other_fwd", fl, nb, lambda: hip.conv2d_fwd(
                        a[s.src], wf, s.mod.bias_device(), s.k, s.stride, s.pad, s.relu, out=a[s.dst], workspace=self.ws))
            elif s.kind == "relu":
                hip.relu_(a[s.src])
            elif s.kind == "pool":
                if id(s) in self.pooled_by_conv:
                    pass  # done in the epilogue of the convolution in front of it
                elif id(s) in self.pool_codes:
                    hip.pool2x2_fwd_codes(a[s.src], a[s.dst], self.pool_codes[id(s)])
                else:
                    hip.pool2d_fwd(a[s.src], s.k, s.stride, s.ceil, s.mode, out=a[s.dst])
            elif s.kind == "style" and self.independent and a[s.src].shape[0] > 1 and self._active(s, a[s.src].shape):
                f = a[s.src]
                c, n = f.shape[1], f[0].nelement()
                lw, gw = self._coefficients(s)  # the single-frame weights: static + dynamic term of the same Gram
                fs = self.fused_style.get(id(s))
                self.fork()  # joined before the backward pass starts: nothing in the rest of the forward pass needs these
                for b in range(f.shape[0]):
                    mean_b = self.mean[id(s)][b] if s.mod.use_covariance else None
                    ctx, wsb = self.frame_stream(b)
                    with ctx:
                        if self.ledger is not None and hip.gram_mse_ledger_supported(c):
                            self._timed("gram_fwd", 2 * c * c * (n // c), n * 4 + c * c * 4, lambda: hip.gram_fwd_mse_ledger(
                                f[b:b + 1], 1.0 / n, s.mod.use_covariance, self.gram[id(s)][b], mean_b, s.mod.target,
                                self.dmat[id(s)][b], lw / (c * c), gw * 4.0 / (c * c) / n, self.ledger[b], s.slot, workspace=wsb))
                            if fs is not None:
                                hip.conv_pack_dmat_x3w(self.dmat[id(s)][b], fs[1][b], fs[2][b:b + 1])
                            continue
                        self._timed("gram_fwd", 2 * c * c * (n // c), n * 4 + c * c * 4, lambda: hip.gram_fwd(
                            f[b:b + 1], 1.0 / n, s.mod.use_covariance, out=self.gram[id(s)][b], mean_out=mean_b, workspace=wsb))
                        hip.mse_fwd_bwd(self.gram[id(s)][b], s.mod.target, self.dmat[id(s)][b], lw / (c * c),
                                        gw * 4.0 / (c * c) / n, False, self.slots_all[b, s.slot:s.slot + 1], workspace=wsb)
                        if fs is not None:
                            hip.conv_pack_dmat_x3w(self.dmat[id(s)][b], fs[1][b], fs[2][b:b + 1])
            elif s.kind == "style" and self._active(s, a[s.src].shape) and a[s.src].shape[0] > 1:
                f = a[s.src]
                B, c, n = f.shape[0], f.shape[1], f[0].nelement()
                (lw, gw), dynamic = self._style_terms(s, B)
                slots = self.terms[id(s)]
                cov = s.mod.use_covariance
                if dynamic is not None:
                    # One (B C) x (B C) Gram over the window (rows = (frame, channel) pairs of the contiguous activation): its
                    # diagonal C x C blocks are the per-frame Grams (up to the normalisation, 1 / (B n) instead of 1 / n), so the
                    # static terms cost no extra pass over the feature maps - forward or backward.
                    lwd, gwd = dynamic
                    bc, nall = B * c, B * n
                    gd, dd = self.gram_d[id(s)], self.dmat_d[id(s)]
                    self._timed("gram_fwd", 2 * bc * bc * (n // c), nall * 4 + bc * bc * 4, lambda: hip.gram_fwd(
                        f.view(1, bc, f.shape[2], f.shape[3]), 1.0 / nall, cov, out=gd, mean_out=self.mean_d[id(s)],
                        workspace=self.ws))
                    hip.mse_fwd_bwd(gd, s.mod.video_target, dd, lwd / (bc * bc), gwd * 4.0 / (bc * bc) / nall, False,
                                    self.slots_all[slots[B]:slots[B] + 1], workspace=self.ws)
                    for b in range(B):
                        blk = slice(b * c, (b + 1) * c)
                        torch.mul(gd[blk, blk], float(B), out=self.gram[id(s)][b])
                        hip.mse_fwd_bwd(self.gram[id(s)][b], s.mod.target, self.dmat[id(s)][b], lw / (c * c), gw * 4.0 / (c * c) / n,
                                        False, self.slots_all[slots[b]:slots[b] + 1], workspace=self.ws)
                        dd[blk, blk] += self.dmat[id(s)][b]  # the frame's static gradient matrix joins the diagonal block
                else:
                    for b in range(B):  # static terms only: one C x C Gram per frame against the shared target
                        mean_b = self.mean[id(s)][b] if cov else None
                        self._timed("gram_fwd", 2 * c * c * (n // c), n * 4 + c * c * 4, lambda: hip.gram_fwd(
                            f[b:b + 1], 1.0 / n, cov, out=self.gram[id(s)][b], mean_out=mean_b, workspace=self.ws))
                        hip.mse_fwd_bwd(self.gram[id(s)][b], s.mod.target, self.dmat[id(s)][b], lw / (c * c),
                                        gw * 4.0 / (c * c) / n, False, self.slots_all[slots[b]:slots[b] + 1], workspace=self.ws)
            elif s.kind == "style" and self._active(s, a[s.src].shape):
                f = a[s.src]
                c, n = f.shape[1], f[0].nelement()
                lw, gw = self._coefficients(s)
                # loss = lw * mean((G-T)^2); D = gw * (2/C^2) * (2/n) * (G - T)   (dG/dF = (D + D^T) F / n, D symmetric)
                if self.ledger is not None and hip.gram_mse_ledger_supported(c):
                    if self.style_aside and self.timer is None:
                        self.fork()  # (a later fork only adds the dependency on the layers in between)
                        forked = True
                        ctx, wsb = self.frame_stream(0)
                    else:
                        ctx, wsb = contextlib.nullcontext(), self.ws
                    with ctx:
                        if self.gram_batch_on and not forked:
                            # (large images keep the per-layer form: their chains run on the side stream beside the convolutions, finishing
                            #  launches included - batching those would expose them at the join in front of the backward pass)
                            # only the split-K slabs now (into the layer's own workspace); ONE finishing launch for all style layers
                            # of the evaluation follows the forward pass (their D matrices are not needed before the backward pass)
                            gws = self._gram_ws(s, c, n // c, f.device)
                            # the slabs wait too - the partial kernels of all style layers (and, covariance form, their row means before) go
                            # out together behind the forward pass (hip.GramFinishBatch.run_partial: each alone is 10-25 us of latency on a
                            # part of the chip)
                            later = self.gram_partial_batch_on
                            assert later or id(s) not in emitted
                            if not later:
                                self._timed("gram_fwd", 2 * c * c * (n // c), n * 4 + c * c * 4, lambda: hip.gram_partial(
                                    f, s.mod.use_covariance, self.mean[id(s)], gws))
                            # (covariance form: its row means come out of the batched call too, ahead of the centred products)
                            batch.append(dict(step=s, workspace=gws, gram=self.gram[id(s)], target=s.mod.target, dmat=self.dmat[id(s)], c=c,
                                              hw=n // c, scale=1.0 / n, loss_scale=lw / (c * c), grad_scale=gw * 4.0 / (c * c) / n,
                                              ledger=self.ledger[0], slot=s.slot, f=f if later else None,
                                              mean=self.mean[id(s)] if later and s.mod.use_covariance else None,
                                              slabs=self.image_gram_slabs(s) if id(s) in emitted else 0))
                            continue
                        if id(s) in emitted:  # (side stream: fold and finish the slabs the image layer's launch left)
                            l = dict(step=s, workspace=self._gram_ws(s, c, n // c, f.device), gram=self.gram[id(s)], target=s.mod.target,
                                     dmat=self.dmat[id(s)], c=c, hw=n // c, scale=1.0 / n, loss_scale=lw / (c * c), grad_scale=gw * 4.0 / (c * c) / n,
                                     ledger=self.ledger[0], slot=s.slot, f=None, slabs=self.image_gram_slabs(s))
                            key = (id(s), l["target"].data_ptr(), l["workspace"].data_ptr(), l["gram"].data_ptr(), l["dmat"].data_ptr(),
                                   l["ledger"].data_ptr(), l["slot"], l["loss_scale"], l["grad_scale"], l["slabs"])
                            fin = self._gram_batches.get(("img", id(s)))
                            if fin is None or fin[0] != key:
                                fin = (key, hip.GramFinishBatch([l]))
                                self._gram_batches[("img", id(s))] = fin
                            fin[1].run_partial()
                            fin[1].run()
                        else:
                            self._timed("gram_fwd", 2 * c * c * (n // c), n * 4 + c * c * 4, lambda: hip.gram_fwd_mse_ledger(
                                f, 1.0 / n, s.mod.use_covariance, self.gram[id(s)], self.mean[id(s)], s.mod.target, self.dmat[id(s)],
                                lw / (c * c), gw * 4.0 / (c * c) / n, self.ledger[0], s.slot, workspace=wsb))
                        if id(s) in self.fused_style:
                            hip.conv_pack_dmat_x3w(self.dmat[id(s)], self.fused_style[id(s)][1][0], self.fused_style[id(s)][2])
                    continue
                self._timed("gram_fwd", 2 * c * c * (n // c), n * 4 + c * c * 4, lambda: hip.gram_fwd(
                    f, 1.0 / n, s.mod.use_covariance, out=self.gram[id(s)], mean_out=self.mean[id(s)], workspace=self.ws))
                hip.mse_fwd_bwd(self.gram[id(s)], s.mod.target, self.dmat[id(s)], lw / (c * c), gw * 4.0 / (c * c) / n,
                                False, self.slots[s.slot:s.slot + 1], workspace=self.ws)
                if id(s) in self.fused_style:
                    hip.conv_pack_dmat_x3w(self.dmat[id(s)], self.fused_style[id(s)][1][0], self.fused_style[id(s)][2])
        if batch:  # the finishing pass of every style layer's Gram / loss chain in one launch (groups of eight), then the D banks
            for k0 in range(0, len(batch), 8):
                grp = batch[k0:k0 + 8]
                key = tuple((id(l["step"]), l["target"].data_ptr(), l["workspace"].data_ptr(), l["gram"].data_ptr(), l["dmat"].data_ptr(),
                             l["ledger"].data_ptr(), l["slot"], l["loss_scale"], l["grad_scale"],
                             None if l["f"] is None else l["f"].data_ptr(), l["slabs"]) for l in grp)
                later = self.gram_partial_batch_on  # (then every entry waits with its feature map; otherwise none does)
                fin = self._gram_batches.get(k0)
                if fin is None or fin[0] != key:
                    fin = (key, hip.GramFinishBatch(grp))
                    self._gram_batches[k0] = fin
                if later:
                    self._timed("gram_fwd", sum(2 * l["c"] * l["c"] * l["hw"] for l in grp), sum(l["c"] * l["hw"] * 4 for l in grp), fin[1].run_partial)
                self._timed("gram_fwd", 0, sum(l["c"] * l["c"] * 8 for l in grp), fin[1].run)
            packs = [(l["dmat"], self.fused_style[id(l["step"])][1][0], self.fused_style[id(l["step"])][2]) for l in batch
                     if id(l["step"]) in self.fused_style]
            if not plan.on("dmat_pack_batch"):
                for d, b, i in packs:
                    hip.conv_pack_dmat_x3w(d, b, i)
                packs = []
            for k0 in range(0, len(packs), 4):  # the one-tap banks of the fused layers' D matrices, one launch
                grp = packs[k0:k0 + 4]
                key = tuple((d.data_ptr(), b.data_ptr(), i.data_ptr()) for d, b, i in grp)
                pk = self._gram_batches.get(("pack", k0))
                if pk is None or pk[0] != key:
                    pk = (key, hip.DmatPackBatch(grp))
                    self._gram_batches[("pack", k0)] = pk
                pk[1].run()
        # ---------------- backward
        # Gradient buffers of fused conv+ReLU activations are kept PRE-MASKED: the last kernel that writes g[k] (the
        # backward of the consumer, or the last loss term attached to k) zeroes it where a[k] <= 0, so no backward-data
        # pass has to apply threshold_backward while it stages its input.
        final_writer = {}
        for s in self.steps:  # forward order == reverse of execution order: the first hit per activation wins
            if s.kind in ("conv", "pool") or (s.kind in ("style", "content") and self._active(s, a[s.src].shape)):
                final_writer.setdefault(s.src, s)
        relu_acts = {s.dst for s in self.steps if s.kind == "conv" and s.relu}

        def premask(s):
            return final_writer.get(s.src) is s and s.src in relu_acts

        fused_done = set()
        cur = None  # activation index whose gradient buffer currently holds d loss / d act
        indep = self.independent and x.shape[0] > 1
        if indep or forked:
            self.join()  # the per-frame Gram / loss kernels of the forward pass
        for s in reversed(self.steps):
            if indep and s.kind in ("style", "content", "tv"):
                if s.kind != "tv" and not self._active(s, a[s.src].shape):
                    continue
                if id(s) in fused_done:
                    continue  # its Gram backward went along with the convolution's backward pass
                f = a[s.src]
                acc = cur == s.src
                last_step = s.kind == "tv"  # module 0: the totals can be formed on the frame's stream right behind it
                self.fork()
                for b in range(f.shape[0]):
                    slot = self.slots_all[b, s.slot:s.slot + 1]
                    ctx, wsb = self.frame_stream(b)
                    with ctx:
                        if s.kind == "style":
                            c, n = f.shape[1], f[0].nelement()
                            self._timed("gram_bwd", 2 * c * c * (n // c), n * 4 * 3 + c * c * 4, lambda: hip.gram_bwd(
                                self.dmat[id(s)][b], f[b], self.mean[id(s)][b] if s.mod.use_covariance else None, g[s.src][b],
                                acc, workspace=wsb, relu_mask=f[b] if premask(s) else None))
                        elif s.kind == "content":
                            lw, gw = self._coefficients(s)
                            n = f[0].nelement()
                            if self.ledger is not None:
                                hip.mse_fwd_bwd_ledger(f[b], s.mod.target[b], g[s.src][b], lw / n, gw * 2.0 / n, acc, self.ledger[b],
                                                       s.slot, mask_grad_by_x=premask(s))
                            else:
                                hip.mse_fwd_bwd(f[b], s.mod.target[b], g[s.src][b], lw / n, gw * 2.0 / n, acc, slot, workspace=wsb,
                                                mask_grad_by_x=premask(s))
                        elif self.ledger is not None:
                            hip.tv_fwd_bwd_ledger(f[b:b + 1], g[0][b:b + 1], s.mod.strength, acc, self.ledger[b], s.slot)
                        else:
                            hip.tv_fwd_bwd(f[b:b + 1], g[0][b:b + 1], s.mod.strength, acc, slot, workspace=wsb)
                self.join()
                cur = s.src
            elif s.kind == "style" and a[s.src].shape[0] > 1:
                if self._active(s, a[s.src].shape):
                    f = a[s.src]
                    B, c, n = f.shape[0], f.shape[1], f[0].nelement()
                    _, dynamic = self._style_terms(s, B)
                    acc = cur == s.src
                    cov = s.mod.use_covariance
                    if dynamic is not None:  # one pass: D = dynamic matrix + per-frame static matrices on its diagonal blocks
                        bc = B * c
                        self._timed("gram_bwd", 2 * bc * bc * (n // c), B * n * 4 * 3 + bc * bc * 4, lambda: hip.gram_bwd(
                            self.dmat_d[id(s)], f, self.mean_d[id(s)], g[s.src], acc, workspace=self.ws,
                            relu_mask=f if premask(s) else None))
                    else:
                        for b in range(B):
                            self._timed("gram_bwd", 2 * c * c * (n // c), n * 4 * 3 + c * c * 4, lambda: hip.gram_bwd(
                                self.dmat[id(s)][b], f[b], self.mean[id(s)][b] if cov else None, g[s.src][b], acc,
                                workspace=self.ws, relu_mask=f[b] if premask(s) else None))
                    cur = s.src
            elif s.kind == "content" and a[s.src].shape[0] > 1:
                if self._active(s, a[s.src].shape):
                    # loss.py:48-58 loops over the frames: MSE of each frame against the single-frame target, entering with
                    # strength / B (through ScaleGradients when normalising: sign * strength^2, no 1 / B)
                    x = a[s.src]
                    B, n = x.shape[0], x[0].nelement()
                    st = s.mod.strength
                    lw = st / B
                    gw = _scale_grad_coeff(st / B, st) if s.mod.normalize else st / B
                    slots = self.terms[id(s)]
                    for b in range(B):
                        hip.mse_fwd_bwd(x[b], s.mod.target[0], g[s.src][b], lw / n, gw * 2.0 / n, cur == s.src,
                                        self.slots_all[slots[b]:slots[b] + 1], workspace=self.ws, mask_grad_by_x=premask(s))
                    cur = s.src
            elif s.kind == "style":
                if id(s) in fused_done:
                    continue  # its Gram backward went along with the convolution's backward pass
                if self._active(s, a[s.src].shape):
                    f = a[s.src]
                    c, n = f.shape[1], f[0].nelement()
                    acc = cur == s.src
                    rm = f if premask(s) else None
                    self._timed("gram_bwd", 2 * c * c * (n // c), n * 4 * 3 + c * c * 4, lambda: hip.gram_bwd(
                        self.dmat[id(s)], f, self.mean[id(s)], g[s.src], acc, workspace=self.ws, relu_mask=rm))
                    cur = s.src
            elif s.kind == "content":
                if self._active(s, a[s.src].shape):
                    lw, gw = self._coefficients(s)
                    n = a[s.src].nelement()
                    if s.mod.weights is not None:  # temporal loss on the pixels: MSE(x * w, target), loss.py:52-56
                        hip.mse_weighted_fwd_bwd(a[s.src], s.mod.weights, s.mod.target, g[s.src], lw / n, gw * 2.0 / n,
                                                 cur == s.src, self.slots[s.slot:s.slot + 1], workspace=self.ws)
                    elif self.ledger is not None:
                        hip.mse_fwd_bwd_ledger(a[s.src], s.mod.target, g[s.src], lw / n, gw * 2.0 / n, cur == s.src, self.ledger[0],
                                               s.slot, mask_grad_by_x=premask(s))
                    else:
                        hip.mse_fwd_bwd(a[s.src], s.mod.target, g[s.src], lw / n, gw * 2.0 / n, cur == s.src,
                                        self.slots[s.slot:s.slot + 1], workspace=self.ws, mask_grad_by_x=premask(s))
                    cur = s.src
            elif s.kind == "tv":
                if self.ledger is not None:
                    hip.tv_fwd_bwd_ledger(a[0], g[0], s.mod.strength, cur == 0, self.ledger[0], s.slot)
                else:
                    hip.tv_fwd_bwd(a[0], g[0], s.mod.strength, cur == 0, self.slots[s.slot:s.slot + 1], workspace=self.ws)
                cur = 0
            elif cur is None:
                continue  # nothing flows through layers behind the last loss
            elif s.kind == "conv":
                assert cur == s.dst
                fl, nb = self._conv_work(s, a[s.src].shape, a[s.dst].shape, True)
                im = a[s.src] if premask(s) else None
                fg = self.fused_gram.get(id(s))
                with_gram = fg is not None and self._active(fg[0], a[s.src].shape) and premask(fg[0])
                up = self.fused_unpool.get(id(s))
                if up is not None:  # from the pooled map's gradient and the pool's decisions (the pool step below was skipped)
                    c, n = a[s.src].shape[1], a[s.src][0].nelement()
                    self._timed("conv3x3_split_bwd", fl + (2 * c * c * (n // c) if with_gram else 0), nb, lambda: models_mod.conv3x3_bwd_from_pooled(
                        g[up.dst], self.pool_codes[id(up)], premask(up), s.mod, g[s.src], out_relu_mask=a[s.src] if with_gram else im,
                        dmat_bank=fg[1] if with_gram else None, dmat_inv_scale=fg[2] if with_gram else None, workspace=self.ws))
                    if with_gram:
                        fused_done.add(id(fg[0]))
                elif with_gram:
                    c, n = a[s.src].shape[1], a[s.src][0].nelement()
                    self._timed("conv3x3_split_bwd", fl + 2 * c * c * (n // c), nb, lambda: models_mod.conv3x3_bwd_with_gram(
                        g[s.dst], s.mod, a[s.src], fg[1], fg[2], out=g[s.src], workspace=self.ws))
                    fused_done.add(id(fg[0]))
                elif self.x6_bwd and self._x6_ok(s, s.mod.in_channels):
                    self._timed("conv3x3_split_bwd", fl, nb, lambda: models_mod.conv3x3_mfma(
                        g[s.dst], s.mod, True, out=g[s.src], out_relu_mask=im, workspace=self.ws, pool_group=id(s) in self.pool_groups))
                elif self.x6_bwd and models_mod.conv1x1_is_mfma(s.mod, True):
                    self._timed("conv_1x1_bwd", fl, nb, lambda: models_mod.conv1x1_mfma(
                        g[s.dst], s.mod, True, out=g[s.src], out_relu_mask=im, workspace=self.ws))
                elif self.x6_bwd and models_mod.conv5x5_is_mfma(s.mod, True):
                    self._timed("conv_5x5_bwd", fl, nb, lambda: models_mod.conv5x5_mfma(
                        g[s.dst], s.mod, True, out=g[s.src], out_relu_mask=im, workspace=self.ws))
                elif im is None and models_mod.conv_few_is_mfma(s.mod, *g[s.dst].shape[0:1], *g[s.dst].shape[2:]):
                    self._timed("conv_other_bwd", fl, nb, lambda: models_mod.conv_few_mfma(g[s.dst], s.mod, g[s.src]))
                elif id(s) in self.strided_sites and im is None:
                    self._timed("conv_other_bwd", fl, nb, lambda: models_mod.conv_strided_bwd_as_3x3(
                        g[s.dst], s.mod, g[s.src], workspace=self.ws, sites=self.strided_sites[id(s)]))
                else:
                    _, wb = s.mod.banks()
                    models_mod._route("conv3x3_few_out" if s.k == 3 and s.stride == 1 and s.mod.in_channels <= 4 and s.mod.out_channels >= 16 else "conv2d_bwd_data",
                                      g[s.dst], s.mod.in_channels, s.k - 1 - s.pad, True, "direct / fp32 MFMA", mask=im is not None)
                    self._timed("conv_other_bwd", fl, nb, lambda: hip.conv2d_bwd_data(
                        g[s.dst], None, wb, s.mod.weight.detach(), a[s.src].shape, s.k, s.stride, s.pad, out=g[s.src],
                        in_relu_mask=im, workspace=self.ws))
                cur = s.src
            elif s.kind == "relu":
                hip.relu_bwd(g[s.src], a[s.src], out=g[s.src])
            elif s.kind == "pool":
                assert cur == s.dst
                if id(s) in self.unpooled_by_conv:
                    pass  # its backward pass happens in the staging of the convolution's (fused_unpool)
                elif id(s) in self.pool_codes:
                    hip.pool2x2_bwd_codes(g[s.dst], self.pool_codes[id(s)], g[s.src], premask(s))
                else:
                    hip.pool2d_bwd(g[s.dst], a[s.src], s.k, s.stride, s.ceil, s.mode, out=g[s.src],
                                   relu_mask_by_x=premask(s))
                cur = s.src
        if cur != 0:
            hip.fill_(g[0], 0.0)
        if self.ledger is not None:  # (B == 1 or independent frames: no per-module terms to fold afterwards)
            hip.loss_ledger_sum(self.ledger, self.slots_all, self.total, self.slots_f64)
            return
        if indep:
            for b in range(x.shape[0]):
                hip.sum_small(self.slots_all[b], self.total[b:b + 1])
            return
        hip.sum_small(self.slots_all, self.total)
        for s in self.steps:  # B > 1: a module's reported loss is the sum of its terms (after the total, which has them once)
            if id(s) in self.terms:
                t = self.terms[id(s)]
                self.slots_all[s.slot:s.slot + 1] = self.slots_all[t[0]:t[-1] + 1].sum(0, keepdim=True)

    def capture_content(self, x):
        """optim.set_content_targets on the fused plan: one forward pass (convs and pools only, preallocated buffers) and
        a copy of every content layer's activation into its module's `.target` (ContentLoss 'capture', loss.py:61-62).
        Same kernels as the module-by-module pass, without its per-module launches and allocations."""
        self._prepare(x)
        a = self.act
        a[0] = x
        want = [s for s in self.steps if s.kind == "content" and "temporal" not in getattr(s.mod, "name", "")]
        last = max((self.steps.index(s) for s in want), default=-1)
        for s in self.steps[:last + 1]:
            if s.kind == "conv":
                if id(s) in self.fused_pool:
                    ps = self.fused_pool[id(s)]
                    models_mod.conv3x3_relu_pool(a[s.src], s.mod, a[ps.dst], self.pool_codes[id(ps)], workspace=self.ws)
                elif self.x6_fwd and self._x6_ok(s, s.mod.out_channels):
                    models_mod.conv3x3_mfma(a[s.src], s.mod, False, out=a[s.dst], relu=s.relu, workspace=self.ws)
                elif self.x6_fwd and models_mod.conv1x1_is_mfma(s.mod, False):
                    models_mod.conv1x1_mfma(a[s.src], s.mod, False, out=a[s.dst], relu=s.relu, workspace=self.ws)
                elif self.x6_fwd and models_mod.conv5x5_is_mfma(s.mod, False):
                    models_mod.conv5x5_mfma(a[s.src], s.mod, False, out=a[s.dst], relu=s.relu, workspace=self.ws)
                elif id(s) in self.strided_fwd_sites:
                    models_mod.conv_strided_fwd_as_3x3(a[s.src], s.mod, a[s.dst], s.relu, self.strided_fwd_sites[id(s)], workspace=self.ws)
                else:
                    wf, _ = s.mod.banks()
                    hip.conv2d_fwd(a[s.src], wf, s.mod.bias_device(), s.k, s.stride, s.pad, s.relu, out=a[s.dst],
                                   workspace=self.ws)
            elif s.kind == "relu":
                hip.relu_(a[s.src])
            elif s.kind == "pool":
                if id(s) not in self.pooled_by_conv:
                    hip.pool2d_fwd(a[s.src], s.k, s.stride, s.ceil, s.mode, out=a[s.dst])
        for s in want:
            t = getattr(s.mod, "target", None)
            if torch.is_tensor(t) and t.shape == a[s.src].shape and t.device == a[s.src].device and t.dtype == a[s.src].dtype:
                t.copy_(a[s.src])  # in place: a captured iteration (optim.PixelOptimizer's graph bundles) that reads this target stays valid
            else:
                s.mod.target = a[s.src].detach().clone()

    def _graph_key(self):
        """Everything a captured evaluation bakes in besides the image: the addresses and shapes of every target / weight
        tensor, the Python-side decisions of `_active` and the loss coefficients.  A graph is replayed only while this is
        unchanged (the same network is reused by vid_img per frame and by img_vid per window, each call installing new
        targets); a tensor rewritten IN PLACE keeps its address, so the graph stays valid and reads the new values."""
        def sig(t):
            return None if t is None or not torch.is_tensor(t) else (t.data_ptr(), tuple(t.shape))
        key = []
        for s in self.steps:
            m = s.mod
            if s.kind in ("tv", "content", "style"):
                key.append((s.kind, id(m), getattr(m, "mode", None), float(m.strength), bool(getattr(m, "normalize", False)),
                            sig(getattr(m, "target", None)), sig(getattr(m, "video_target", None)), sig(getattr(m, "weights", None)),
                            float(getattr(m, "video_style_factor", 0.0)), bool(getattr(m, "use_covariance", False))))
            elif s.kind == "conv":
                key.append((id(m), m.weight.data_ptr(), m.weight._version))
        key.append(("hint", self.batch_hint, self.independent))
        return tuple(key)

    def feval(self, x, capture=False):
        """Evaluate at `x` (B,3,H,W fp32 on the GPU; B > 1 = a window of frames with per-frame and cross-frame style terms).  Returns (per-module loss slots in `losses` order, total
        loss, gradient) - device tensors owned by the engine, overwritten by the next call; no host sync."""
        self._prepare(x)
        if not capture:
            self._run(x)
            return self.slots, self.total, self.gbuf[0]
        key = self._graph_key()
        if self.graph is None or self.graph_key != key:
            self.graph, self.graph_key = None, key
            self.x_static.copy_(x)
            self._run(self.x_static)  # warm-up outside capture (filter banks, lazy init)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._run(self.x_static)
        self.x_static.copy_(x)
        self.graph.replay()
        return self.slots, self.total, self.gbuf[0]

    def describe_routes(self, x, independent=None, batch_hint=None):
        """Which kernel every convolution launch of one evaluation at `x` takes, as planned and launched (one eager evaluation with
        models.ROUTE_LOG recording): a list of records in launch order - kernel family, pass, channels, plane, tile, K splits, and what
        rides along (mask, pool, unpool, gram).  The conv1_1 backward (64 -> 3: conv3x3_few_out) and NIN's non-3x3 layers are listed by
        the generic entry points' names.  bench.py prints it as `routes`.
        This IS an evaluation: it overwrites the loss slots, gradient buffers and ledger of the engine (call it when the last evaluation's
        results are no longer needed); `independent` / `batch_hint`: what a frame-batch optimiser would set on the engine (they change K
        splits and routes), restored afterwards.  Not re-entrant (one module-level log)."""
        assert models_mod.ROUTE_LOG is None, "describe_routes: a recording is already running"
        keep = (self.independent, self.batch_hint)
        if independent is not None:
            self.independent = bool(independent)
        if batch_hint is not None:
            self.batch_hint = int(batch_hint)
        models_mod.ROUTE_LOG = []
        try:
            self._prepare(x)  # (the plan depends on both settings; the next evaluation re-plans if they were changed here)
            self._run(x)
            torch.cuda.synchronize()
            log = models_mod.ROUTE_LOG
        finally:
            models_mod.ROUTE_LOG = None
            self.independent, self.batch_hint = keep
        return log

    def drop_graphs(self):
        """Release the captured iteration bundles (graph memory pool, image and L-BFGS history of each): for a caller that is done with
        this image size and keeps the network."""
        self.iter_graphs = {}
        self.graph, self.graph_key = None, None

    def saved_bytes(self):
        """Bytes held by activations + gradient buffers for the current shape."""
        tot = 0
        for d in (self.act, self.gbuf):
            for t in d.values():
                if t is not None and not t.is_meta:
                    tot += t.numel() * 4
        return tot
