"""Small helpers with the reference's names (reference utils.py).

`match_histogram` is the colour transfer run before/after every scale (reference
utils.py:88-151, called at style.py:24,67,71).  The reference calls `torch.symeig`,
which modern torch removed; its own `except RuntimeError` then turns the whole function
into a no-op.  This version performs the transfer the code describes - on the device
(csrc/image.hip: one reduction pass, a 3x3 eigen-problem in one thread, one colour-map
pass) - and draws the jitter from the global RNG in the same order so seeded runs see
the same numbers.
"""
import numpy as np
import torch as th


def name(s):
    """File stem used to build output names (reference utils.py:53-54, config.py:94)."""
    return s.split("/")[-1].split(".")[0]


def info(x, y=None, z=None):
    """Debug print of min/mean/max/shape of one or two tensors (reference utils.py:10-50)."""
    def stats(t):
        return [f"{float(t.min()):.2f}", f"{float(t.mean()):.2f}", f"{float(t.max()):.2f}", t.shape]
    parts = ([] if z is None else [z]) + stats(x) + ([] if y is None else stats(y))
    print(*parts)


def fetch(path_or_url):
    """Open a local path (reference utils.py:69-72); there is no network in this build."""
    if path_or_url.startswith("http://") or path_or_url.startswith("https://"):
        raise OSError("remote inputs are not supported in this offline build: " + path_or_url)
    return open(path_or_url, "rb")


def wrapping_slice(tensor, start, length, return_indices=False):
    """Window of `length` frames starting at `start`, wrapping around the end (reference utils.py:76-85)."""
    n = tensor.shape[0]
    if n == 1:
        idx = th.zeros(1, dtype=th.int64)
    elif start + length <= n:
        idx = th.arange(start, start + length)
    else:
        idx = th.cat((th.arange(start, n), th.arange(0, (start + length) % n)))
    return idx if return_indices else tensor[idx]


def get_histogram(tensor, eps):
    """Channel means and regularised covariance of ONE frame given as a b,w,h,c view (reference utils.py:88-93), from the
    device reduction: (mu [3], None, cov [3,3]) as float64 CPU tensors.  (The reference also returns the centred C x N matrix;
    the colour map is applied by a kernel here, so nothing of that size is materialised.)"""
    import hip
    frame = _device(tensor.permute(0, 3, 2, 1))[:1]  # back to b,c,h,w
    st = hip.channel_stats(frame)[0].cpu()
    n = frame[0, 0].numel()
    mu = st[:3] / n
    cov = th.empty(3, 3, dtype=th.float64)
    k = 3
    for i in range(3):
        for j in range(i, 3):
            cov[i, j] = cov[j, i] = st[k] / n - mu[i] * mu[j]
            k += 1
    return mu, None, cov + eps * th.eye(3, dtype=th.float64)


def _device(t):
    return t.to(device="cuda", dtype=th.float32).contiguous()


_SOURCE_CACHE = {}


def _source_frames(source, per_frame):
    """The source on the device, cached per tensor (the style images are matched against on every call of a job)."""
    key = (source.data_ptr(), tuple(source.shape), source._version, str(source.device), per_frame)
    hit = _SOURCE_CACHE.get(key)
    if hit is None:
        dev = _device(source)
        if per_frame:
            dev = dev.mean(0, keepdim=True)  # utils.py:118: matched against the source clip's mean frame
        if len(_SOURCE_CACHE) > 64:
            _SOURCE_CACHE.clear()
        hit = _SOURCE_CACHE[key] = (dev, source)  # the tensor itself is kept so that its data_ptr stays unique
    return hit[0]


def draw_match_noise(target_shape, source_tensor, mode="avg"):
    """The global-RNG draws one match_histogram call makes (torch jitter, numpy frame choice), in its order, WITHOUT running
    it: lets a caller that batches frames draw for every frame in the reference's per-frame order first and hand each draw
    to the matching call later (`_noise=`)."""
    if not mode:
        return None
    per_frame = mode == "avg"
    sources = source_tensor if isinstance(source_tensor, list) else [source_tensor]
    B, _, H, W = target_shape
    out = []
    for source in sources:
        pick = None if per_frame else int(np.random.randint(0, source.shape[0]))
        sh, sw = source.shape[2], source.shape[3]
        for _ in range(B if per_frame else 1):
            out.append((pick, th.randn(size=(1 if per_frame else B, W, H, 3)), th.randn(size=(1, sw, sh, 3))))
    return out


def match_histogram(target_tensor, source_tensor, eps=1e-2, mode="avg", _noise=None):
    """PCA colour transfer of `target_tensor` towards the colour statistics of each source, averaged over sources
    (reference utils.py:96-151; called before / after every scale, style.py:24,67,71, and per frame, style.py:292).
    `mode` falsy -> identity; "avg" -> every target frame against the source's mean frame; anything else -> the whole
    target against one random source frame.

    Runs on the MI355X (csrc/image.hip): the result lives on the device the target lives on (a CPU target is uploaded and
    the result brought back), the statistics, the 3x3 eigen-problem and the colour map never touch the host - no
    synchronisation.  The jitter `1e-3 * randn` is drawn from torch's GLOBAL CPU generator in the reference's order
    (target draw, then source draw, per frame, per source) and uploaded, so seeded runs see the same numbers and every
    later draw of the job (e.g. --init random) is unchanged.  Where the reference's `except RuntimeError` would return the
    untouched input (non-finite statistics, singular square root), so does the device path (flag in the solve kernel)."""
    import hip
    if not mode:
        return target_tensor
    per_frame = mode == "avg"
    sources = source_tensor if isinstance(source_tensor, list) else [source_tensor]
    tgt = _device(target_tensor)
    B, C, H, W = tgt.shape
    if C != 3:
        raise ValueError("match_histogram works on 3-channel images")
    out = th.empty_like(tgt)
    n_solves = len(sources) * (B if per_frame else 1)
    coef = th.empty(n_solves, 16, device=tgt.device)
    plan = []  # (batch view being matched, its slice of `out`, its jitter, coefficient row)
    drawn = iter(_noise) if _noise is not None else None
    for source in sources:
        src = _source_frames(source, per_frame)
        pick = None
        if not per_frame and drawn is None:
            pick = np.random.randint(0, src.shape[0])  # utils.py:120 (numpy's global RNG, as there)
        sh, sw = src.shape[2], src.shape[3]
        for idx in range(B if per_frame else 1):
            batch, dst = (tgt[idx:idx + 1], out[idx:idx + 1]) if per_frame else (tgt, out)
            if drawn is not None:  # draws made earlier by draw_match_noise, in the reference's order
                pick, noise_t, noise_s = next(drawn)
            else:
                # utils.py:123-124: randn(size=frame.shape) with the frame viewed as (b, W, H, C) - the layout the kernels index
                noise_t = th.randn(size=(batch.shape[0], W, H, 3))
                noise_s = th.randn(size=(1, sw, sh, 3))
            if not per_frame and idx == 0:
                src = src[pick][None]
            noise_t = noise_t.to(tgt.device, non_blocking=True)
            noise_s = noise_s.to(tgt.device, non_blocking=True)
            st_t = hip.channel_stats(batch, noise_t)
            st_s = hip.channel_stats(src[:1], noise_s)
            hip.color_match_solve(st_t, H * W, st_s, sh * sw, eps, coef[len(plan)])
            plan.append((batch, dst, noise_t, len(plan)))
    written = set()
    for batch, dst, noise_t, row in plan:  # every solve is enqueued before the first map: a failed one makes the whole call the identity
        key = dst.data_ptr()
        hip.color_match_apply(batch, noise_t, 1e-3, coef[row], coef, 1.0 / len(sources), key in written, dst)
        written.add(key)
    return out if target_tensor.is_cuda else out.cpu()


def limit_host_threads(limit=16):
    """The host side of the path only touches 3-channel images (resizing, histogram matching, PNG coding).  On a
    256-core MI355X host torch defaults to 128 intra-op threads, and fork/join overhead then dominates those small ops
    (measured on BASELINE config 4: 2.7 s instead of 0.1 s of host work per frame).  MAUA_HOST_THREADS overrides."""
    import os
    import torch
    want = int(os.environ.get("MAUA_HOST_THREADS", limit))
    if want > 0 and torch.get_num_threads() > want:
        torch.set_num_threads(want)
