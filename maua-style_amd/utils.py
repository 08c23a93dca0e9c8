"""Small helpers with the reference's names (reference utils.py).

`match_histogram` is the colour transfer run before/after every scale (reference
utils.py:88-151, called at style.py:24,67,71).  The reference calls `torch.symeig`,
which modern torch removed; its own `except RuntimeError` then turns the whole function
into a no-op.  This version performs the transfer the code describes, with
`torch.linalg.eigh(UPLO="U")` in place of `symeig(upper=True)`, and draws from the global
RNG in the same order so seeded runs see the same jitter.
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
    """Channel mean, centred channel matrix (C x N) and regularised covariance of a b,w,h,c tensor
    (reference utils.py:88-93)."""
    mu = tensor.mean(list(range(tensor.dim() - 1)))
    h = (tensor - mu).permute(0, 3, 1, 2).reshape(tensor.size(3), -1)
    cov = h @ h.T / h.shape[1] + eps * th.eye(h.shape[0])
    return mu, h, cov


def _sqrt_psd(cov):
    eva, eve = th.linalg.eigh(cov, UPLO="U")
    root = th.sqrt(th.diagflat(eva))
    root[root != root] = 0  # negative eigenvalues -> nan -> 0, as the reference does
    return eve @ root @ eve.T


def match_histogram(target_tensor, source_tensor, eps=1e-2, mode="avg"):
    """PCA colour transfer of `target_tensor` towards the colour statistics of each source, averaged over
    sources (reference utils.py:96-151).  `mode` falsy -> identity; "avg" -> per-frame matching against the
    source's mean frame; anything else -> one random source frame."""
    if not mode:
        return target_tensor
    per_frame = mode == "avg"
    sources = source_tensor if isinstance(source_tensor, list) else [source_tensor]
    out = th.zeros_like(target_tensor)
    for source in sources:
        tgt = target_tensor.permute(0, 3, 2, 1)  # b,w,h,c
        src = source.permute(0, 3, 2, 1)
        if per_frame:
            src = src.mean(0).unsqueeze(0)
        else:
            src = src[np.random.randint(0, src.shape[0])].unsqueeze(0)
        matched = th.zeros_like(tgt)
        for idx in range(tgt.shape[0] if per_frame else 1):
            frame = tgt[idx].unsqueeze(0) if per_frame else tgt
            _, t, cov_t = get_histogram(frame + 1e-3 * th.randn(size=frame.shape), eps)
            mu_s, _, cov_s = get_histogram(src + 1e-3 * th.randn(size=src.shape), eps)
            q_t, q_s = _sqrt_psd(cov_t), _sqrt_psd(cov_s)
            ts = q_s @ th.inverse(q_t) @ t
            m = ts.reshape(*frame.permute(0, 3, 1, 2).shape).permute(0, 2, 3, 1) + mu_s
            if per_frame:
                matched[idx] = m
            else:
                matched = m
        out += matched.permute(0, 3, 2, 1) / len(sources)
    return out


def limit_host_threads(limit=16):
    """The host side of the path only touches 3-channel images (resizing, histogram matching, PNG coding).  On a
    256-core MI355X host torch defaults to 128 intra-op threads, and fork/join overhead then dominates those small ops
    (measured on BASELINE config 4: 2.7 s instead of 0.1 s of host work per frame).  MAUA_HOST_THREADS overrides."""
    import os
    import torch
    want = int(os.environ.get("MAUA_HOST_THREADS", limit))
    if want > 0 and torch.get_num_threads() > want:
        torch.set_num_threads(want)
