"""ctypes binding of libmaua_hip.so (declared in include/maua_hip.h).

There is NO fallback: if the library cannot be loaded, or a kernel reports an error, a
`HipError` is raised.  Tensors handed to the wrappers must be contiguous float32 ROCm
tensors; every call is enqueued on torch's current stream.
"""
import ctypes
import threading
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libmaua_hip.so")

c_f = ctypes.c_float
c_i = ctypes.c_int
c_i64 = ctypes.c_int64
c_p = ctypes.c_void_p
c_sz = ctypes.c_size_t

# name -> (restype, argtypes); mirrors include/maua_hip.h one to one
SIGNATURES = {
    "maua_abi_version": (c_i, []),
    "maua_last_error": (ctypes.c_char_p, []),
    "maua_conv_pack_filters": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "maua_conv_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv2d_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv2d_bwd_data": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz,
                                   c_p]),
    "maua_conv_x6_bank_bytes": (c_sz, [c_i, c_i]),
    "maua_conv_pack_filters_x6": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    "maua_conv_x6_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv_x3_bank_bytes": (c_sz, [c_i, c_i]),
    "maua_conv_pack_filters_x3": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_p]),
    "maua_conv_x3w_bank_bytes": (c_sz, [c_i, c_i]),
    "maua_conv_pack_filters_x3w": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_p]),
    "maua_conv_x3w_supported": (c_i, [c_i, c_i, c_i, c_i]),
    "maua_conv_x3w_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv3x3_x3w": (c_i, [c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv_kxk_x3_bank_bytes": (c_sz, [c_i, c_i, c_i]),
    "maua_conv_pack_filters_kxk_x3": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_f, c_p]),
    "maua_conv_kxk_x3_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv_kxk_x3": (c_i, [c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv1x1_x3_workspace_bytes": (c_sz, [c_i, c_i, c_i64, c_i]),
    "maua_conv1x1_x3": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i64, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv_x3_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv3x3_x3": (c_i, [c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv3x3_x6": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv_image_bank_bytes": (c_sz, [c_i, c_i]),
    "maua_conv_pack_filters_image": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    "maua_conv3x3_image": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "maua_conv_image_gram_slabs": (c_i, [c_i, c_i, c_i]),
    "maua_conv_image_supported": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv3x3_image_gram": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "maua_relu_fwd": (c_i, [c_p, c_i64, c_p]),
    "maua_relu_bwd": (c_i, [c_p, c_p, c_p, c_i64, c_p]),
    "maua_pool_out_size": (c_i, [c_i, c_i, c_i, c_i]),
    "maua_pool2d_fwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "maua_pool2d_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "maua_pool2x2_codes_supported": (c_i, [c_i, c_i, c_i, c_i]),
    "maua_pool2x2_fwd_codes": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "maua_pool2x2_bwd_codes": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "maua_gram_workspace_bytes": (c_sz, [c_i, c_i64]),
    "maua_gram_block": (c_i, [c_i, c_i64]),
    "maua_gram_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i64, c_f, c_i, c_p, c_sz, c_p]),
    "maua_reduce_workspace_bytes": (c_sz, [c_i64]),
    "maua_mse_fwd_bwd": (c_i, [c_p, c_p, c_p, c_i64, c_f, c_f, c_i, c_i, c_p, c_p, c_sz, c_p]),
    "maua_mse_weighted_fwd_bwd": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i64, c_i, c_f, c_f, c_i, c_p, c_p, c_sz, c_p]),
    "maua_gram_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i64, c_i, c_p, c_sz, c_p]),
    "maua_tv_fwd_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_p, c_sz, c_p]),
    "maua_fill": (c_i, [c_p, c_i64, c_f, c_p]),
    "maua_depth_to_space": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "maua_conv_few_mfma_bank_bytes": (c_sz, []),
    "maua_conv_pack_filters_few_mfma": (c_i, [c_p, c_p, c_i, c_i, c_p]),
    "maua_conv_few_mfma_supported": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv3x3_few_mfma": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "maua_space_to_depth": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "maua_axpy": (c_i, [c_p, c_p, c_f, c_i64, c_p]),
    "maua_sum_small": (c_i, [c_p, c_i, c_p, c_p]),
    "maua_adam_step": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_f, c_f, c_f, c_f, c_p]),
    "maua_channel_stats_workspace_bytes": (c_sz, [c_i, c_i]),
    "maua_channel_stats": (c_i, [c_p, c_p, c_f, c_i, c_i, c_i, c_i, c_p, c_p, c_sz, c_p]),
    "maua_color_match_solve": (c_i, [c_p, c_i, c_i64, c_p, c_i64, c_f, c_p, c_p]),
    "maua_color_match_apply": (c_i, [c_p, c_p, c_f, c_p, c_p, c_i, c_f, c_i, c_i, c_i, c_i, c_i, c_p, c_p]),
    "maua_resize_bilinear": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_p]),
    "maua_deprocess_u8": (c_i, [c_p, c_p, c_i, c_i, c_f, c_f, c_f, c_p]),
    "maua_set_tuning": (c_i, [ctypes.c_char_p, ctypes.c_double]),
    "maua_get_tuning": (ctypes.c_double, [ctypes.c_char_p, ctypes.c_double]),
    "maua_set_split_batch_hint": (None, [c_i]),
    "maua_conv_arm_workspace": (c_i, [c_p, c_p, c_sz, c_i, c_p]),
    "maua_get_split_batch_hint": (c_i, []),
    "maua_conv_x3w_split": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv_x3q_bank_bytes": (c_sz, [c_i, c_i]),
    "maua_conv_pack_filters_x3q": (c_i, [c_p, c_p, c_p, c_i, c_i, c_f, c_p]),
    "maua_conv_x3q_supported": (c_i, [c_i, c_i, c_i, c_i]),
    "maua_conv_x3q_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv_x3q_split": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv_x3q_preferred": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv3x3_x3q": (c_i, [c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv3x3_x3q_relu_pool": (c_i, [c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv3x3_x3q_unpool": (c_i, [c_p, c_p, c_i, c_p, c_f, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv_x3p_supported": (c_i, [c_i, c_i, c_i, c_i, c_i]),
    "maua_conv_x3p_split": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv_x3p_workspace_bytes": (c_sz, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv_x3p_preferred": (c_i, [c_i, c_i, c_i, c_i, c_i, c_i]),
    "maua_conv_x3p_set_max_groups": (c_i, [c_i]),
    "maua_conv3x3_x3p": (c_i, [c_p, c_p, c_i, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv3x3_x3w_relu_pool": (c_i, [c_p, c_p, c_f, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv_x3w_dmat_bank_bytes": (c_sz, [c_i]),
    "maua_conv_pack_dmat_x3w": (c_i, [c_p, c_i, c_p, c_p, c_p]),
    "maua_conv_pack_dmat_x3w_batch": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p]),
    "maua_conv3x3_x3w_gram": (c_i, [c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_conv3x3_x3w_unpool": (c_i, [c_p, c_p, c_i, c_p, c_f, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p, c_sz, c_p]),
    "maua_loss_ledger_bytes": (c_sz, [c_i, c_i]),
    "maua_mse_fwd_bwd_ledger": (c_i, [c_p, c_p, c_p, c_i64, c_f, c_f, c_i, c_i, c_p, c_i, c_p]),
    "maua_tv_fwd_bwd_ledger": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p, c_i, c_p]),
    "maua_gram_mse_ledger_supported": (c_i, [c_i]),
    "maua_gram_fwd_mse_ledger": (c_i, [c_p, c_p, c_p, c_i, c_i64, c_f, c_i, c_p, c_p, c_f, c_f, c_p, c_i, c_p, c_sz, c_p]),
    "maua_gram_partial": (c_i, [c_p, c_p, c_i, c_i64, c_i, c_p, c_sz, c_p]),
    "maua_gram_finish_mse_batch": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "maua_gram_partial_batch": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "maua_gram_row_means": (c_i, [c_p, c_p, c_i, c_i64, c_p, c_sz, c_p]),
    "maua_loss_ledger_sum": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p]),
    "maua_loss_ledger_sum_f64": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    "maua_lbfgs_state_bytes": (c_sz, [c_i64, c_i]),
    "maua_lbfgs_init": (c_i, [c_p, c_sz, c_i64, c_i, c_p]),
    "maua_lbfgs_iterate": (c_i, [c_p, c_p, c_p, c_p, c_i64, c_i, c_f, c_f, c_f, c_p]),
    "maua_lbfgs_status": (c_i, [c_p, c_i64, c_i, c_p, c_p]),
}


class HipError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libmaua_hip.so once; raise HipError when it is missing (no CPU path exists)."""
    global _lib, LIB_PATH
    if _lib is None:
        LIB_PATH = os.environ.get("MAUA_HIP_LIB", LIB_PATH)  # developer knob: A/B another build of the same library
        if not os.path.exists(LIB_PATH):
            raise HipError(
                f"{LIB_PATH} not found: build it with `python maua-style_amd/build_native.py` "
                "(this package has no CPU or PyTorch fallback)")
        try:
            L = ctypes.CDLL(LIB_PATH)
        except OSError as e:
            raise HipError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so is stale: also loud
            fn.restype = res
            fn.argtypes = args
        _lib = L
        here = os.path.dirname(os.path.abspath(__file__))
        if here not in sys.path:  # (tools that load this module by path: the planner configuration lives beside it)
            sys.path.insert(0, here)
        import plan
        plan.forward_to_library(L)  # the library's share of the planner configuration (it never reads the environment)
        plan.warn_ignored()         # MAUA_* variables nothing reads any more: said once, loudly
    return _lib


def apply_plan():
    """Hand the planner configuration's library fields to the library again (after a programmatic change of plan.OVERRIDES)."""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import plan
    plan.forward_to_library(lib())


def _check(rc, what):
    if rc != 0:
        msg = lib().maua_last_error().decode(errors="replace")
        raise HipError(f"{what} failed (rc={rc}): {msg}")


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise HipError("libmaua_hip needs ROCm device tensors (got a CPU tensor): there is no CPU path")
    if t.dtype != torch.float32 and t.dtype != torch.uint8 and t.dtype != torch.float64:
        raise HipError(f"unsupported dtype {t.dtype}")
    if not t.is_contiguous():
        raise HipError("tensor must be contiguous")
    return t.data_ptr()


def _stream():
    """hipStream_t of torch's current stream on the current device.  The raw getters are ~20x cheaper than building a
    torch.cuda.Stream object per launch (19 us, which made eager iterations at 512x512 CPU-bound)."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:  # private API moved: the public, slower spelling
        return torch.cuda.current_stream().cuda_stream


def _f32(t, name):
    if t is not None and t.dtype != torch.float32:
        raise HipError(f"{name}: expected float32, got {t.dtype}")
    return t


# ------------------------------------------------------------------------------------------
# thin, shape-aware wrappers
# ------------------------------------------------------------------------------------------
def conv_pack_filters(w):
    """OIHW weights -> (forward bank [taps][cin][cout], backward bank [taps][cout][cin])."""
    cout, cin, kh, kw = w.shape
    wf = torch.empty(kh * kw, cin, cout, device=w.device, dtype=torch.float32)
    wb = torch.empty(kh * kw, cout, cin, device=w.device, dtype=torch.float32)
    _check(lib().maua_conv_pack_filters(_ptr(_f32(w, "w")), _ptr(wf), _ptr(wb), cout, cin, kh, kw, _stream()),
           "maua_conv_pack_filters")
    return wf, wb


def conv_out_hw(h, w, k, stride, pad):
    if stride <= 0 or k <= 0 or pad < 0:
        raise HipError(f"conv geometry k={k} stride={stride} pad={pad} is invalid")
    if h + 2 * pad < k or w + 2 * pad < k:
        raise HipError(f"conv input {h}x{w} (pad {pad}) smaller than the {k}x{k} filter")
    return (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1


def conv_workspace_bytes(n, cin, h, w, cout, k, stride, pad):
    return lib().maua_conv_workspace_bytes(n, cin, h, w, cout, k, k, stride, pad)


def _ws_args(workspace, need, device):
    """(pointer, bytes) of a split-K workspace: the caller's when given, a fresh one when the geometry wants one."""
    if workspace is None and need:
        workspace = torch.empty(need, dtype=torch.uint8, device=device)
    if workspace is None:
        return None, 0
    return workspace.data_ptr(), workspace.numel() * workspace.element_size()


def conv2d_fwd(x, wf, bias, k, stride, pad, relu, out=None, in_mask=None, accumulate=False, workspace=None):
    n, cin, h, w = x.shape
    cout = wf.shape[2]
    oh, ow = conv_out_hw(h, w, k, stride, pad)
    if out is None:
        out = torch.empty(n, cout, oh, ow, device=x.device, dtype=torch.float32)
    wp, wn = _ws_args(workspace, conv_workspace_bytes(n, cin, h, w, cout, k, stride, pad) if workspace is None else 0, x.device)
    _check(lib().maua_conv2d_fwd(_ptr(_f32(x, "x")), _ptr(in_mask), _ptr(wf), _ptr(bias), _ptr(out), n, cin, h, w, cout,
                                 k, k, stride, pad, int(relu), int(accumulate), wp, wn, _stream()), "maua_conv2d_fwd")
    return out


def conv2d_bwd_data(gy, out_mask, wb, w_oihw, in_shape, k, stride, pad, out=None, accumulate=False, in_relu_mask=None,
                    workspace=None):
    n, cin, h, w = in_shape
    cout = gy.shape[1]
    if out is None:
        out = torch.empty(n, cin, h, w, device=gy.device, dtype=torch.float32)
    need = 0
    if workspace is None and stride == 1 and k - 1 - pad >= 0:
        need = conv_workspace_bytes(n, cout, gy.shape[2], gy.shape[3], cin, k, 1, k - 1 - pad)
    wp, wn = _ws_args(workspace, need, gy.device)
    _check(lib().maua_conv2d_bwd_data(_ptr(_f32(gy, "gy")), _ptr(out_mask), _ptr(wb), _ptr(w_oihw), _ptr(in_relu_mask),
                                      _ptr(out), n, cin, h, w, cout, k, k, stride, pad, int(accumulate), wp, wn, _stream()),
           "maua_conv2d_bwd_data")
    return out


def conv_pack_filters_x6(w):
    """OIHW 3x3 weights -> (forward bank, backward-data bank) of pre-split bf16 triples (uint8 tensors)."""
    cout, cin, kh, kw = w.shape
    if kh != 3 or kw != 3:
        raise HipError("the bf16x6 path covers 3x3 filters")
    bf = torch.empty(lib().maua_conv_x6_bank_bytes(cout, cin), dtype=torch.uint8, device=w.device)
    bb = torch.empty(lib().maua_conv_x6_bank_bytes(cin, cout), dtype=torch.uint8, device=w.device)
    _check(lib().maua_conv_pack_filters_x6(_ptr(_f32(w, "w")), bf.data_ptr(), bb.data_ptr(), cout, cin, _stream()),
           "maua_conv_pack_filters_x6")
    return bf, bb


def conv_pack_filters_x3(w):
    """OIHW 3x3 weights -> (forward bank, backward-data bank, w_scale) of pre-split, pre-scaled fp16 pairs.  w_scale is the
    power of two that brings max|w| into [32, 64) (one host read of the maximum per weight version)."""
    import math
    cout, cin = w.shape[:2]
    m = float(w.abs().max())
    w_scale = 2.0 ** (5 - math.floor(math.log2(m))) if m > 0 and math.isfinite(m) else 1.0
    wc = _f32(w, "w").contiguous()
    bf = torch.empty(lib().maua_conv_x3_bank_bytes(cout, cin), dtype=torch.uint8, device=w.device)
    bb = torch.empty(lib().maua_conv_x3_bank_bytes(cin, cout), dtype=torch.uint8, device=w.device)
    _check(lib().maua_conv_pack_filters_x3(_ptr(wc), bf.data_ptr(), bb.data_ptr(), cout, cin, w_scale, _stream()),
           "maua_conv_pack_filters_x3")
    return bf, bb, w_scale


def conv_x3_workspace_bytes(n, cin, h, w, cout, pad):
    return lib().maua_conv_x3_workspace_bytes(n, cin, h, w, cout, pad)


def conv3x3_x3(x, bank, w_scale, bias, cout, pad, relu, out=None, out_relu_mask=None, accumulate=False, workspace=None):
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, cout, h + 2 * pad - 2, w + 2 * pad - 2, device=x.device, dtype=torch.float32)
    wp, wn = _ws_args(workspace, conv_x3_workspace_bytes(n, cin, h, w, cout, pad) if workspace is None else 0, x.device)
    _check(lib().maua_conv3x3_x3(_ptr(_f32(x, "x")), bank.data_ptr(), float(w_scale), _ptr(bias), _ptr(out_relu_mask), _ptr(out), n,
                                 cin, h, w, cout, pad, int(relu), int(accumulate), wp, wn, _stream()), "maua_conv3x3_x3")
    return out


def conv_pack_filters_x3w(w):
    """OIHW 3x3 weights -> (forward bank, backward-data bank, w_scale) for conv_x3w.hip (16-channel chunks; same split and
    the same power-of-two filter scale as conv_pack_filters_x3)."""
    import math
    cout, cin = w.shape[:2]
    m = float(w.abs().max())
    w_scale = 2.0 ** (5 - math.floor(math.log2(m))) if m > 0 and math.isfinite(m) else 1.0
    wc = _f32(w, "w").contiguous()
    bf = torch.empty(lib().maua_conv_x3w_bank_bytes(cout, cin), dtype=torch.uint8, device=w.device)
    bb = torch.empty(lib().maua_conv_x3w_bank_bytes(cin, cout), dtype=torch.uint8, device=w.device)
    _check(lib().maua_conv_pack_filters_x3w(_ptr(wc), bf.data_ptr(), bb.data_ptr(), cout, cin, w_scale, _stream()),
           "maua_conv_pack_filters_x3w")
    return bf, bb, w_scale


def conv_x3w_supported(cin, h, w, pad):
    return bool(lib().maua_conv_x3w_supported(int(cin), int(h), int(w), int(pad)))


def conv_x3w_workspace_bytes(n, cin, h, w, cout, pad):
    return lib().maua_conv_x3w_workspace_bytes(n, cin, h, w, cout, pad)


def conv3x3_x3w(x, bank, w_scale, bias, cout, pad, relu, out=None, out_relu_mask=None, accumulate=False, workspace=None):
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, cout, h + 2 * pad - 2, w + 2 * pad - 2, device=x.device, dtype=torch.float32)
    wp, wn = _ws_args(workspace, conv_x3w_workspace_bytes(n, cin, h, w, cout, pad) if workspace is None else 0, x.device)
    _check(lib().maua_conv3x3_x3w(_ptr(_f32(x, "x")), bank.data_ptr(), float(w_scale), _ptr(bias), _ptr(out_relu_mask), _ptr(out), n,
                                  cin, h, w, cout, pad, int(relu), int(accumulate), wp, wn, _stream()), "maua_conv3x3_x3w")
    return out


def conv_pack_filters_x3q(w):
    """OIHW 3x3 weights -> (forward bank, backward-data bank, w_scale) for conv_x3q.hip (32-channel chunks; same split and
    the same power-of-two filter scale as conv_pack_filters_x3)."""
    import math
    cout, cin = w.shape[:2]
    m = float(w.abs().max())
    w_scale = 2.0 ** (5 - math.floor(math.log2(m))) if m > 0 and math.isfinite(m) else 1.0
    wc = _f32(w, "w").contiguous()
    bf = torch.empty(lib().maua_conv_x3q_bank_bytes(cout, cin), dtype=torch.uint8, device=w.device)
    bb = torch.empty(lib().maua_conv_x3q_bank_bytes(cin, cout), dtype=torch.uint8, device=w.device)
    _check(lib().maua_conv_pack_filters_x3q(_ptr(wc), bf.data_ptr(), bb.data_ptr(), cout, cin, w_scale, _stream()),
           "maua_conv_pack_filters_x3q")
    return bf, bb, w_scale


def conv_x3q_supported(cin, h, w, pad):
    return bool(lib().maua_conv_x3q_supported(int(cin), int(h), int(w), int(pad)))


def conv_x3q_workspace_bytes(n, cin, h, w, cout, pad):
    return lib().maua_conv_x3q_workspace_bytes(n, cin, h, w, cout, pad)


def conv_x3q_preferred(n, cin, h, w, cout, pad):
    return bool(lib().maua_conv_x3q_preferred(int(n), int(cin), int(h), int(w), int(cout), int(pad)))


def conv_x3q_split(n, cin, h, w, cout, pad):
    return int(lib().maua_conv_x3q_split(int(n), int(cin), int(h), int(w), int(cout), int(pad)))


def conv3x3_x3q(x, bank, w_scale, bias, cout, pad, relu, out=None, out_relu_mask=None, accumulate=False, workspace=None):
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, cout, h + 2 * pad - 2, w + 2 * pad - 2, device=x.device, dtype=torch.float32)
    wp, wn = _ws_args(workspace, conv_x3q_workspace_bytes(n, cin, h, w, cout, pad) if workspace is None else 0, x.device)
    _check(lib().maua_conv3x3_x3q(_ptr(_f32(x, "x")), bank.data_ptr(), float(w_scale), _ptr(bias), _ptr(out_relu_mask), _ptr(out), n,
                                  cin, h, w, cout, pad, int(relu), int(accumulate), wp, wn, _stream()), "maua_conv3x3_x3q")
    return out


def _unpooled_hw(ph, pw, pad, out):
    """Extent of the plane a 2x2 / 2 floor-mode max pool reduced to ph x pw: taken from the output the caller provides (the plane may be
    odd: 181 -> 90), 2 ph x 2 pw without one."""
    if out is None:
        return 2 * ph, 2 * pw
    h, w = out.shape[2] - 2 * pad + 2, out.shape[3] - 2 * pad + 2
    if h // 2 != ph or w // 2 != pw:
        raise ValueError(f"unpool: a {ph}x{pw} pooled map does not come from a {h}x{w} plane")
    return h, w


def conv3x3_x3q_relu_pool(x, bank, w_scale, bias, cout, pad, pooled, codes, workspace=None):
    """conv + bias + ReLU + 2x2 / 2 max pool on conv_x3q.hip: writes `pooled` and the pool's decision bytes only (conv3x3_x3w_relu_pool's
    contract)."""
    n, cin, h, w = x.shape
    wp, wn = (workspace.data_ptr(), workspace.numel() * workspace.element_size()) if workspace is not None else (None, 0)
    _check(lib().maua_conv3x3_x3q_relu_pool(_ptr(_f32(x, "x")), bank.data_ptr(), float(w_scale), _ptr(bias), _ptr(pooled),
                                            codes.data_ptr(), n, cin, h, w, cout, pad, wp, wn, _stream()), "maua_conv3x3_x3q_relu_pool")
    return pooled


def conv3x3_x3q_unpool(pooled_x, codes, honour_relu_bit, bank, w_scale, cout, pad, out=None, out_relu_mask=None, workspace=None):
    """Backward-data pass on conv_x3q.hip staged straight from the pooled map's gradient and the pool's decision bytes
    (conv3x3_x3w_unpool's contract, without the Gram term)."""
    n, cin, hp, wp_ = pooled_x.shape
    h, w = _unpooled_hw(hp, wp_, pad, out)
    if out is None:
        out = torch.empty(n, cout, h + 2 * pad - 2, w + 2 * pad - 2, device=pooled_x.device, dtype=torch.float32)
    wp, wn = _ws_args(workspace, conv_x3q_workspace_bytes(n, cin, h, w, cout, pad) if workspace is None else 0, pooled_x.device)
    _check(lib().maua_conv3x3_x3q_unpool(_ptr(_f32(pooled_x, "pooled_x")), codes.data_ptr(), int(bool(honour_relu_bit)), bank.data_ptr(),
                                         float(w_scale), _ptr(out_relu_mask) if out_relu_mask is not None else None, _ptr(out), n, cin, h, w,
                                         cout, pad, wp, wn, _stream()), "maua_conv3x3_x3q_unpool")
    return out


def conv_x3p_supported(cin, h, w, cout, pad):
    return bool(lib().maua_conv_x3p_supported(int(cin), int(h), int(w), int(cout), int(pad)))


def conv_x3p_split(n, cin, h, w, cout, pad):
    return int(lib().maua_conv_x3p_split(int(n), int(cin), int(h), int(w), int(cout), int(pad)))


def conv_x3p_preferred(n, cin, h, w, cout, pad):
    return bool(lib().maua_conv_x3p_preferred(int(n), int(cin), int(h), int(w), int(cout), int(pad)))


def conv_x3p_set_max_groups(groups):
    """Tests: workgroups a conv_x3p launch may use (multiple of 8; 0 = no override: one per CU / the planner's x3p_groups); returns the
    previous override (0 = none), so passing it back restores the state exactly."""
    return int(lib().maua_conv_x3p_set_max_groups(int(groups)))


def conv_x3p_workspace_bytes(n, cin, h, w, cout, pad):
    return lib().maua_conv_x3p_workspace_bytes(n, cin, h, w, cout, pad)


def conv3x3_x3p(x, bank, w_scale, bias, cout, pad, relu, out=None, out_relu_mask=None, workspace=None, in_codes=None, honour_relu_bit=True,
                dmat_bank=None, dmat_inv_scale=None, pool_codes=None):
    """The persistent fp16x3 3x3 kernel (conv_x3p.hip) in any of its forms: plain / masked, staged from a pooled gradient (`in_codes`: x is
    the pooled map's gradient, `out` fixes the full-size plane), with the Gram backward along (`dmat_bank`, out_relu_mask = F), or with ReLU
    + 2x2 max pool in the epilogue (`pool_codes`: out = the pooled map).  Banks: conv_pack_filters_x3q's."""
    n, cin = x.shape[0], x.shape[1]
    if in_codes is not None:
        h, w = _unpooled_hw(x.shape[2], x.shape[3], pad, out)
    else:
        h, w = x.shape[2], x.shape[3]
    oh, ow = h + 2 * pad - 2, w + 2 * pad - 2
    if out is None:
        out = torch.empty((n, cout, oh // 2, ow // 2) if pool_codes is not None else (n, cout, oh, ow), device=x.device, dtype=torch.float32)
    wp, wn = _ws_args(workspace, conv_x3p_workspace_bytes(n, cin, h, w, cout, pad) if workspace is None else 0, x.device)
    _check(lib().maua_conv3x3_x3p(_ptr(_f32(x, "x")), in_codes.data_ptr() if in_codes is not None else None, int(bool(honour_relu_bit)),
                                  bank.data_ptr(), float(w_scale), _ptr(bias), _ptr(out_relu_mask) if out_relu_mask is not None else None,
                                  dmat_bank.data_ptr() if dmat_bank is not None else None,
                                  _ptr(dmat_inv_scale) if dmat_inv_scale is not None else None, _ptr(out),
                                  pool_codes.data_ptr() if pool_codes is not None else None, n, cin, h, w, cout, pad, int(relu), wp, wn,
                                  _stream()), "maua_conv3x3_x3p")
    return out


def conv_x3w_split(n, cin, h, w, cout, pad):
    """Channel-loop splits conv3x3_x3w would use for this geometry under the current batch hint (1 = one pass)."""
    return int(lib().maua_conv_x3w_split(int(n), int(cin), int(h), int(w), int(cout), int(pad)))


def conv3x3_x3w_relu_pool(x, bank, w_scale, bias, cout, pad, pooled, codes, workspace=None):
    """conv + bias + ReLU + 2x2 / 2 max pool: writes `pooled` and the pool's decision bytes only.  One launch without a workspace (one
    pass over the channels); with one, small grids split the channel loop and the pass that adds the slabs applies ReLU and pool."""
    n, cin, h, w = x.shape
    wp, wn = (workspace.data_ptr(), workspace.numel() * workspace.element_size()) if workspace is not None else (None, 0)
    _check(lib().maua_conv3x3_x3w_relu_pool(_ptr(_f32(x, "x")), bank.data_ptr(), float(w_scale), _ptr(bias), _ptr(pooled),
                                            codes.data_ptr(), n, cin, h, w, cout, pad, wp, wn, _stream()), "maua_conv3x3_x3w_relu_pool")
    return pooled


def conv_x3w_dmat_bank(c, device, frames=1):
    """(banks (frames, bytes), inverse scales (frames,)) for conv_pack_dmat_x3w / conv3x3_x3w_gram; None when c is not a multiple
    of 16."""
    nbytes = lib().maua_conv_x3w_dmat_bank_bytes(int(c))
    if nbytes == 0:
        return None
    return torch.empty(frames, nbytes, dtype=torch.uint8, device=device), torch.ones(frames, dtype=torch.float32, device=device)


def conv_pack_dmat_x3w(dmat, bank, inv_scale):
    c = dmat.shape[0]
    _check(lib().maua_conv_pack_dmat_x3w(_ptr(_f32(dmat, "dmat")), c, bank.data_ptr(), _ptr(inv_scale), _stream()),
           "maua_conv_pack_dmat_x3w")


class DmatPackBatch:
    """conv_pack_dmat_x3w for up to four layers in one launch; the argument arrays are built once per engine plan."""

    def __init__(self, items):
        """items: list of (dmat (C, C), bank, inv_scale) as for conv_pack_dmat_x3w"""
        n = len(items)
        self.n, self.keep = n, items
        self.args = ((ctypes.c_void_p * n)(*[_ptr(_f32(d, "dmat")) for d, _, _ in items]), (ctypes.c_int * n)(*[int(d.shape[0]) for d, _, _ in items]),
                     (ctypes.c_void_p * n)(*[b.data_ptr() for _, b, _ in items]), (ctypes.c_void_p * n)(*[_ptr(i) for _, _, i in items]))

    def run(self):
        a = self.args
        _check(lib().maua_conv_pack_dmat_x3w_batch(self.n, a[0], a[1], a[2], a[3], _stream()), "maua_conv_pack_dmat_x3w_batch")


def conv3x3_x3w_gram(x, bank, w_scale, feature_map, dmat_bank, dmat_inv_scale, cout, pad, out=None, accumulate=False,
                     workspace=None):
    """Backward-data pass + Gram backward in one launch: out = [F > 0] * (conv(x) + D . F)."""
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, cout, h + 2 * pad - 2, w + 2 * pad - 2, device=x.device, dtype=torch.float32)
    wp, wn = _ws_args(workspace, conv_x3w_workspace_bytes(n, cin, h, w, cout, pad) if workspace is None else 0, x.device)
    _check(lib().maua_conv3x3_x3w_gram(_ptr(_f32(x, "x")), bank.data_ptr(), float(w_scale), _ptr(_f32(feature_map, "feature_map")),
                                       dmat_bank.data_ptr(), _ptr(dmat_inv_scale), _ptr(out), n, cin, h, w, cout, pad,
                                       int(accumulate), wp, wn, _stream()), "maua_conv3x3_x3w_gram")
    return out


def conv3x3_x3w_unpool(pooled_x, codes, honour_relu_bit, bank, w_scale, cout, pad, out=None, out_relu_mask=None, dmat_bank=None,
                       dmat_inv_scale=None, workspace=None):
    """Backward-data pass of a conv + ReLU + 2x2 max pool group straight from the POOLED gradient and the pool's decision bytes (the
    pool's backward pass happens while the kernel stages its input); with `dmat_bank`, the Gram backward of the style loss on the
    layer's input (out_relu_mask = F) goes along as in conv3x3_x3w_gram."""
    n, cin, ph, pw = pooled_x.shape
    h, w = _unpooled_hw(ph, pw, pad, out)
    if out is None:
        out = torch.empty(n, cout, h + 2 * pad - 2, w + 2 * pad - 2, device=pooled_x.device, dtype=torch.float32)
    wp, wn = _ws_args(workspace, conv_x3w_workspace_bytes(n, cin, h, w, cout, pad) if workspace is None else 0, pooled_x.device)
    _check(lib().maua_conv3x3_x3w_unpool(_ptr(_f32(pooled_x, "pooled_x")), codes.data_ptr(), int(bool(honour_relu_bit)), bank.data_ptr(),
                                         float(w_scale), _ptr(out_relu_mask) if out_relu_mask is not None else None,
                                         dmat_bank.data_ptr() if dmat_bank is not None else None,
                                         _ptr(dmat_inv_scale) if dmat_inv_scale is not None else None, _ptr(out), n, cin, h, w, cout, pad,
                                         wp, wn, _stream()), "maua_conv3x3_x3w_unpool")
    return out


def conv_pack_filters_kxk_x3(w):
    """OIHW k x k weights -> (forward bank, backward-data bank, w_scale) of pre-split, pre-scaled fp16 pairs (as
    conv_pack_filters_x3, with k*k taps)."""
    import math
    cout, cin, ks = w.shape[0], w.shape[1], w.shape[2]
    m = float(w.abs().max())
    w_scale = 2.0 ** (5 - math.floor(math.log2(m))) if m > 0 and math.isfinite(m) else 1.0
    wc = _f32(w, "w").contiguous()
    bf = torch.empty(lib().maua_conv_kxk_x3_bank_bytes(cout, cin, ks), dtype=torch.uint8, device=w.device)
    bb = torch.empty(lib().maua_conv_kxk_x3_bank_bytes(cin, cout, ks), dtype=torch.uint8, device=w.device)
    _check(lib().maua_conv_pack_filters_kxk_x3(_ptr(wc), bf.data_ptr(), bb.data_ptr(), cout, cin, ks, w_scale, _stream()),
           "maua_conv_pack_filters_kxk_x3")
    return bf, bb, w_scale


def conv_kxk_x3_workspace_bytes(n, cin, h, w, cout, ks, pad):
    return lib().maua_conv_kxk_x3_workspace_bytes(n, cin, h, w, cout, ks, pad)


def conv_kxk_x3(x, bank, w_scale, bias, cout, ks, pad, relu, out=None, out_relu_mask=None, accumulate=False, workspace=None):
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, cout, h + 2 * pad - ks + 1, w + 2 * pad - ks + 1, device=x.device, dtype=torch.float32)
    wp, wn = _ws_args(workspace, conv_kxk_x3_workspace_bytes(n, cin, h, w, cout, ks, pad) if workspace is None else 0, x.device)
    _check(lib().maua_conv_kxk_x3(_ptr(_f32(x, "x")), bank.data_ptr(), float(w_scale), _ptr(bias), _ptr(out_relu_mask), _ptr(out), n,
                                  cin, h, w, cout, ks, pad, int(relu), int(accumulate), wp, wn, _stream()), "maua_conv_kxk_x3")
    return out


def conv1x1_x3_workspace_bytes(n, cin, hw, cout):
    return lib().maua_conv1x1_x3_workspace_bytes(n, cin, hw, cout)


def conv1x1_x3(x, w_rowmajor, bias=None, relu=False, out=None, out_relu_mask=None, accumulate=False, workspace=None,
               x_shift=None):
    """y[n][co][p] (+)= sum_ci w[co][ci] x[n][ci][p] in fp16x3 arithmetic; x is [n, cin, ...] (any trailing plane shape);
    x_shift ([cin], optional) is subtracted from x first."""
    n, cin = x.shape[:2]
    hw = x[0, 0].numel()
    cout = w_rowmajor.shape[0]
    assert w_rowmajor.numel() == cout * cin and w_rowmajor.is_contiguous()
    if out is None:
        out = torch.empty((n, cout) + tuple(x.shape[2:]), device=x.device, dtype=torch.float32)
    wp, wn = _ws_args(workspace, conv1x1_x3_workspace_bytes(n, cin, hw, cout) if workspace is None else 0, x.device)
    _check(lib().maua_conv1x1_x3(_ptr(_f32(x, "x")), _ptr(x_shift), _ptr(_f32(w_rowmajor, "w")), _ptr(bias), _ptr(out_relu_mask), _ptr(out), n, cin,
                                 hw, cout, int(relu), int(accumulate), wp, wn, _stream()), "maua_conv1x1_x3")
    return out


def conv_x6_workspace_bytes(n, cin, h, w, cout, pad):
    return lib().maua_conv_x6_workspace_bytes(n, cin, h, w, cout, pad)



def conv_pack_filters_image(w, bias=None):
    """OIHW 3x3 weights (and the bias) of a layer that consumes 1-3 channels -> the bank of conv3x3_image (bf16 triples in MFMA lane
    order; the bias as one more filter column)."""
    cout, cin = w.shape[:2]
    bank = torch.empty(lib().maua_conv_image_bank_bytes(cout, cin), dtype=torch.uint8, device=w.device)
    _check(lib().maua_conv_pack_filters_image(_ptr(_f32(w, "w").contiguous()), _ptr(bias), bank.data_ptr(), cout, cin, _stream()),
           "maua_conv_pack_filters_image")
    return bank


def conv3x3_image(x, bank, cout, pad, relu, out=None):
    """The image layer (1-3 input channels) in exact bf16x6 arithmetic, forward; the bias is part of the bank."""
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, cout, h + 2 * pad - 2, w + 2 * pad - 2, device=x.device, dtype=torch.float32)
    _check(lib().maua_conv3x3_image(_ptr(_f32(x, "x")), bank.data_ptr(), _ptr(out), n, cin, h, w, cout, pad, int(relu), _stream()),
           "maua_conv3x3_image")
    return out


def conv_image_supported(n, cin, h, w, cout, pad):
    """Whether conv3x3_image takes this layer (1-3 channels, planes of fewer than 2^24 pixels: its offsets are 32-bit)."""
    return bool(lib().maua_conv_image_supported(int(n), int(cin), int(h), int(w), int(cout), int(pad)))


def conv_image_gram_slabs(h, w, pad):
    """Number of 64 x 64 slabs conv3x3_image_gram leaves for an h x w image."""
    return int(lib().maua_conv_image_gram_slabs(int(h), int(w), int(pad)))


def conv3x3_image_gram(x, bank, pad, out, gram_slabs):
    """conv3x3_image (one image, 64 channels, ReLU) that also leaves the partial Gram matrices of its output as split-K slabs in
    `gram_slabs` (a byte or float buffer of at least conv_image_gram_slabs(...) * 16 KiB)."""
    n, cin, h, w = x.shape
    assert n == 1 and gram_slabs.numel() * gram_slabs.element_size() >= conv_image_gram_slabs(h, w, pad) * 64 * 64 * 4
    _check(lib().maua_conv3x3_image_gram(_ptr(_f32(x, "x")), bank.data_ptr(), _ptr(out), gram_slabs.data_ptr(), cin, h, w, pad, _stream()),
           "maua_conv3x3_image_gram")
    return out


def conv3x3_x6(x, bank, bias, cout, pad, relu, out=None, out_relu_mask=None, accumulate=False, workspace=None):
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, cout, h + 2 * pad - 2, w + 2 * pad - 2, device=x.device, dtype=torch.float32)
    if workspace is None:
        need = conv_x6_workspace_bytes(n, cin, h, w, cout, pad)
        workspace = torch.empty(need, dtype=torch.uint8, device=x.device) if need else None
    _check(lib().maua_conv3x3_x6(_ptr(_f32(x, "x")), bank.data_ptr(), _ptr(bias), _ptr(out_relu_mask), _ptr(out), n, cin, h,
                                 w, cout, pad, int(relu), int(accumulate),
                                 workspace.data_ptr() if workspace is not None else None,
                                 workspace.numel() * workspace.element_size() if workspace is not None else 0, _stream()),
           "maua_conv3x3_x6")
    return out


def relu_(x):
    _check(lib().maua_relu_fwd(_ptr(_f32(x, "x")), x.numel(), _stream()), "maua_relu_fwd")
    return x


def relu_bwd(gy, y, out=None):
    if out is None:
        out = torch.empty_like(gy)
    _check(lib().maua_relu_bwd(_ptr(gy), _ptr(y), _ptr(out), gy.numel(), _stream()), "maua_relu_bwd")
    return out


def pool_out_size(n, k, stride, ceil_mode):
    return lib().maua_pool_out_size(n, k, stride, int(ceil_mode))


def pool2d_fwd(x, k, stride, ceil_mode, mode, out=None):
    n, c, h, w = x.shape
    oh, ow = pool_out_size(h, k, stride, ceil_mode), pool_out_size(w, k, stride, ceil_mode)
    if out is None:
        out = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.float32)
    _check(lib().maua_pool2d_fwd(_ptr(_f32(x, "x")), _ptr(out), n, c, h, w, k, stride, int(ceil_mode),
                                 0 if mode == "max" else 1, _stream()), "maua_pool2d_fwd")
    return out


def pool2d_bwd(gy, x, k, stride, ceil_mode, mode, out=None, relu_mask_by_x=False):
    n, c, h, w = x.shape
    if out is None:
        out = torch.empty_like(x)
    _check(lib().maua_pool2d_bwd(_ptr(_f32(gy, "gy")), _ptr(x), _ptr(out), n, c, h, w, k, stride, int(ceil_mode),
                                 0 if mode == "max" else 1, int(relu_mask_by_x), _stream()), "maua_pool2d_bwd")
    return out


def pool2x2_codes_supported(n, c, h, w):
    return bool(lib().maua_pool2x2_codes_supported(int(n), int(c), int(h), int(w)))


def pool2x2_fwd_codes(x, out, codes):
    """2x2/2 max pooling that also records each window's decision (one byte) for pool2x2_bwd_codes."""
    n, c, h, w = x.shape
    _check(lib().maua_pool2x2_fwd_codes(_ptr(_f32(x, "x")), _ptr(out), codes.data_ptr(), n, c, h, w, _stream()),
           "maua_pool2x2_fwd_codes")
    return out


def pool2x2_bwd_codes(gy, codes, out, relu_mask):
    n, c, h, w = out.shape
    _check(lib().maua_pool2x2_bwd_codes(_ptr(_f32(gy, "gy")), codes.data_ptr(), _ptr(out), n, c, h, w, int(relu_mask), _stream()),
           "maua_pool2x2_bwd_codes")
    return out


def gram_workspace_bytes(c, hw):
    return lib().maua_gram_workspace_bytes(c, hw)


def reduce_workspace_bytes(count):
    return lib().maua_reduce_workspace_bytes(count)


def _ws(workspace, need, device):
    if workspace is None or workspace.numel() * workspace.element_size() < need:
        workspace = torch.empty(max(need, 256), dtype=torch.uint8, device=device)
    return workspace


def gram_block(c, hw):
    """Edge of the blocks the Gram kernels multiply this shape in: 64, or 128 (MAUA_GRAM_T128=1, large layers)."""
    return int(lib().maua_gram_block(int(c), int(hw)))


def gram_fwd(f, scale, center=False, out=None, mean_out=None, workspace=None):
    """f: (1,C,H,W) or (C,HW).  Returns (gram CxC, row means or None)."""
    c = f.shape[1] if f.dim() == 4 else f.shape[0]
    hw = f.numel() // c
    if out is None:
        out = torch.empty(c, c, device=f.device, dtype=torch.float32)
    if center and mean_out is None:
        mean_out = torch.empty(c, device=f.device, dtype=torch.float32)
    need = gram_workspace_bytes(c, hw)
    workspace = _ws(workspace, need, f.device)
    _check(lib().maua_gram_fwd(_ptr(_f32(f, "f")), _ptr(out), _ptr(mean_out) if center else None, c, hw, float(scale),
                               int(center), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                               _stream()), "maua_gram_fwd")
    return out, (mean_out if center else None)


def loss_ledger(frames, slots, device):
    """Zero-filled ledger of frames x slots records for the *_ledger entry points (float64 tensor, frames x slots x stride)."""
    n = lib().maua_loss_ledger_bytes(int(frames), int(slots)) // 8
    return torch.zeros(frames, slots, n // (frames * slots), dtype=torch.float64, device=device)


def gram_mse_ledger_supported(c):
    return bool(lib().maua_gram_mse_ledger_supported(int(c)))


def gram_fwd_mse_ledger(f, scale, center, out, mean_out, target, dmat, loss_scale, grad_scale, ledger, slot, workspace=None):
    """gram_fwd + mse_fwd_bwd(gram, target -> dmat) with the loss left in record `slot` of `ledger` (one frame's records)."""
    c = f.shape[1] if f.dim() == 4 else f.shape[0]
    hw = f.numel() // c
    workspace = _ws(workspace, gram_workspace_bytes(c, hw), f.device)
    _check(lib().maua_gram_fwd_mse_ledger(_ptr(_f32(f, "f")), _ptr(out), _ptr(mean_out) if center else None, c, hw, float(scale),
                                          int(center), _ptr(_f32(target, "target")), _ptr(dmat), float(loss_scale),
                                          float(grad_scale), ledger.data_ptr(), int(slot), workspace.data_ptr(),
                                          workspace.numel() * workspace.element_size(), _stream()), "maua_gram_fwd_mse_ledger")
    return out


def gram_row_means(f, mean_out, workspace):
    """Row means of the covariance form into `mean_out` (the layer's workspace holds the fp64 partial sums meanwhile)."""
    c = f.shape[1] if f.dim() == 4 else f.shape[0]
    _check(lib().maua_gram_row_means(_ptr(_f32(f, "f")), _ptr(mean_out), c, f.numel() // c, workspace.data_ptr(),
                                     workspace.numel() * workspace.element_size(), _stream()), "maua_gram_row_means")


def gram_partial(f, center, mean_out, workspace):
    """First half of gram_fwd_mse_ledger: the split-K slabs of F F^T (and the row means) into the layer's own `workspace`."""
    c = f.shape[1] if f.dim() == 4 else f.shape[0]
    hw = f.numel() // c
    _check(lib().maua_gram_partial(_ptr(_f32(f, "f")), _ptr(mean_out) if center else None, c, hw, int(center), workspace.data_ptr(),
                                   workspace.numel() * workspace.element_size(), _stream()), "maua_gram_partial")


class GramFinishBatch:
    """Second half for up to eight layers in one launch.  The argument arrays are built once (device addresses and coefficients of a
    fixed engine plan) and reused by every evaluation."""

    def __init__(self, layers):
        """layers: list of dicts with workspace, gram, target, dmat, c, hw, scale, loss_scale, grad_scale, ledger (one frame's records), slot;
        optional: f (the feature map, for run_partial), mean (covariance form: the array run_partial leaves the row means in before it
        centres the product), slabs (> 0: that many slabs are in the workspace already, conv3x3_image_gram)."""
        n = len(layers)
        self.n = n
        self.keep = layers  # (the tensors must outlive the addresses)
        arr = lambda ct, vals: (ct * n)(*vals)
        self.args = (
            arr(ctypes.c_void_p, [l["workspace"].data_ptr() for l in layers]), arr(ctypes.c_void_p, [_ptr(l["gram"]) for l in layers]),
            arr(ctypes.c_void_p, [_ptr(_f32(l["target"], "target")) for l in layers]), arr(ctypes.c_void_p, [_ptr(l["dmat"]) for l in layers]),
            arr(ctypes.c_int, [int(l["c"]) for l in layers]), arr(ctypes.c_int64, [int(l["hw"]) for l in layers]),
            arr(ctypes.c_float, [float(l["scale"]) for l in layers]), arr(ctypes.c_float, [float(l["loss_scale"]) for l in layers]),
            arr(ctypes.c_float, [float(l["grad_scale"]) for l in layers]), arr(ctypes.c_void_p, [l["ledger"].data_ptr() for l in layers]),
            arr(ctypes.c_int, [int(l["slot"]) for l in layers]), arr(ctypes.c_int, [int(l.get("slabs") or 0) for l in layers]))

        self.partial_args = (
            arr(ctypes.c_void_p, [_ptr(_f32(l["f"], "f")) if l.get("f") is not None else None for l in layers]), self.args[4], self.args[5],
            self.args[0], arr(ctypes.c_size_t, [l["workspace"].numel() * l["workspace"].element_size() for l in layers]),
            arr(ctypes.c_void_p, [_ptr(l["mean"]) if l.get("mean") is not None else None for l in layers]))

    def run_partial(self):
        """maua_gram_partial of every layer (each dict's "f" = its feature map) in at most three launches."""
        a = self.partial_args
        _check(lib().maua_gram_partial_batch(self.n, a[0], a[5], a[1], a[2], a[3], a[4], self.args[11], _stream()), "maua_gram_partial_batch")

    def run(self):
        a = self.args
        _check(lib().maua_gram_finish_mse_batch(self.n, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], _stream()),
               "maua_gram_finish_mse_batch")


def mse_fwd_bwd_ledger(x, target, grad, loss_scale, grad_scale, accumulate, ledger, slot, mask_grad_by_x=False):
    _check(lib().maua_mse_fwd_bwd_ledger(_ptr(_f32(x, "x")), _ptr(_f32(target, "target")), _ptr(grad), x.numel(), float(loss_scale),
                                         float(grad_scale), int(accumulate), int(mask_grad_by_x), ledger.data_ptr(), int(slot),
                                         _stream()), "maua_mse_fwd_bwd_ledger")


def tv_fwd_bwd_ledger(x, grad, strength, accumulate, ledger, slot):
    n, c, h, w = x.shape
    _check(lib().maua_tv_fwd_bwd_ledger(_ptr(_f32(x, "x")), _ptr(grad), n, c, h, w, float(strength), int(accumulate),
                                        ledger.data_ptr(), int(slot), _stream()), "maua_tv_fwd_bwd_ledger")


def loss_ledger_sum(ledger, losses, totals, losses_f64=None):
    """losses: (frames, slots) or (slots,) float32; totals: (frames,) float32; losses_f64 (optional, float64, same shape as
    losses): every filled record's loss before its rounding to fp32."""
    frames, slots = ledger.shape[0], ledger.shape[1]
    if losses_f64 is not None:
        assert losses_f64.dtype == torch.float64 and losses_f64.numel() == frames * slots and losses_f64.is_contiguous()
        _check(lib().maua_loss_ledger_sum_f64(ledger.data_ptr(), frames, slots, _ptr(losses), _ptr(totals), losses_f64.data_ptr(),
                                              _stream()), "maua_loss_ledger_sum_f64")
        return
    _check(lib().maua_loss_ledger_sum(ledger.data_ptr(), frames, slots, _ptr(losses), _ptr(totals), _stream()),
           "maua_loss_ledger_sum")


def gram_bwd(d_sym, f, row_mean, gf, accumulate, workspace=None, relu_mask=None):
    c = d_sym.shape[0]
    hw = f.numel() // c
    workspace = _ws(workspace, 4 * c + 256, f.device)
    _check(lib().maua_gram_bwd(_ptr(_f32(d_sym, "d")), _ptr(f), _ptr(row_mean), _ptr(relu_mask), _ptr(gf), c, hw, int(accumulate),
                               workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()),
           "maua_gram_bwd")
    return gf


def mse_fwd_bwd(x, target, grad, loss_scale, grad_scale, accumulate, loss_out, workspace=None, mask_grad_by_x=False):
    n = x.numel()
    workspace = _ws(workspace, reduce_workspace_bytes(n), x.device)
    _check(lib().maua_mse_fwd_bwd(_ptr(_f32(x, "x")), _ptr(_f32(target, "target")), _ptr(grad), n, float(loss_scale),
                                  float(grad_scale), int(accumulate), int(mask_grad_by_x), _ptr(loss_out),
                                  workspace.data_ptr(),
                                  workspace.numel() * workspace.element_size(), _stream()), "maua_mse_fwd_bwd")
    return loss_out


def mse_weighted_fwd_bwd(x, weights, target, grad, loss_scale, grad_scale, accumulate, loss_out, workspace=None):
    """Temporal ContentLoss: loss = loss_scale * sum((x*w - t)^2), grad (+)= grad_scale * w * (x*w - t)."""
    n, c, h, w = x.shape
    wplanes = weights.numel() // (h * w)
    workspace = _ws(workspace, reduce_workspace_bytes(x.numel()), x.device)
    _check(lib().maua_mse_weighted_fwd_bwd(_ptr(_f32(x, "x")), _ptr(_f32(weights, "weights")), _ptr(_f32(target, "target")),
                                           _ptr(grad), n * c, h * w, wplanes, float(loss_scale), float(grad_scale),
                                           int(accumulate), _ptr(loss_out), workspace.data_ptr(),
                                           workspace.numel() * workspace.element_size(), _stream()),
           "maua_mse_weighted_fwd_bwd")
    return grad


def tv_fwd_bwd(x, grad, strength, accumulate, loss_out, workspace=None):
    n, c, h, w = x.shape
    workspace = _ws(workspace, reduce_workspace_bytes(x.numel()), x.device)
    _check(lib().maua_tv_fwd_bwd(_ptr(_f32(x, "x")), _ptr(grad), n, c, h, w, float(strength), int(accumulate),
                                 _ptr(loss_out), workspace.data_ptr(), workspace.numel() * workspace.element_size(),
                                 _stream()), "maua_tv_fwd_bwd")
    return loss_out


def fill_(x, value):
    _check(lib().maua_fill(_ptr(x), x.numel(), float(value), _stream()), "maua_fill")
    return x


def axpy_(y, x, alpha):
    _check(lib().maua_axpy(_ptr(y), _ptr(x), float(alpha), x.numel(), _stream()), "maua_axpy")
    return y


def sum_small(slots, out):
    _check(lib().maua_sum_small(_ptr(slots), slots.numel(), _ptr(out), _stream()), "maua_sum_small")
    return out


def adam_step(x, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    _check(lib().maua_adam_step(_ptr(x), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), x.numel(), int(step), float(lr),
                                beta1, beta2, eps, _stream()), "maua_adam_step")


# ------------------------------------------------------------------------------------------
# image-space steps between two optimisation runs (csrc/image.hip)
# ------------------------------------------------------------------------------------------
def conv_pack_filters_few_mfma(w):
    """OIHW weights (64, 1-3, 3, 3) -> the bank of conv3x3_few_mfma (bf16 triples in MFMA lane order)."""
    cout, cin = w.shape[:2]
    bank = torch.empty(lib().maua_conv_few_mfma_bank_bytes(), dtype=torch.uint8, device=w.device)
    _check(lib().maua_conv_pack_filters_few_mfma(_ptr(_f32(w, "w")), bank.data_ptr(), cout, cin, _stream()), "maua_conv_pack_filters_few_mfma")
    return bank


def conv_few_mfma_supported(n, cin, h, w, cout, pad):
    return bool(lib().maua_conv_few_mfma_supported(int(n), int(cin), int(h), int(w), int(cout), int(pad)))


def conv3x3_few_mfma(gy, bank, cin, out=None, tile=0):
    """Backward-data of the image layer on the matrix cores: gy (n, 64, h, w) -> (n, cin, h, w).  tile: 0 = the library's choice,
    1 / 2 / 3 = 4 / 8 / 14 output rows x 62 columns per workgroup."""
    n, cout, h, w = gy.shape
    if out is None:
        out = torch.empty(n, cin, h, w, device=gy.device, dtype=torch.float32)
    _check(lib().maua_conv3x3_few_mfma(_ptr(_f32(gy, "gy")), bank.data_ptr(), _ptr(out), n, cin, h, w, cout, int(tile), _stream()), "maua_conv3x3_few_mfma")
    return out


def space_to_depth(x, r, out):
    """out[n][(ry r + rx) C + c][qy][qx] = x[n][c][r qy + ry][r qx + rx] (0 beyond x), C = x.shape[1]."""
    n, c, h, w = x.shape
    assert out.shape[0] == n and out.shape[1] == r * r * c
    _check(lib().maua_space_to_depth(_ptr(_f32(x, "x")), _ptr(out), n, c, r, h, w, out.shape[2], out.shape[3], _stream()), "maua_space_to_depth")
    return out


def depth_to_space(x, r, out, accumulate=False):
    """out[n][c][r qy + ry][r qx + rx] (+)= x[n][(ry r + rx) C + c][qy][qx], C = out.shape[1]; pixels beyond the sites of x: 0."""
    n, cc, qh, qw = x.shape
    c_out = out.shape[1]
    assert cc == r * r * c_out and out.shape[0] == n
    _check(lib().maua_depth_to_space(_ptr(_f32(x, "x")), _ptr(out), n, c_out, r, qh, qw, out.shape[2], out.shape[3], int(accumulate), _stream()),
           "maua_depth_to_space")
    return out


ARRIVAL_COUNTER_BYTES = 4096 * 4
_ARMED = threading.local()


def conv_arm_workspace(workspace, counters=None, zero=True):
    """Arm `workspace` (a device tensor handed to the convolution launches of this host thread) for split channel loops that are finished
    inside the producing launch (maua_conv_arm_workspace): returns the counters tensor (16 KiB, zeroed by the library here and left zeroed
    by every launch) - keep it alive as long as the workspace is in use.  zero=False: only point this thread back at a workspace it armed
    before (nothing is enqueued).  workspace None: disarm."""
    if workspace is None:
        _check(lib().maua_conv_arm_workspace(None, None, 0, 0, _stream()), "maua_conv_arm_workspace")
        _ARMED.pair = None
        return None
    if counters is None:
        counters = torch.empty(ARRIVAL_COUNTER_BYTES, dtype=torch.uint8, device=workspace.device)
        zero = True
    _check(lib().maua_conv_arm_workspace(workspace.data_ptr(), counters.data_ptr(), counters.numel() * counters.element_size(), int(bool(zero)),
                                         _stream()), "maua_conv_arm_workspace")
    _ARMED.pair = (workspace, counters)  # (the library holds raw pointers: both tensors stay alive while this thread is armed with them)
    return counters


def set_split_batch_hint(frames):
    """Frames per launch the job plans with (split-K policy of the convolutions); returns the previous value."""
    prev = lib().maua_get_split_batch_hint()
    lib().maua_set_split_batch_hint(int(frames))
    return prev


def channel_stats(x, noise=None, noise_amp=1e-3, out=None, workspace=None):
    """Raw colour statistics of a batch x (B,3,H,W), optionally jittered by noise_amp * noise (laid out [B][W][H][3], the
    order the reference draws it in): out[B][9] doubles, one row per plane slot j of the reference's reshape (component k =
    plane k * B + j; for B = 1 the three channels): component sums (3) and the upper triangle of the product sums (6)."""
    b, c, h, w = x.shape
    if c != 3:
        raise HipError("channel_stats: 3-channel images")
    if out is None:
        out = torch.empty(b, 9, dtype=torch.float64, device=x.device)
    workspace = _ws(workspace, lib().maua_channel_stats_workspace_bytes(h, w), x.device)
    if noise is not None and noise.numel() != x.numel():
        raise HipError("channel_stats: noise must have the batch's element count")
    for j in range(b):
        _check(lib().maua_channel_stats(_ptr(_f32(x, "x")), _ptr(noise), float(noise_amp), b, j, h, w, _ptr(out[j]),
                                        workspace.data_ptr(), workspace.numel() * workspace.element_size(), _stream()),
               "maua_channel_stats")
    return out


def color_match_solve(stats_t, pixels_t, stats_s, pixels_s, eps, coef_out):
    """coef_out (16 floats) = [M = cov_s^(1/2) cov_t^(-1/2) (9), mean_t (3), mean_s (3), ok]; no host sync."""
    _check(lib().maua_color_match_solve(_ptr(stats_t), stats_t.shape[0], int(pixels_t), _ptr(stats_s), int(pixels_s), float(eps),
                                        _ptr(coef_out), _stream()), "maua_color_match_solve")
    return coef_out


def color_match_apply(x, noise, noise_amp, coef, all_coef, weight, accumulate, out):
    b, c, h, w = x.shape
    for j in range(b):
        _check(lib().maua_color_match_apply(_ptr(_f32(x, "x")), _ptr(noise), float(noise_amp), _ptr(coef), _ptr(all_coef),
                                            all_coef.numel() // 16, float(weight), int(accumulate), b, j, h, w, _ptr(out), _stream()),
               "maua_color_match_apply")
    return out


def resize_bilinear(x, size=None, scale_factor=None):
    """F.interpolate(x, size | scale_factor=..., mode="bilinear", align_corners=False) for (N,C,H,W) device tensors, with
    ATen's conventions: scale_factor form -> output floor(in * s), source step float(1 / s); size form -> in / out."""
    import math
    import numpy as np
    n, c, h, w = x.shape
    if (size is None) == (scale_factor is None):
        raise HipError("resize_bilinear: give exactly one of size / scale_factor")
    if scale_factor is not None:
        oh, ow = int(math.floor(float(h) * scale_factor)), int(math.floor(float(w) * scale_factor))
        sh = sw = float(np.float32(1.0 / scale_factor))
    else:
        oh, ow = (int(size), int(size)) if isinstance(size, int) else (int(size[0]), int(size[1]))
        sh, sw = float(np.float32(h) / np.float32(oh)), float(np.float32(w) / np.float32(ow))
    if oh <= 0 or ow <= 0:
        raise HipError(f"resize_bilinear: output {oh}x{ow}")
    out = torch.empty(n, c, oh, ow, device=x.device, dtype=torch.float32)
    _check(lib().maua_resize_bilinear(_ptr(_f32(x, "x")), _ptr(out), n * c, h, w, oh, ow, sh, sw, _stream()), "maua_resize_bilinear")
    return out


def deprocess_u8(x, mean_bgr):
    """(1,3,H,W) or (3,H,W) network-space BGR image -> (H,W,3) uint8 RGB device tensor (load.deprocess's arithmetic)."""
    x3 = x.reshape(3, x.shape[-2], x.shape[-1])
    _, h, w = x3.shape
    out = torch.empty(h, w, 3, dtype=torch.uint8, device=x.device)
    _check(lib().maua_deprocess_u8(_ptr(_f32(x3, "x")), out.data_ptr(), h, w, float(mean_bgr[0]), float(mean_bgr[1]),
                                   float(mean_bgr[2]), _stream()), "maua_deprocess_u8")
    return out


class LbfgsState:
    """Device-resident L-BFGS state (history slab + bookkeeping) for one flat fp32 vector."""

    def __init__(self, count, history, device):
        self.count, self.history = int(count), int(history)
        if not 0 < self.history <= 254:
            raise HipError(f"--lbfgs_num_correction {history}: the device-side recursion holds at most 254 pairs")
        nbytes = lib().maua_lbfgs_state_bytes(self.count, self.history)
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self._status = torch.zeros(5, dtype=torch.float32, device=device)
        _check(lib().maua_lbfgs_init(self.buf.data_ptr(), nbytes, self.count, self.history, _stream()), "maua_lbfgs_init")

    def reset(self):
        """Back to the state of a new object (no pairs, iteration 0) in the same memory: a captured graph that holds this state's
        address serves the next optimisation problem."""
        _check(lib().maua_lbfgs_init(self.buf.data_ptr(), self.buf.numel(), self.count, self.history, _stream()), "maua_lbfgs_init")

    def iterate(self, x, grad, lr=1.0, tolerance_change=-1.0, tolerance_grad=-1.0, loss=None):
        """One trip of LBFGS.step's loop; `loss` (device scalar of the evaluation that produced `grad`) feeds the
        |loss - prev_loss| < tolerance_change test and may be None."""
        _check(lib().maua_lbfgs_iterate(self.buf.data_ptr(), _ptr(x), _ptr(grad), _ptr(loss), self.count, self.history, float(lr),
                                        float(tolerance_change), float(tolerance_grad), _stream()), "maua_lbfgs_iterate")

    def status(self):
        """Host copy of {n_iter, history_len, stopped, g.d, t} - this one synchronises."""
        _check(lib().maua_lbfgs_status(self.buf.data_ptr(), self.count, self.history, _ptr(self._status), _stream()),
               "maua_lbfgs_status")
        n_iter, hlen, stopped, gtd, t = self._status.tolist()
        return dict(n_iter=int(n_iter), history_len=int(hlen), stopped=bool(stopped), gtd=gtd, t=t)
