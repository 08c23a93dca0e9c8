"""ctypes binding of tools/_build/libmaua_wino.so - the Winograd F(2x2, 3x3) experiment of round 3 (rejected on time:
profiles/probes_r03.md section 1), kept OUTSIDE the product library since round 4.  Build with tools/wino/build.sh;
tools/wino/test_conv_wino_gpu.py and tools/wino/bench_wino.py use this module (run them from the repository root on a GPU box:
`python -m pytest tools/wino/test_conv_wino_gpu.py -q`)."""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(REPO, "maua-style_amd")]
import hip as _hip  # noqa: E402

c_i, c_p, c_sz = ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t
_SIG = {
    "maua_conv_wino_bank_bytes": (c_sz, [c_i, c_i]),
    "maua_conv_pack_filters_wino": (c_i, [c_p, c_p, c_p, c_i, c_i, c_p]),
    "maua_conv_wino_supported": (c_i, [c_i, c_i, c_i, c_i]),
    "maua_conv3x3_wino": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
}
_LIB = None


def _wlib():
    global _LIB
    if _LIB is None:
        path = os.path.join(REPO, "tools", "_build", "libmaua_wino.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run tools/wino/build.sh")
        _hip.lib()  # (error messages and the split-K helpers live in the product library)
        _LIB = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
        for name, (res, args) in _SIG.items():
            fn = getattr(_LIB, name)
            fn.restype, fn.argtypes = res, args
    return _LIB


def conv_pack_filters_wino(w):
    """OIHW 3x3 weights -> (forward bank, backward-data bank) for conv_wino.hip: G g G^T in fp64, pre-split fp16 pairs in MFMA
    lane order, the power-of-two scale in the bank's header (no host synchronisation)."""
    cout, cin = w.shape[:2]
    wc = _hip._f32(w, "w").contiguous()
    bf = torch.empty(_wlib().maua_conv_wino_bank_bytes(cout, cin), dtype=torch.uint8, device=w.device)
    bb = torch.empty(_wlib().maua_conv_wino_bank_bytes(cin, cout), dtype=torch.uint8, device=w.device)
    _hip._check(_wlib().maua_conv_pack_filters_wino(_hip._ptr(wc), bf.data_hip._ptr(), bb.data_hip._ptr(), cout, cin, _hip._stream()),
           "maua_conv_pack_filters_wino")
    return bf, bb


def conv_wino_supported(cin, h, w, pad):
    return bool(_wlib().maua_conv_wino_supported(int(cin), int(h), int(w), int(pad)))


def conv3x3_wino(x, bank, bias, cout, pad, relu, out=None, out_relu_mask=None, accumulate=False):
    n, cin, h, w = x.shape
    if out is None:
        out = torch.empty(n, cout, h + 2 * pad - 2, w + 2 * pad - 2, device=x.device, dtype=torch.float32)
    _hip._check(_wlib().maua_conv3x3_wino(_hip._ptr(_hip._f32(x, "x")), bank.data_hip._ptr(), _hip._ptr(bias), _hip._ptr(out_relu_mask), _hip._ptr(out), n, cin, h, w,
                                   cout, pad, int(relu), int(accumulate), _hip._stream()), "maua_conv3x3_wino")
    return out


