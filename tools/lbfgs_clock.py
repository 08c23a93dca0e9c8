"""Where the L-BFGS coefficient kernel spends its cycles: shader-clock stamps at its phase boundaries (diagnostic build):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DLB_STAMP -o tools/_build/libmaua_lbstamp.so maua-style_amd/csrc/*.hip
    python tools/lbfgs_clock.py [N=196608] [HISTORY=100]"""
import ctypes
import os
import sys

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
os.environ.setdefault("MAUA_HIP_LIB", os.path.join(REPO, "tools", "_build", "libmaua_lbstamp.so"))
import hip  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 196608
hist = int(sys.argv[2]) if len(sys.argv) > 2 else 100
L = hip.lib()
L.maua_lbfgs_read_stamps.argtypes = [ctypes.c_void_p]
L.maua_lbfgs_read_stamps.restype = ctypes.c_int
g = torch.Generator(device="cuda").manual_seed(0)
A = 10.0 ** (torch.rand(n, device="cuda", generator=g) * 6 - 3)  # an ill-conditioned quadratic: pairs keep being accepted, the history fills
x = torch.randn(n, device="cuda", generator=g)
st = hip.LbfgsState(n, hist, "cuda")
names = ["stop tests, dots of the pair, commit", "refresh M", "s.y block -> LDS, pair vectors -> registers",
         "first loop | y.y block -> registers", "y.y product", "second loop", "coefficients, g.d, header"]
acc = [0.0] * 7  # intervals between the eight marks
cnt = 0
for it in range(hist + 40):
    grad = A * x
    st.iterate(x, grad, 1.0, -1.0, -1.0, None)
    if it >= hist + 10:
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 16)()
        assert L.maua_lbfgs_read_stamps(buf) == 0
        for k in range(7):
            acc[k] += buf[k + 1] - buf[k]
        cnt += 1
tot = sum(acc) / cnt
print(f"n = {n}, history {hist}: {tot:.0f} shader cycles per launch of lbfgs_coeffs_tri_kernel")
for k in range(7):
    print(f"  {names[k]:42s} {acc[k] / cnt:8.0f} cycles  {100 * acc[k] / cnt / tot:5.1f} %")
print("status", st.status() if hasattr(st, "status") else "")
