"""Microbenchmark of maua_conv1x1_x3 on the Gram-backward shapes of VGG-19 (1024 x 1024) and NIN's 1x1 layers (1024 x 1024 image):
MAUA_CONV1X1_WIDE=0 / 1 A/B in two processes.  python tools/bench_conv1x1.py"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("maua-style_amd.hip")

def timeit(f, reps=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

shapes = [("gram bwd relu4_1", 512, 512, 16384, True), ("gram bwd relu5_1", 512, 512, 4096, True), ("gram bwd relu3_1", 256, 256, 65536, True),
          ("nin cccp1 96->96 @254^2", 96, 96, 254 * 254, False), ("nin cccp3 256->256 @127^2", 256, 256, 127 * 127, False),
          ("nin cccp5 384->384 @63^2", 384, 384, 63 * 63, False), ("nin cccp7 1024->1024 @31^2", 1024, 1024, 31 * 31, False),
          ("nin cccp8 1024->1000 @31^2", 1024, 1000, 31 * 31, False), ("vid B6 relu4_1", 3072, 3072, 4096, True)]
for name, cin, cout, hw, acc in shapes:
    x = torch.relu(torch.randn(1, cin, hw, 1, device="cuda"))
    w = torch.randn(cout, cin, device="cuda") * 0.05
    out = torch.zeros(1, cout, hw, 1, device="cuda")
    ws = torch.empty(max(hip.conv1x1_x3_workspace_bytes(1, cin, hw, cout), 16), dtype=torch.uint8, device="cuda")
    t = timeit(lambda: hip.conv1x1_x3(x, w, out=out, accumulate=acc, out_relu_mask=x if acc and cin == cout else None, workspace=ws))
    ref = torch.einsum("oc,cp->op", w.double(), x.reshape(cin, hw).double())
    y = hip.conv1x1_x3(x, w, workspace=ws)
    err = float((y.reshape(cout, hw).double() - ref).norm() / ref.norm())
    gf = 2.0 * cin * cout * hw / 1e9
    print(f"{name:28s} {cin:5d} -> {cout:5d} x {hw:7d}  {t:7.1f} us  {gf / t * 1e3:6.1f} TFLOP/s  HBM floor {(cin + (2 if acc else 1) * cout) * hw * 4 / 6.3e6:6.1f} us  rel err {err:.1e}", flush=True)
