"""Microbenchmark of the fp16x3 5x5 convolution on NIN's conv2 geometry (96 -> 256 channels at 126 x 126), both passes."""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("maua-style_amd.hip")

def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3

cin, cout, H = 96, 256, 126
x = torch.randn(1, cin, H, H, device="cuda"); w = torch.randn(cout, cin, 5, 5, device="cuda") * 0.03
gy = torch.randn(1, cout, H, H, device="cuda")
bf, bb, wsc = hip.conv_pack_filters_kxk_x3(w)
y = torch.empty(1, cout, H, H, device="cuda"); gx = torch.empty(1, cin, H, H, device="cuda")
ws = torch.empty(max(1, hip.conv_kxk_x3_workspace_bytes(1, cout, H, H, cin, 5, 2)), dtype=torch.uint8, device="cuda")
gf = 2.0 * 25 * cin * cout * H * H / 1e9
t = timeit(lambda: hip.conv_kxk_x3(x, bf, wsc, None, cout, 5, 2, True, out=y))
print(f"fwd {t:7.1f} us  {gf / t * 1e3:6.1f} TFLOP/s algorithmic ({3 * gf / t * 1e3:6.1f} fp16)")
t = timeit(lambda: hip.conv_kxk_x3(gy, bb, wsc, None, cin, 5, 2, False, out=gx, workspace=ws))
print(f"bwd {t:7.1f} us  {gf / t * 1e3:6.1f} TFLOP/s algorithmic ({3 * gf / t * 1e3:6.1f} fp16)")
