"""Microbenchmark of the fp16x3 1x1 product against the fp32-MFMA route, on the Gram-backward and NIN 1x1 shapes."""
import importlib, sys, os, torch
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

shapes = [("gram_bwd relu1_1", 64, 64, 1024 * 1024), ("gram_bwd relu2_1", 128, 128, 512 * 512), ("gram_bwd relu3_1", 256, 256, 256 * 256),
          ("gram_bwd relu4_1", 512, 512, 128 * 128), ("gram_bwd relu5_1", 512, 512, 64 * 64),
          ("nin cccp1 96", 96, 96, 253 * 253), ("nin cccp3 256", 256, 256, 126 * 126), ("nin cccp5 384", 384, 384, 63 * 63),
          ("nin cccp7 1024", 1024, 1024, 32 * 32), ("nin cccp8 1000", 1024, 1000, 32 * 32)]
for name, cin, cout, hw in shapes:
    x = torch.randn(1, cin, hw, device="cuda"); w = torch.randn(cout, cin, device="cuda") / cin ** 0.5
    y = torch.empty(1, cout, hw, device="cuda")
    ws = torch.empty(max(1, hip.conv1x1_x3_workspace_bytes(1, cin, hw, cout)), dtype=torch.uint8, device="cuda")
    t3 = timeit(lambda: hip.conv1x1_x3(x, w, out=y, workspace=ws))
    ref = torch.einsum("oc,ncp->nop", w.double(), x.double())
    err = float((y.double() - ref).norm() / ref.norm())
    x4 = x.view(1, cin, 1, hw) if hw < 65536 else x.view(1, cin, hw // 1024, 1024) if hw % 1024 == 0 else x.view(1, cin, 1, hw)
    wf = hip.conv_pack_filters(w.view(cout, cin, 1, 1))[0] if hasattr(hip, "conv_pack_filters") else None
    t32 = float("nan")
    if wf is not None:
        y4 = torch.empty((1, cout) + tuple(x4.shape[2:]), device="cuda")
        t32 = timeit(lambda: hip.conv2d_fwd(x4, wf, None, 1, 1, 0, False, out=y4))
    hbm = (cin + cout) * hw * 4 / 1e6
    print(f"{name:22s} x3 {t3:8.1f} us  fp32 {t32:8.1f} us  err {err:.2e}  HBM floor {hbm / 6.3e3 * 1e3:6.1f} us  ({hbm:.0f} MB)", flush=True)
