# profiles/probes_r05.md section 6: isolated launches of conv_few_mfma against conv3x3_few_out (us) at five image sizes.
import sys, torch, torch.nn.functional as F
sys.path[:0] = ["/root/repo/maua-style_amd", "/root/repo"]
import hip
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
def timeit(f, reps=30):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
g = torch.Generator(device="cuda").manual_seed(1)
w = torch.randn(64, 3, 3, 3, device="cuda", generator=g) * 0.1
bank = hip.conv_pack_filters_few_mfma(w)
_, wb = hip.conv_pack_filters(w)
for (n, H, W) in [(1, 256, 256), (1, 512, 512), (1, 724, 724), (1, 1024, 1024), (1, 2048, 2048)]:
    gy = torch.randn(n, 64, H, W, device="cuda", generator=g) * (torch.rand(n, 64, H, W, device="cuda", generator=g) > 0.5)
    ref = torch.nn.grad.conv2d_input((n, 3, H, W), w.double(), gy.double(), padding=1) if H <= 1024 else None
    old = hip.conv2d_bwd_data(gy, None, wb, w, (n, 3, H, W), 3, 1, 1)
    line = f"{n}x{H}x{W}: few_out {timeit(lambda: hip.conv2d_bwd_data(gy, None, wb, w, (n, 3, H, W), 3, 1, 1, out=old)):7.1f} us"
    if ref is not None: line += f" (rel {rel(old, ref):.2e})"
    for rows in (0, 1, 2, 3):
        out = torch.full((n, 3, H, W), float("nan"), device="cuda")
        hip.conv3x3_few_mfma(gy, bank, 3, out=out, tile=rows)
        t = timeit(lambda: hip.conv3x3_few_mfma(gy, bank, 3, out=out, tile=rows))
        line += f" | rows {rows}: {t:7.1f} us"
        line += f" rel {rel(out, ref):.2e}" if ref is not None else f" vs old {rel(out, old):.2e}"
    print(line, flush=True)
