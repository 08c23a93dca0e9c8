import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip
def t(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for C, HW in ((64, 1 << 20), (128, 1 << 18), (256, 1 << 16), (512, 1 << 14)):
    f = torch.relu(torch.randn(1, C, HW, 1, device="cuda")); d = torch.randn(C, C, device="cuda"); d = d + d.t()
    g = torch.randn(C, HW, device="cuda")
    for name, kw in (("plain", dict(acc=False, mask=None)), ("acc", dict(acc=True, mask=None)), ("acc+mask", dict(acc=True, mask=f))):
        us = t(lambda: hip.gram_bwd(d, f, None, g, kw["acc"], relu_mask=kw["mask"]))
        byt = C * HW * 4 * (2 + (1 if kw["acc"] else 0) + (1 if kw["mask"] is not None else 0))
        print(f"C={C:4d} HW={HW:8d} {name:9s} {us:8.1f} us  {byt/us/1e6:6.2f} TB/s  {2*C*C*HW/us/1e6:6.1f} TF")
