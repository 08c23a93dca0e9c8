"""The image layer alone: conv_img.hip against conv_x6.hip's general kernel on the same input (µs per launch, HBM write rate).
    python tools/bench_image.py [sizes ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "maua-style_amd"))
import torch
import hip

sizes = [int(v) for v in sys.argv[1:]] or [1024, 512, 256]
torch.manual_seed(0)
w = torch.randn(64, 3, 3, 3, device="cuda") * 0.3
b = torch.randn(64, device="cuda")
bank = hip.conv_pack_filters_image(w, b)
bf, _ = hip.conv_pack_filters_x6(w)


def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        e.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(e) * 1e3 / reps)
    return best


for s in sizes:
    x = torch.randn(1, 3, s, s, device="cuda") * 50
    y = torch.empty(1, 64, s, s, device="cuda")
    t_img = timed(lambda: hip.conv3x3_image(x, bank, 64, 1, True, out=y))
    t_x6 = timed(lambda: hip.conv3x3_x6(x, bf, b, 64, 1, True, out=y))
    slabs = torch.empty(hip.conv_image_gram_slabs(s, s, 1), 64, 64, device="cuda")
    t_g = timed(lambda: hip.conv3x3_image_gram(x, bank, 1, y, slabs))
    ws = torch.empty(hip.gram_workspace_bytes(64, s * s), dtype=torch.uint8, device="cuda")
    t_p = timed(lambda: hip.gram_partial(y, False, None, ws))
    print(f"{s}x{s}: conv_img {t_img:7.1f} us ({y.numel() * 4 / t_img / 1e6:.2f} TB/s written)   conv_x6 {t_x6:7.1f} us   "
          f"conv_img with the Gram slabs {t_g:7.1f} us   (separate Gram partial kernel of relu1_1: {t_p:7.1f} us)")
