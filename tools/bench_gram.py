"""Microbenchmark of maua_gram_fwd on the VGG-19 style-layer shapes at 1024x1024 and on img_vid's (B C) x (B C) shapes at 512x512."""
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

shapes = [("relu1_1", 64, 1 << 20), ("relu2_1", 128, 1 << 18), ("relu3_1", 256, 1 << 16), ("relu4_1", 512, 1 << 14), ("relu5_1", 512, 1 << 12),
          ("vid B6 relu1_1", 384, 1 << 18), ("vid B6 relu2_1", 768, 1 << 16), ("vid B6 relu3_1", 1536, 1 << 14), ("vid B6 relu4_1", 3072, 1 << 12),
          ("vid B6 relu5_1", 3072, 1 << 10)]
if len(sys.argv) > 1 and sys.argv[1] == "sizes":  # every style layer of VGG-19 / NIN at the image sizes of the scale pyramid
    shapes = []
    for S in (256, 362, 512, 724, 1448, 2048):
        for name, c, d in (("relu2_1", 128, 2), ("relu3_1", 256, 4), ("relu4_1", 512, 8), ("relu5_1", 512, 16)):
            shapes.append((f"{S} {name}", c, (S // d) * (S // d)))
    shapes += [("nin 512 relu3", 256, 63 * 63), ("nin 512 relu5", 256, 31 * 31), ("nin 512 relu7", 384, 15 * 15), ("nin 512 relu10", 1024, 15 * 15),
               ("nin 1024 relu7", 384, 31 * 31), ("nin 1024 relu10", 1024, 31 * 31)]
for name, c, hw in shapes:
    f = torch.relu(torch.randn(1, c, hw, 1, device="cuda"))
    ws = torch.empty(hip.gram_workspace_bytes(c, hw), dtype=torch.uint8, device="cuda")
    out = torch.empty(c, c, device="cuda")
    t = timeit(lambda: hip.gram_fwd(f, 1.0 / (c * hw), False, out=out, workspace=ws))
    gf = 2.0 * c * c * hw / 1e9
    print(f"{name:18s} C={c:5d} HW={hw:8d}  {t:8.1f} us  {gf / t * 1e3:7.1f} TFLOP/s  HBM floor {c * hw * 4 / 6.3e6:6.1f} us", flush=True)
