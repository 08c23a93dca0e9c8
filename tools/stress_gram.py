"""Determinism stress of the Gram partial kernels (64 x 64 and 128 x 128 blocks, per-layer and batched launches): the same bits on every
launch, with a convolution on changing data scribbling LDS and registers in between and layers of both kinds side by side on the CUs.
(profiles/probes_r04.md section 2: two builds of the 128 x 128 kernel passed every single-launch test and differed now and then in a
batched launch.)  python tools/stress_gram.py [reps]"""
import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
hip = importlib.import_module("maua-style_amd.hip")
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
torch.manual_seed(0)
w = torch.randn(128, 128, 3, 3, device="cuda") * 0.05
bf, bb, wsc = hip.conv_pack_filters_x3w(w)
sets = {
    "vgg 128": [(64, 16384, False), (128, 4096, False), (256, 1024, False), (512, 256, False), (512, 64, False)],
    "vgg 256": [(64, 65536, False), (128, 16384, False), (256, 4096, False), (512, 1024, False), (512, 256, False)],
    "vgg 362": [(128, 32761, False), (256, 8100, False), (512, 2025, False), (512, 484, False)],
    "vgg 512": [(128, 65536, False), (256, 16384, False), (512, 4096, False), (512, 1024, False)],
    "covariance": [(128, 4131, True), (256, 4096, True), (192, 5000, True), (384, 961, True)],
    "ragged": [(1000, 4200, False), (192, 4097, False), (320, 6000, True), (96, 900, False)],
    "one layer": [(128, 4096, False)],
    "vid": [(768, 16384, False), (1024, 4096, False), (384, 65536, False)],
}
total_bad = 0
for name, shapes in sets.items():
    fs = [torch.relu(torch.randn(1, c, hw, 1, device="cuda")) + (0.5 if cen else 0.0) for c, hw, cen in shapes]
    led = hip.loss_ledger(1, 8, "cuda")
    layers = []
    for k, ((c, hw, cen), f) in enumerate(zip(shapes, fs)):
        layers.append(dict(workspace=torch.zeros(hip.gram_workspace_bytes(c, hw), dtype=torch.uint8, device="cuda"), gram=torch.empty(c, c, device="cuda"),
                           target=torch.zeros(c, c, device="cuda"), dmat=torch.empty(c, c, device="cuda"), c=c, hw=hw, scale=1.0 / (c * hw),
                           loss_scale=0.5 / (c * c), grad_scale=3.0 / (c * c), ledger=led[0], slot=k, f=f, mean=torch.empty(c, device="cuda") if cen else None))
    fin = hip.GramFinishBatch(layers)
    first, bad = None, 0
    for rep in range(reps):
        x = torch.randn(1, 128, 96, 96, device="cuda") * (10.0 ** (rep % 5 - 2))
        hip.conv3x3_x3w(x, bf, wsc, None, 128, 1, True)
        fin.run_partial()
        fin.run()
        res = [l["gram"].clone() for l in layers]
        hip.conv3x3_x3w(x, bf, wsc, None, 128, 1, True)
        res += [hip.gram_fwd(f, 1.0 / (c * hw), cen)[0] for (c, hw, cen), f in zip(shapes, fs)]
        torch.cuda.synchronize()
        if first is None:
            first = res
            for k, (c, hw, cen) in enumerate(shapes):
                ff = fs[k].reshape(c, hw).double()
                ff = ff - ff.mean(1, keepdim=True) if cen else ff
                ref = ff @ ff.t() / (c * hw)
                err = float((res[k].double() - ref).norm() / ref.norm())
                assert err < 2e-5 and torch.equal(res[k], res[len(shapes) + k]), (name, c, hw, cen, err)
        else:
            bad += sum(0 if torch.equal(a, b) else 1 for a, b in zip(res, first))
    print(f"{name:12s} {len(shapes)} layers x {reps} launches (batched + per layer): {bad} results that differ from the first launch's", flush=True)
    total_bad += bad
print("differing results:", total_bad)
sys.exit(1 if total_bad else 0)
