"""GPU micro-benchmark of the convolution kernels through the C ABI: TFLOP/s per VGG-19 layer shape, fwd and bwd-data.
   python tools/bench_conv.py [S] [reps]        (S = image side, default 1024)"""
import os, sys, time
import torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
only = sys.argv[3] if len(sys.argv) > 3 else ""
layers = [("conv1_1", 3, 64, 1), ("conv1_2", 64, 64, 1), ("conv2_1", 64, 128, 2), ("conv2_2", 128, 128, 2),
          ("conv3_1", 128, 256, 4), ("conv3_2", 256, 256, 4), ("conv4_1", 256, 512, 8), ("conv4_2", 512, 512, 8),
          ("conv5_1", 512, 512, 16)]
tot = {"fwd": 0.0, "bwd": 0.0}
mult = {"conv3_2": 3, "conv4_2": 3}
for name, cin, cout, div in layers:
    if only and only not in name:
        continue
    H = W = S // div
    x = torch.randn(1, cin, H, W, device="cuda")
    w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
    b = torch.randn(cout, device="cuda")
    wf, wb = hip.conv_pack_filters(w)
    y = torch.empty(1, cout, H, W, device="cuda")
    gx = torch.empty(1, cin, H, W, device="cuda")
    gy = torch.randn(1, cout, H, W, device="cuda")
    flops = 2 * 9 * cin * cout * H * W
    for mode in ("fwd", "bwd"):
        def run():
            if mode == "fwd":
                hip.conv2d_fwd(x, wf, b, 3, 1, 1, True, out=y)
            else:
                hip.conv2d_bwd_data(gy, y, wb, w, x.shape, 3, 1, 1, out=gx)
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        tot[mode] += ms * mult.get(name, 1)
        print(f"{name} {mode} {cin:4d}->{cout:4d} @{H:4d}  {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s", flush=True)
print(f"sum over the 13 convs: fwd {tot['fwd']:.3f} ms  bwd {tot['bwd']:.3f} ms  ({(4*sum(2*9*c*o*(S//d)**2*mult.get(n,1) for n,c,o,d in layers))/((tot['fwd']+tot['bwd'])*1e9):.1f} TFLOP/s overall)")
