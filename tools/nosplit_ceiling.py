"""The ceiling of a split-free K loop: conv_x3q as it is against its TIMING-ONLY form -DXQ_NO_SPLIT (the patch arrives as
ready fp16 pairs by LDS-DMA: no dword loads through registers, no maximum, no split, no ds_write - wrong numbers), on the shapes of conv1_2,
conv2_2, conv3_2 and conv4_2 at 1024 x 1024: microseconds per launch (interleaved rounds, one process per library) and the stamped builds'
cycles per chunk and in-kernel clock.
    tools/build_stamp_libs.sh ; python tools/nosplit_ceiling.py [rounds]"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
B = os.path.join(REPO, "tools", "_build")
SHAPES = [("conv1_2", 64, 64, 1024), ("conv2_2", 128, 128, 512), ("conv3_2", 256, 256, 256), ("conv4_2", 512, 512, 128)]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
code = r'''
import os, sys, torch
sys.path[:0] = [%r, %r]
import hip
g = torch.Generator(device="cuda").manual_seed(3)
for name, cin, cout, H in %r:
    x = torch.relu(torch.randn(1, cin, H, H, device="cuda", generator=g))
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * (2.0 / (9 * cin)) ** 0.5
    f, _, wsc = hip.conv_pack_filters_x3q(w)
    y = torch.empty(1, cout, H, H, device="cuda")
    for _ in range(30):
        hip.conv3x3_x3q(x, f, wsc, None, cout, 1, True, out=y)
    ts = []
    for _ in range(%d):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.conv3x3_x3q(x, f, wsc, None, cout, 1, True, out=y)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 50)
    ts.sort()
    print(f"{name} {cin}->{cout}@{H}: median {ts[len(ts) // 2]:.1f} us  min {ts[0]:.1f}  ({2 * 9 * cin * cout * H * H / ts[len(ts) // 2] / 1e6:.0f} TFLOP/s algorithmic)")
''' % (REPO, os.path.join(REPO, "maua-style_amd"), SHAPES, rounds)
for tag, lib in (("product kernel (maua-style_amd/libmaua_hip.so)", None), ("XQ_NO_SPLIT (timing only)", os.path.join(B, "libmaua_qnosplit.so")),
                 ("product kernel again", None)):
    env = dict(os.environ)
    if lib:
        env["MAUA_HIP_LIB"] = lib
    print(f"== {tag}", flush=True)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(out.stdout.strip(), out.stderr.strip()[-400:] if out.returncode else "", flush=True)
for tag, lib in (("stamped product kernel", "libmaua_qstamp.so"), ("stamped XQ_NO_SPLIT", "libmaua_qstamp_nosplit.so")):
    print(f"== {tag}", flush=True)
    for name, cin, cout, H in SHAPES:
        out = subprocess.run([sys.executable, os.path.join(REPO, "tools", "x3q_clock.py"), str(cin), str(cout), str(H)],
                             env=dict(os.environ, MAUA_HIP_LIB=os.path.join(B, lib)), capture_output=True, text=True)
        keep = [l for l in out.stdout.splitlines() if "cycles per chunk" in l or "in-kernel clock" in l or "stamped launch" in l or "K loop per wave" in l]
        print("\n".join(keep) if out.returncode == 0 else out.stderr.strip()[-400:], flush=True)
