"""conv_x3w start stagger sweep (planner field x3w_stagger, handed to the library when it is loaded: MAUA_PLAN="x3w_stagger=.."): one layer, several values, one process each."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys, torch
sys.path[:0] = [%r, %r]
import hip
for cin, cout, H in ((512, 512, 128), (256, 256, 256), (64, 64, 1024), (128, 128, 512)):
    x = torch.relu(torch.randn(1, cin, H, H, device="cuda"))
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    fw, bw, wsc = hip.conv_pack_filters_x3w(w)
    y = torch.empty(1, cout, H, H, device="cuda")
    for _ in range(10):
        hip.conv3x3_x3w(x, fw, wsc, None, cout, 1, True, out=y)
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.conv3x3_x3w(x, fw, wsc, None, cout, 1, True, out=y)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 100)
    print(f"  {cin}->{cout}@{H}: {sorted(ts)[3]:.1f} us", end="")
print()
''' % (REPO, os.path.join(REPO, "maua-style_amd"))
for st in sys.argv[1:]:
    env = dict(os.environ, MAUA_PLAN=",".join(p for p in (os.environ.get("MAUA_PLAN", ""), f"x3w_stagger={st}") if p))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(f"stagger {st:>3}:", out.stdout.strip(), out.stderr.strip()[-300:] if out.returncode else "")
