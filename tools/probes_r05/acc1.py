# profiles/probes_r05.md section 1b: conv_x3p / conv_x3q against fp64 (rel-L2, the fp32 CPU convolution in front) on three layer shapes.
import math, os, sys, torch, torch.nn.functional as F
sys.path[:0] = ["/root/repo", "/root/repo/maua-style_amd"]
import hip
g = torch.Generator(device="cuda").manual_seed(5)
small = torch.empty(16, dtype=torch.uint8, device="cuda")
for cin, cout, H in [(64, 64, 96), (128, 128, 96), (256, 256, 64), (512, 512, 64), (512, 64, 64)]:
    x = torch.relu(torch.randn(1, cin, H, H, device="cuda", generator=g))
    w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g) * math.sqrt(2.0 / (9 * cin))
    fq, bq, wsc = hip.conv_pack_filters_x3q(w)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    ref32 = F.conv2d(x.cpu(), w.cpu(), padding=1).cuda()
    yq = hip.conv3x3_x3q(x, fq, wsc, None, cout, 1, False, workspace=small)
    yp = hip.conv3x3_x3p(x, fq, wsc, None, cout, 1, False, workspace=small)
    rel = lambda a: float((a.double() - ref).norm() / ref.norm())
    print(f"{os.environ.get('TAG','')} {cin}->{cout}@{H}: fp32-CPU {rel(ref32):.3e}  x3q {rel(yq):.3e}  x3p {rel(yp):.3e}")
