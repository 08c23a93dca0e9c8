import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [REPO, os.path.join(REPO, "maua-style_amd")]
import hip
cin, cout, H = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
x = torch.randn(1, cin, H, H, device="cuda"); w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.05
f6, b6 = hip.conv_pack_filters_x6(w); y = torch.empty(1, cout, H, H, device="cuda")
for _ in range(4):
    hip.conv3x3_x6(x, f6, None, cout, 1, True, out=y)
torch.cuda.synchronize()
