"""The fp32 oracle against the fp64 oracle on NIN (config 5) at 128 / 256 / 301 px: how far fp32 rounding alone moves the gradient when a
ReLU / max-pool decision flips (CPU only; cited by tests/test_strided_as_3x3_gpu.py)."""
import sys, torch
sys.path[:0] = ["/root/repo/maua-style_amd", "/root/repo", "/root/repo/tests"]
import synth
from conftest import NIN_LAYERS, make_cfg
from oracle.style_oracle import OracleNet, build_spec
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for S in (128, 256, 301):
    content, style, init = synth.images(S)
    cfg = make_cfg(use_covariance=True, **NIN_LAYERS)
    res = {}
    for dt in (torch.float64, torch.float32):
        onet = OracleNet(build_spec(cfg), synth.nin_state_dict(), dt)
        onet.capture_content(content); onet.capture_style([style], cfg.style_blend_weights)
        t, _, g = onet.feval(init)
        res[dt] = (float(t), g)
    print(S, "total rel", abs(res[torch.float32][0] - res[torch.float64][0]) / abs(res[torch.float64][0]), "grad rel", rel(res[torch.float32][1], res[torch.float64][1]))
