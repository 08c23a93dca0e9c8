"""Probe (round 6): maua_gram_fwd on channel counts below and beside 64 over large planes (the pruned VGG-16's relu1_1 has 24 channels on the
whole image) - time per call against the time the map takes to leave memory.    python tools/probes_r06/gram_small_c.py"""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
hip = importlib.import_module("maua-style_amd.hip")


def timeit(f, reps=10):
    for _ in range(2):
        f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        f()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for c in (8, 24, 41, 51, 64, 89, 108, 128):
    for hw in (1 << 20, 1 << 22):
        f = torch.relu(torch.randn(1, c, hw, 1, device="cuda"))
        ws = torch.empty(hip.gram_workspace_bytes(c, hw), dtype=torch.uint8, device="cuda")
        out = torch.empty(c, c, device="cuda")
        t = timeit(lambda: hip.gram_fwd(f, 1.0 / (c * hw), False, out=out, workspace=ws))
        want = (f[0, :, :, 0].double() @ f[0, :, :, 0].double().t()) / (c * hw)
        err = float((out.double() - want).norm() / want.norm())
        print(f"C={c:4d} HW={hw:8d}  {t:8.1f} us   map leaves memory in {c * hw * 4 / 6.3e6:6.1f} us   rel err {err:.1e}", flush=True)
