"""BASELINE config 1 end to end through the product's own command line (style.py -> img_img -> optim.optimize ->
libmaua_hip), compared with the same run of the unmodified reference (tools/make_golden.py, group "cli"), and the
flow-less vid_img frame loop."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import GOLDEN, PKG, REPO, rel_l2

pytestmark = pytest.mark.gpu


def run_style(argv, cwd):
    env = dict(os.environ, PYTHONPATH=PKG)
    return subprocess.run([sys.executable, os.path.join(PKG, "style.py")] + argv, cwd=cwd, env=env, capture_output=True,
                          text=True, timeout=1200)


def test_config1_cli_matches_reference(tmp_path, weight_files):
    g = np.load(os.path.join(GOLDEN, "cli_config1.npz"))
    scaling = tmp_path / "scaling.json"
    scaling.write_text(json.dumps({"100000": {"gpu": "0", "multidevice": False}}))
    out = tmp_path / "out"
    out.mkdir()
    r = run_style(["--content", os.path.join(REPO, "tests", "synth_content_256.png"), "--style",
                   os.path.join(REPO, "tests", "synth_style_256.png"), "--image_sizes", "256", "--num_iters", "50",
                   "--model_file", weight_files["vgg19"], "--disable_check", "--scaling_args", str(scaling), "--seed", "0",
                   "--no_hist_match", "--init", "content", "--output_dir", str(out)], cwd=PKG)
    assert r.returncode == 0, r.stderr[-2000:]
    png = out / "synth_content_256_synth_style_256_256.png"  # <output_dir>/<content>_<style>_<size>.png
    assert png.exists()
    import load
    got = load.preprocess(str(png))  # back to network space (quantised to 8 bits, clamped)
    ref64 = torch.from_numpy(g["out_f64"])
    ref32 = torch.from_numpy(g["out_f32"])
    # compare in image space: both reference results pushed through the same 8-bit PNG round trip
    def roundtrip(t):
        p = tmp_path / "rt.png"
        load.deprocess(t.clone()).save(p)
        return load.preprocess(str(p))
    floor = rel_l2(roundtrip(ref32), roundtrip(ref64))
    err = rel_l2(got, roundtrip(ref64))
    assert err <= max(1e-3, 2 * floor), (err, floor)
    # and the golden PNG written by the reference itself decodes to the same thing as its raw tensor round trip
    ref_png = load.preprocess(os.path.join(GOLDEN, "cli_config1_ref.png"))
    assert rel_l2(ref_png, roundtrip(ref32)) <= 1e-6
    # resume-by-file-exists: a second run skips the finished scale and leaves the file untouched
    mtime = png.stat().st_mtime_ns
    r = run_style(["--content", os.path.join(REPO, "tests", "synth_content_256.png"), "--style",
                   os.path.join(REPO, "tests", "synth_style_256.png"), "--image_sizes", "256", "--num_iters", "50",
                   "--model_file", weight_files["vgg19"], "--disable_check", "--scaling_args", str(scaling), "--seed", "0",
                   "--no_hist_match", "--init", "content", "--output_dir", str(out)], cwd=PKG)
    assert r.returncode == 0 and png.stat().st_mtime_ns == mtime


def test_multires_with_histogram_matching_runs(tmp_path, weight_files):
    """Coarse-to-fine 64 -> 128 with the stock colour transfer; checks plumbing and determinism under --seed."""
    scaling = tmp_path / "scaling.json"
    scaling.write_text(json.dumps({"100": {"gpu": "0", "multidevice": False, "optimizer": "lbfgs"},
                                   "100000": {"gpu": "0", "multidevice": False, "optimizer": "adam"}}))
    outs = []
    for k in range(2):
        out = tmp_path / f"out{k}"
        out.mkdir()
        r = run_style(["--content", os.path.join(REPO, "tests", "synth_content_256.png"), "--style",
                       os.path.join(REPO, "tests", "synth_style_256.png"), "--image_sizes", "64,128", "--num_iters", "6,4",
                       "--model_file", weight_files["vgg19"], "--disable_check", "--scaling_args", str(scaling), "--seed",
                       "3", "--output_dir", str(out)], cwd=PKG)
        assert r.returncode == 0, r.stderr[-2000:]
        files = sorted(os.listdir(out))
        assert files == ["synth_content_256_synth_style_256_128.png", "synth_content_256_synth_style_256_64.png"]
        outs.append(np.asarray(Image.open(out / files[0])))
    assert outs[0].shape == (128, 128, 3)
    assert np.array_equal(outs[0], outs[1])


def test_vid_img_frames_without_flow(tmp_path, weight_files):
    import synth
    frames_dir = tmp_path / "clip"
    frames_dir.mkdir()
    fr = synth.frames(3, 48)
    import load
    for i, f in enumerate(fr):
        load.deprocess(f[None].clone()).save(frames_dir / f"{i:05d}.png")
    scaling = tmp_path / "scaling.json"
    scaling.write_text(json.dumps({"100000": {"gpu": "0", "multidevice": False}}))
    out = tmp_path / "out"
    r = run_style(["--transfer_type", "vid_img", "--content", str(frames_dir), "--style",
                   os.path.join(REPO, "tests", "synth_style_256.png"), "--image_sizes", "32,48", "--num_iters", "4,4",
                   "--passes_per_scale", "2", "--init", "content", "--model_file", weight_files["vgg19"], "--disable_check",
                   "--scaling_args", str(scaling), "--seed", "0", "--no_hist_match", "--output_dir", str(out)], cwd=PKG)
    assert r.returncode == 0, r.stderr[-2000:]
    base = out / "clip_synth_style_256"
    for size in ("32", "48"):
        assert sorted(os.listdir(base / size)) == sorted(f"{p}_{i:05d}.png" for p in (1, 2) for i in range(3))
    assert Image.open(base / "48" / "2_00002.png").size == (48, 48)
