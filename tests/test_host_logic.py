"""Host-side behaviour of the drop-in surface (config / models.assemble / optim.set_model_args / lbfgs_moves),
checked against dumps taken from the unmodified reference (tests/golden/host_logic.json).  No GPU needed."""
import argparse
import json
import os
import types

import pytest
import torch

from conftest import GOLDEN, PKG

HOST = json.load(open(os.path.join(GOLDEN, "host_logic.json")))


def normalise(d):
    out = {}
    for k, v in d.items():
        out[k] = v if isinstance(v, (int, float, str, bool, list, dict, type(None))) else repr(v)
    return out


@pytest.mark.parametrize("tag", ["defaults", "lists", "gpu_multi", "gpu_c_multi", "load_args_vid"])
def test_get_args_matches_reference(tag, monkeypatch):
    import config
    monkeypatch.chdir(PKG)  # default --ffmpeg_args / --scaling_args / --load_args are relative paths, as in the reference
    want = HOST[f"args_{tag}"]
    if "__error__" in want:  # the reference raised on this command line: same exception type, same message
        with pytest.raises(eval(want["__error__"])) as e:
            config.get_args(HOST[f"argv_{tag}"])
        assert str(e.value) == want["__msg__"]
        return
    got = normalise(vars(config.get_args(HOST[f"argv_{tag}"])))
    if tag == "load_args_vid":
        # the preset file is this build's own (single MI355X instead of 2x11 GB): compare everything that does not come
        # from file-only keys of the reference preset
        for k in ("scaling_file", "model_file", "gpu", "multidevice", "backward_device", "dtype"):
            want.pop(k, None)
            got.pop(k, None)
    assert set(got) == set(want), (sorted(set(got) ^ set(want)))
    for k in want:
        assert got[k] == want[k], (k, got[k], want[k])


def test_flag_types_quirks():
    import config
    p = config.build_parser()
    a = p.parse_args(["--content", "c", "--style", "s", "--lbfgs_tolerance_change", "3"])
    assert a.lbfgs_tolerance_change == 3 and isinstance(a.lbfgs_tolerance_change, int)  # type=int in the reference
    assert a.style == ["s"] and a.learning_rate == 1 and a.video_style_factor == 100
    with pytest.raises(SystemExit):
        p.parse_args(["--content", "c", "--style", "s", "--lbfgs_tolerance_grad", "0.5"])


def test_postprocess_assertions(monkeypatch):
    import config
    monkeypatch.chdir(PKG)
    with pytest.raises(AssertionError):
        config.get_args(["--content", "c", "--style", "s", "--image_sizes", "1,2", "--num_iters", "3"])
    with pytest.raises(AssertionError):
        config.get_args(["--content", "c", "--style", "s", "--style_blend_weights", "1,2"])
    with pytest.raises(ValueError):
        config.get_args(["--content", "c", "--style", "s", "--gpu", "c", "--backend", "mkldnn"])


def test_set_model_args_matches_reference_table():
    """The decision rule on the reference's own stock table (values copied into the test as data)."""
    import optim
    stock = {"1456": {"model_file": "vgg19", "optimizer": "lbfgs", "multidevice": False, "gpu": "0"},
             "2448": {"model_file": "vgg19", "optimizer": "adam", "multidevice": False, "gpu": "0"},
             "2656": {"model_file": "vgg19", "optimizer": "adam", "multidevice": True, "gpu": "0,1"},
             "3760": {"model_file": "prune", "optimizer": "adam", "multidevice": False, "gpu": "0"},
             "4096": {"model_file": "prune", "optimizer": "adam", "multidevice": True, "gpu": "0,1"},
             "5312": {"model_file": "nin", "style_layers": "relu1,relu3,relu5,relu7,relu9,relu11", "content_layers": "relu8",
                      "optimizer": "adam", "multidevice": False, "gpu": "0"},
             "6896": {"model_file": "nin", "style_layers": "relu1,relu3,relu5,relu7,relu9,relu11", "content_layers": "relu8",
                      "optimizer": "adam", "multidevice": True, "gpu": "0,1"}}
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".json", delete=False) as f:
        json.dump(stock, f)
    for key, want in HOST["set_model_args"].items():
        size, gpus = key.split("|")
        a = types.SimpleNamespace(scaling_args=f.name, gpu=gpus, model_file="vgg19", optimizer="lbfgs", multidevice=False)
        optim.set_model_args(a, int(size))
        got = {k: v for k, v in vars(a).items() if k != "scaling_args"}
        assert got == want, (key, got, want)


def test_shipped_scaling_table_keeps_the_optimizer_schedule(monkeypatch):
    """BASELINE config 3: sizes <= 1456 run L-BFGS, larger ones Adam, whatever --optimizer says."""
    import optim
    monkeypatch.chdir(PKG)
    for size, opt in ((256, "lbfgs"), (512, "lbfgs"), (1024, "lbfgs"), (1456, "lbfgs"), (2048, "adam"), (4096, "adam")):
        a = types.SimpleNamespace(scaling_args="config/scaling-img.json", gpu="0", optimizer="adam" if opt == "lbfgs" else "lbfgs")
        optim.set_model_args(a, size)
        assert a.optimizer == opt and a.model_file == "vgg19" and a.gpu == "0"


def _args(**over):
    d = dict(content_layers="relu4_2", style_layers="relu1_1,relu2_1,relu3_1,relu4_1,relu5_1", tv_weight=1e-3,
             temporal_weight=50.0, content_weight=5.0, style_weight=100.0, use_covariance=False, normalize_gradients=True,
             video_style_factor=100.0, shift_factor=0.0, verbose=False)
    d.update(over)
    return argparse.Namespace(**d)


NETS = {
    "default": ({}, "vgg19"),
    "no_tv_temporal": (dict(tv_weight=0.0, temporal_weight=0.0), "vgg19"),
    "conv_named": (dict(content_layers="conv2_2", style_layers="conv1_1,relu3_1"), "vgg19"),
    "deep": (dict(content_layers="relu5_2", style_layers="relu5_4"), "vgg19"),
    "nin": (dict(style_layers="relu1,relu3,relu5,relu7,relu9,relu11", content_layers="relu8"), "nin"),
}


@pytest.mark.parametrize("tag", list(NETS))
def test_assembled_network_matches_reference(tag):
    import models
    over, arch = NETS[tag]
    if arch == "vgg19":
        cnn, table = models.VGG(models.build_sequential(models.channel_list["VGG-19"], "max")), models.vgg19_dict
    else:
        cnn, table = models.NIN("max"), models.nin_dict
    net, losses = models.assemble(cnn.features, table, _args(**over))
    want = HOST["nets"][tag]
    mods = list(net)
    assert [type(m).__name__ for m in mods] == [m["type"] for m in want["modules"]]
    for m, w in zip(mods, want["modules"]):
        assert getattr(m, "name", None) == w["name"]
        if w["type"] == "Conv2d":
            assert (m.in_channels, m.out_channels, list(m.kernel_size), list(m.stride), list(m.padding)) == \
                (w["cin"], w["cout"], w["k"], w["stride"], w["pad"])
        if w["type"] in ("MaxPool2d", "AvgPool2d"):
            assert models._as_int(m.kernel_size) == models._as_int(w["k"]) and models._as_int(m.stride) == models._as_int(w["stride"])
            assert bool(m.ceil_mode) == bool(w["ceil"])
        if "strength" in w:
            assert m.strength == w["strength"] and getattr(m, "normalize", None) == w["normalize"]
    assert [m.name for m in losses] == want["losses"]
    assert [m.name for m in net.content_losses] == want["content"]
    assert [m.name for m in net.style_losses] == want["style"]
    assert [m.name for m in net.tv_losses] == want["tv"]
    assert [m.name for m in net.temporal_losses] == want["temporal"]
    assert all(not p.requires_grad for p in net.parameters())


def test_layer_name_tables():
    import models
    assert models.vgg19_dict["C"][:3] == ["conv1_1", "conv1_2", "conv2_1"] and len(models.vgg19_dict["C"]) == 16
    assert models.vgg19_dict["R"][-1] == "relu5_4" and models.vgg19_dict["P"] == [f"pool{i}" for i in range(1, 6)]
    assert len(models.vgg16_dict["C"]) == 13 and models.nin_dict["C"][9] == "conv4-1024"
    assert models.channel_list["VGG-19"].count("P") == 5


def test_select_model_errors(tmp_path):
    import models
    with pytest.raises(ValueError):
        models.select_model("resnet50", "max", False, True)
    with pytest.raises(ValueError):
        models.build_sequential(models.channel_list["VGG-19"], "median")
    with pytest.raises(FileNotFoundError):
        models.select_model(str(tmp_path / "vgg19_missing.pth"), "max", False, True)
    # strict loading complains about missing feature keys, --disable_check does not
    import synth
    sd = synth.vgg19_state_dict()
    sd.pop("features.0.bias")
    f = tmp_path / "vgg19_partial.pth"
    torch.save(sd, f)
    with pytest.raises(RuntimeError):
        models.select_model(str(f), "max", False, False)
    cnn, table = models.select_model(str(f), "max", False, True)
    assert table is models.vgg19_dict and len(list(cnn.features)) == 37


def test_lbfgs_and_adam_step_counts_match_reference():
    import optim
    for n, fevals in HOST["lbfgs_fevals"].items():
        n = int(n)
        # torch re-evaluates after every move except the last one of a step() call; moves = evaluations that are used
        moves = optim.lbfgs_moves(n)
        assert moves == (2 if n == 1 else (n - 1 if n in (2, 3) else n))
        assert fevals == (2 if n == 1 else n)
    assert optim.lbfgs_moves(0) == 0


def test_cpu_mode_and_multidevice_are_refused(tmp_path):
    import models
    import synth
    f = tmp_path / "vgg19_synth.pth"
    torch.save(synth.vgg19_state_dict(), f)
    a = _args(model_file=str(f), pooling="max", disable_check=True, gpu="c", multidevice=False)
    with pytest.raises(RuntimeError):
        models.load_model(a)
    a.gpu, a.multidevice = "0,1", True
    with pytest.raises(NotImplementedError):
        models.load_model(a)


def test_oracle_match_histogram_against_reference_fixture():
    """The CPU oracle of utils.match_histogram (reference utils.py:88-151) against outputs of the reference itself
    (tools/make_golden.py::gen_hist, torch.symeig shimmed by linalg.eigh): single images against one / two sources in both
    modes, a 3-frame clip against a 4-frame and a 1-frame source (the reference's batch reshape mixes frames and channels -
    reproduced), and a strongly correlated image.  The device path is checked against the same fixture in
    tests/test_kernels_gpu.py."""
    import numpy as np
    from oracle import match_histogram
    g = np.load(os.path.join(GOLDEN, "match_histogram.npz"))
    target, src1, src2 = (torch.from_numpy(g[k]) for k in ("target", "src1", "src2"))
    for tag, srcs in (("one", [src1]), ("two", [src1, src2])):
        torch.manual_seed(1234)
        out = match_histogram(target.clone(), srcs, mode=True)
        assert torch.allclose(out, torch.from_numpy(g[f"out_{tag}"]), rtol=1e-5, atol=1e-4)
        torch.manual_seed(1234)
        out = match_histogram(target.clone(), srcs, mode="avg")
        assert torch.allclose(out, torch.from_numpy(g[f"out_avg_{tag}"]), rtol=1e-5, atol=1e-4)
    assert torch.equal(match_histogram(target.clone(), [src1], mode=False), target)
    clip, vsrc = torch.from_numpy(g["clip"]), torch.from_numpy(g["vsrc"])
    for tag, mode in (("avg", "avg"), ("rand", True)):
        torch.manual_seed(77)
        np.random.seed(5)
        out = match_histogram(clip.clone(), [vsrc, src2], mode=mode)
        assert torch.allclose(out, torch.from_numpy(g[f"out_clip_{tag}"]), rtol=1e-5, atol=1e-4), tag
    torch.manual_seed(99)
    out = match_histogram(torch.from_numpy(g["big"]).clone(), [src1], mode=True)
    assert torch.allclose(out, torch.from_numpy(g["out_big"]), rtol=1e-4, atol=1e-3)
    # the transfer does what it says: channel means of the result equal the source's
    torch.manual_seed(0)
    out = match_histogram(target.clone(), [src1], mode=True)
    assert torch.allclose(out.mean((0, 2, 3)), src1.mean((0, 2, 3)), atol=0.05)
    # a non-finite input takes the reference's `except RuntimeError` exit: the untouched image comes back
    bad = target.clone()
    bad[0, 1, 3, 3] = float("nan")
    res = match_histogram(bad.clone(), [src1], mode=True)
    assert torch.equal(torch.isnan(res), torch.isnan(bad)) and torch.equal(res[~torch.isnan(res)], bad[~torch.isnan(bad)])


def test_oracle_resize_matches_reference_fixture():
    import numpy as np
    from oracle import resize_bilinear
    g = np.load(os.path.join(GOLDEN, "resize_bilinear.npz"))
    img = torch.from_numpy(g["img"])
    for k in range(4):
        assert torch.equal(resize_bilinear(img, scale_factor=float(g[f"sf_{k}"])), torch.from_numpy(g[f"out_sf_{k}"]))
        assert torch.equal(resize_bilinear(img, size=tuple(int(v) for v in g[f"hw_{k}"])), torch.from_numpy(g[f"out_hw_{k}"]))


def test_video_windows_schedule_matches_reference_formula():
    """optim.py:113-123: ceil(T / window) + 1 window starts per sequence, spaced over (frames - window / 2); single images
    always start at 0.  Values below were produced by the reference's expression for the same shapes."""
    import torch

    import optim
    init, video, image = torch.zeros(5, 1), torch.zeros(7, 1), torch.zeros(1, 1)
    assert optim.video_windows(init, [video, image], 3) == [[0, 2, 4], [0, 3, 6], [0, 0, 0]]
    init, video = torch.zeros(40, 1), torch.zeros(25, 1)
    w = optim.video_windows(init, [video], 18)
    assert w == [[0, 11, 21, 31], [0, 6, 11, 16]]


def test_bench_refuses_more_ranks_than_devices(tmp_path):
    """`bench.py --gpus N` starts N ranks itself; with fewer than N devices visible (none here) and no gloo override it
    must fail loudly instead of silently measuring one GPU."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MAUA_DIST_BACKEND")}
    out = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--steps", "1"], capture_output=True,
                         text=True, env=env, timeout=300, cwd=str(tmp_path))
    if torch.cuda.device_count() < 2:
        assert out.returncode == 2 and "device(s) visible" in out.stderr and not out.stdout.strip()


def test_cpu_baseline_builds_the_model_it_is_asked_for(monkeypatch):
    """bench.cpu_baseline(model="nin") must time the NIN + covariance oracle (a local named `model` used to shadow
    the argument, so `--model nin` reported the VGG-19 oracle's time as its CPU baseline)."""
    import importlib
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    monkeypatch.syspath_prepend(repo)
    bench = importlib.import_module("bench")
    import oracle
    seen = {}
    real = oracle.build_spec

    def spy(cfg):
        seen["model_file"], seen["cov"], seen["style"] = cfg.model_file, cfg.use_covariance, cfg.style_layers
        return real(cfg)
    monkeypatch.setattr(oracle, "build_spec", spy)
    threads = torch.get_num_threads()
    try:
        r = bench.cpu_baseline(96, "lbfgs", iters=1, repeats=1, model="nin")
    finally:
        torch.set_num_threads(threads)
    assert seen == {"model_file": "nin", "cov": True, "style": "relu1,relu3,relu5,relu7,relu9,relu11"}
    assert r["model"] == "nin" and r["kind"] == "port" and r["value"] > 0
    assert r["cpu_model"] is None or "nin" not in str(r["cpu_model"]).lower()


# ---------------------------------------------------------------------------------------------------------------------------------
# The planner configuration (plan.py): one table of settings, ten honoured environment variables, everything else reported
# ---------------------------------------------------------------------------------------------------------------------------------
def test_plan_reports_unknown_and_stale_maua_variables(monkeypatch):
    """A MAUA_* variable nothing reads (a stale switch of an earlier round, a typo) is ignored AND reported: bench.py prints
    plan.env_overrides() in every line (`extra.env_overrides`), hip.lib() warns once."""
    import plan
    for k in list(os.environ):
        if k.startswith("MAUA_"):
            monkeypatch.delenv(k)
    assert plan.env_overrides() == {"in_effect": {}, "ignored": []}
    monkeypatch.setenv("MAUA_GRAM_T128", "0")          # a switch of rounds 3-4: now a planner field, not an environment variable
    monkeypatch.setenv("MAUA_NO_SUCH_THING", "1")
    monkeypatch.setenv("MAUA_CONV_X3", "0")            # one of the ten honoured variables
    rep = plan.env_overrides()
    assert rep["ignored"] == ["MAUA_GRAM_T128", "MAUA_NO_SUCH_THING"]
    assert rep["in_effect"] == {"conv_x3": "0"} and plan.get("gram_t128") == "1"
    monkeypatch.setenv("MAUA_PLAN", "gram_t128=0, x3p_min_items=1024")
    assert plan.get("gram_t128") == "0" and plan.get_int("x3p_min_items") == 1024
    assert plan.env_overrides()["in_effect"] == {"conv_x3": "0", "gram_t128": "0", "x3p_min_items": "1024"}
    monkeypatch.setenv("MAUA_PLAN", "no_such_field=1")
    with pytest.raises(ValueError):
        plan.get("gram_t128")
    monkeypatch.delenv("MAUA_PLAN")
    monkeypatch.setitem(plan.OVERRIDES, "fuse_pool", "0")  # what the GPU tests use
    assert not plan.on("fuse_pool") and plan.env_overrides()["in_effect"]["fuse_pool"] == "0"
    assert len(plan.ENV_VARS) <= 10


def test_plan_library_fields_reach_the_library(monkeypatch):
    """Every "lib" field of plan.FIELDS is a tuning constant libmaua_hip knows (maua_set_tuning), and a value set through MAUA_PLAN is
    what the library reads back - csrc/ has no getenv left."""
    import glob
    import plan
    import hip
    L = hip.lib()
    for k, (default, who, _) in plan.FIELDS.items():
        if who == "lib":
            assert L.maua_set_tuning(k.encode(), float(default)) == 0, k
    assert L.maua_set_tuning(b"no_such_constant", 1.0) != 0
    monkeypatch.setenv("MAUA_PLAN", "x3q_min_fill=0.5")
    hip.apply_plan()
    assert L.maua_get_tuning(b"x3q_min_fill", -1.0) == 0.5
    monkeypatch.delenv("MAUA_PLAN")
    hip.apply_plan()
    assert L.maua_get_tuning(b"x3q_min_fill", -1.0) == 0.85
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "maua-style_amd", "csrc")
    for f in glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.hpp")):
        assert "getenv(" not in open(f).read().replace("`getenv`", ""), f


def test_integration_md_lists_exactly_the_planner_fields():
    """INTEGRATION.md's planner table is generated from plan.FIELDS (tools/update_plan_table.py): a field added without re-running the tool,
    or a default changed, fails here."""
    import re
    import plan
    text = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "INTEGRATION.md")).read()
    head = "| field | default | read by | decides |\n|---|---|---|---|\n"
    body = text[text.index(head) + len(head):]
    body = body[:body.index("\n\n")]
    rows = [re.match(r"\| `([a-z0-9_]+)` \| `([^`]*)` \| (host|library) \|", line) for line in body.splitlines()]
    assert all(rows), [line for line, m in zip(body.splitlines(), rows) if not m]
    listed = {m.group(1): (m.group(2), m.group(3)) for m in rows}
    want = {k: (d, "library" if who == "lib" else "host") for k, (d, who, _) in plan.FIELDS.items()}
    assert listed == want
    for var in plan.ENV_VARS:
        assert f"`{var}`" in text, var


def test_plan_cache_follows_the_environment(monkeypatch):
    """plan.get() parses MAUA_PLAN once per distinct value of the variable (it is asked several times per launch): a changed,
    removed or malformed value is seen at the next call, programmatic overrides still win."""
    import plan
    monkeypatch.delenv("MAUA_PLAN", raising=False)
    assert plan.get("x3p_min_items") == plan.FIELDS["x3p_min_items"][0]
    monkeypatch.setenv("MAUA_PLAN", "x3p_min_items=1024")
    assert plan.get("x3p_min_items") == "1024" and plan.get("x3p_min_items") == "1024"
    monkeypatch.setenv("MAUA_PLAN", "x3p_min_items=256;gram_t128=2")
    assert plan.get("x3p_min_items") == "256" and plan.get("gram_t128") == "2"
    monkeypatch.setitem(plan.OVERRIDES, "x3p_min_items", "64")
    assert plan.get("x3p_min_items") == "64"
    monkeypatch.setenv("MAUA_PLAN", "nonsense")
    with pytest.raises(ValueError):
        plan.get("gram_t128")
    monkeypatch.delenv("MAUA_PLAN")
    assert plan.get("gram_t128") == plan.FIELDS["gram_t128"][0]
    # the honoured per-field variables still work, and MAUA_PLAN wins over them
    monkeypatch.setenv("MAUA_CONV_X3", "0")
    assert plan.get("conv_x3") == "0"
    monkeypatch.setenv("MAUA_PLAN", "conv_x3=1")
    assert plan.get("conv_x3") == "1"


def test_bench_line_refuses_a_stopped_optimiser():
    """bench.lbfgs_still_moving: the timed region's rate is only valid if every iteration did its whole work - a raised stop flag makes the
    update kernels return early.  The helper reports {n_iter, stopped, history_len} and raises when a stop rule fired
    or an iteration is missing."""
    import bench

    class State:
        def __init__(self, **kw):
            self.kw = kw

        def status(self):
            return self.kw

    class Opt:
        pass
    opt = Opt()
    opt.state = State(n_iter=305, history_len=100, stopped=False, gtd=-1.0, t=1.0)
    assert bench.lbfgs_still_moving(opt, 305, "the test") == {"n_iter": 305, "stopped": False, "history_len": 100, "expected_n_iter": 305}
    opt.state = State(n_iter=305, history_len=100, stopped=True, gtd=1.0, t=1.0)
    with pytest.raises(RuntimeError, match="stopped moving"):
        bench.lbfgs_still_moving(opt, 305, "the test")
    opt.state = State(n_iter=290, history_len=100, stopped=False, gtd=-1.0, t=1.0)
    with pytest.raises(RuntimeError):
        bench.lbfgs_still_moving(opt, 305, "the test")
