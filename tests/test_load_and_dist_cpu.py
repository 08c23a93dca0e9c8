"""CPU-side pieces around the hot path: image pre/post-processing against the reference's (torchvision semantics),
and the frame-sharding / broadcast layer over gloo with world_size 2."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import GOLDEN, PKG, REPO


def test_preprocess_deprocess_match_reference():
    import load
    g = np.load(os.path.join(GOLDEN, "cli_config1.npz"))
    import synth
    pre = load.preprocess(os.path.join(REPO, "tests", "synth_content_256.png"))
    assert pre.shape == (1, 3, 256, 256) and pre.dtype == torch.float32
    np.testing.assert_allclose(np.array(synth.checksum(pre)), g["pre_content_checksum"], rtol=1e-12)
    np.testing.assert_array_equal(pre[0, :, :4, :4].numpy(), g["pre_content_corner"])
    out = np.asarray(load.deprocess(torch.from_numpy(g["deprocess_probe_in"]).clone()))
    np.testing.assert_array_equal(out, g["deprocess_probe_out"])  # clamp + 8-bit truncation identical


def test_save_naming(tmp_path):
    import argparse
    import load
    args = argparse.Namespace(output=str(tmp_path / "c_s"), original_colors=False, content=None)
    t = torch.zeros(1, 3, 8, 8)
    load.save_tensor_to_file(t, args)
    load.save_tensor_to_file(t, args, size=256)
    load.save_tensor_to_file(t, args, iteration=7, size=256)
    assert sorted(os.listdir(tmp_path)) == ["c_s.png", "c_s_256.png", "c_s_256_7.png"]


def test_shard_ranges_cover_everything():
    import dist
    for n in (1, 7, 8, 64, 65):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                lo, hi = dist.shard_range(n, r, world)
                assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
                seen += list(range(lo, hi))
            assert seen == list(range(n))
            for i in range(n):
                lo, hi = dist.shard_range(n, dist.shard_owner(i, n, world), world)
                assert lo <= i < hi


def _worker(rank, world, port, tmp):
    sys.path[:0] = [REPO, PKG]
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import argparse
    import dist
    import models
    import synth
    r, _, w = dist.init(backend="gloo")
    assert (r, w) == (rank, world)
    # every rank builds the same architecture; only rank 0 has the real weights
    cnn = models.VGG(models.build_sequential(models.channel_list["VGG-19"][:5], "max"))
    sd = synth.vgg19_state_dict()
    if rank == 0:
        cnn.load_state_dict({k: v for k, v in sd.items() if k in cnn.state_dict()}, strict=False)
    else:
        for p in cnn.parameters():
            p.data.fill_(float("nan"))
    args = argparse.Namespace(content_layers="relu2_1", style_layers="relu1_1,relu2_1", tv_weight=1e-3, temporal_weight=0.0,
                              content_weight=5.0, style_weight=100.0, use_covariance=False, normalize_gradients=True,
                              video_style_factor=100.0, shift_factor=0.0, verbose=False)
    net, losses = models.assemble(cnn.features, models.vgg19_dict, args)
    dist.broadcast_network(net, src=0)
    params = list(net.parameters())  # conv1_1, conv1_2, conv2_1 weights and biases, in order
    want = [sd[f"features.{i}.{n}"] for i in (0, 2, 5) for n in ("weight", "bias")]
    assert len(params) == len(want) and all(torch.equal(a, b) for a, b in zip(params, want))
    # style targets exist on rank 0 only, then on every rank
    if rank == 0:
        for i, m in enumerate(net.style_losses):
            m.target = torch.full((4 + i, 4 + i), 1.5 + i)
            m.video_target = m.target.clone() * 2
    dist.broadcast_style_targets(net, src=0)
    for i, m in enumerate(net.style_losses):
        assert m.target.shape == (4 + i, 4 + i) and float(m.target[0, 0]) == 1.5 + i
        assert float(m.video_target[1, 1]) == 2 * (1.5 + i)
    # frames are block-partitioned; results gathered on every rank
    n_frames = 5
    lo, hi = dist.shard_range(n_frames, rank, world)
    local = {i: torch.full((2,), float(i)) for i in range(lo, hi)}
    merged = dist.gather_frames(local, n_frames)
    assert sorted(merged) == list(range(n_frames)) and float(merged[4][0]) == 4.0
    assert dist.max_over_ranks(float(rank)) == float(world - 1)
    # vid_img's per-size networks: ONE broadcast per job (dist.ReplicaWeights), later sizes copy the kept weights locally
    def fresh(n_layers, layers):
        c = models.VGG(models.build_sequential(models.channel_list["VGG-19"][:n_layers], "max"))
        if rank == 0:
            c.load_state_dict({k: v for k, v in sd.items() if k in c.state_dict()}, strict=False)
        else:
            for p_ in c.parameters():
                p_.data.fill_(float("nan"))
        a2 = argparse.Namespace(**{**vars(args), "content_layers": layers, "style_layers": layers})
        return models.assemble(c.features, models.vgg19_dict, a2)[0]
    rep = dist.ReplicaWeights()
    want3 = [sd[f"features.{i}.{n}"] for i in (0, 2) for n in ("weight", "bias")]
    net1 = fresh(3, "relu1_2")
    assert rep.sync(net1) == "broadcast" and all(torch.equal(a, b) for a, b in zip(net1.parameters(), want3))
    net2 = fresh(3, "relu1_1")                                   # a shallower network at the next size: no collective at all
    assert rep.sync(net2) == "local copy" and all(torch.equal(a, b) for a, b in zip(net2.parameters(), want3[:2]))
    net3 = fresh(5, "relu2_1")                                   # a deeper one: only its new tail travels
    assert rep.sync(net3) == "broadcast of the new tail" and all(torch.equal(a, b) for a, b in zip(net3.parameters(), want))
    # a --scaling_args table that changes the model between sizes (the reference's config/scaling-img.json: vgg19 -> prune -> nin): the
    # second model is ANOTHER checkpoint of an architecture whose first layers have the same shapes - it must get its own broadcast, not
    # the first model's kept weights, and coming back to the first model is a local copy again
    def other(n_layers, layers, scale):
        c = models.VGG(models.build_sequential(models.channel_list["VGG-19"][:n_layers], "max"))
        for p_ in c.parameters():
            p_.data.fill_(scale if rank == 0 else float("nan"))
        a2 = argparse.Namespace(**{**vars(args), "content_layers": layers, "style_layers": layers})
        return models.assemble(c.features, models.vgg19_dict, a2)[0]
    rep = dist.ReplicaWeights()
    assert rep.sync(fresh(3, "relu1_2"), key=("vgg19.pth", "max")) == "broadcast"
    net_b = other(3, "relu1_2", 0.25)
    assert rep.sync(net_b, key=("prune.pth", "max")) == "broadcast" and all(float(p_.min()) == float(p_.max()) == 0.25 for p_ in net_b.parameters())
    net_a = fresh(3, "relu1_2")
    assert rep.sync(net_a, key=("vgg19.pth", "max")) == "local copy" and all(torch.equal(a, b) for a, b in zip(net_a.parameters(), want3))
    # the same key with other layer shapes (a key that did not tell two architectures apart): whole broadcast again, no exception
    nin_like = torch.nn.Sequential(torch.nn.Conv2d(3, 8, 5), torch.nn.Conv2d(8, 8, 1))
    for p_ in nin_like.parameters():
        p_.data.fill_(0.5 if rank == 0 else float("nan"))
    assert rep.sync(nin_like, key=("vgg19.pth", "max")) == "broadcast" and all(float(p_.min()) == 0.5 for p_ in nin_like.parameters())
    dist.barrier()
    open(os.path.join(tmp, f"ok{rank}"), "w").write("ok")


def test_replica_weights_do_nothing_outside_a_process_group():
    """vid_img calls ReplicaWeights.sync for every size, single-process runs included: no copy, no exception, whatever the networks are."""
    import dist
    rep = dist.ReplicaWeights()
    a, b = torch.nn.Conv2d(3, 8, 3), torch.nn.Conv2d(3, 16, 5)
    wa = a.weight.data.clone()
    assert rep.sync(a, key="m1") == "single process" and rep.sync(b, key="m1") == "single process"
    assert torch.equal(a.weight.data, wa) and not rep.kept


def test_two_rank_gloo_broadcast_and_sharding(tmp_path):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]


def test_eight_rank_gloo_broadcast_and_sharding(tmp_path):
    """The same worker at the world size of the target node (8 ranks, gloo on the CPU: one process per would-be GPU): rank 0's weights and
    style targets reach all eight, five frames shard as 1 + 1 + 1 + 1 + 1 + 0 + 0 + 0 and gather whole, max-over-ranks, the replica
    weights' per-model bookkeeping.  No RCCL run with N > 1 has ever been possible here: this is the N = 8 control
    flow, without the GPU."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(8, port, str(tmp_path)), nprocs=8, join=True)
    assert sorted(os.listdir(tmp_path)) == [f"ok{r}" for r in range(8)]


def test_flow_warp_map_matches_reference_fixture(tmp_path):
    """load.flow_warp_map / read_flo / write_flow against the reference's own reader and writer (tests/golden/flow_warp_map.npz,
    tools/make_golden.py::gen_temporal)."""
    import numpy as np
    import torch.nn.functional as F
    import load
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "flow_warp_map.npz"))
    path = str(tmp_path / "a.flo")
    g["flo_file_bytes"].tofile(path)
    assert np.array_equal(load.read_flo(path), g["flow"])
    for key in ("warp_40x56", "warp_64x80"):
        hh, ww = (int(v) for v in key.split("_")[1].split("x"))
        got = load.flow_warp_map(path, (hh, ww))
        assert got.shape == (1, hh, ww, 2)
        assert float((got - torch.from_numpy(g[key])).abs().max()) <= 1e-6
    warped = F.grid_sample(torch.from_numpy(g["image"]), load.flow_warp_map(path, (64, 80)), padding_mode="border",
                           align_corners=False)
    assert float((warped - torch.from_numpy(g["warped"])).abs().max()) <= 1e-3
    again = str(tmp_path / "b.flo")
    load.write_flow(g["flow"], again)
    assert np.array_equal(np.fromfile(again, dtype=np.uint8), g["flo_file_bytes"])
    with open(str(tmp_path / "bad.flo"), "wb") as f:
        f.write(b"\0" * 32)
    try:
        load.read_flo(str(tmp_path / "bad.flo"))
        raise AssertionError("bad magic accepted")
    except ValueError:
        pass


def test_original_colors_matches_reference_fixture():
    """--original_colors: Y of the generated image, CbCr of the (resized) content image (reference load.py:236-240)."""
    from PIL import Image
    import load
    g = np.load(os.path.join(GOLDEN, "cli_config1.npz"))
    out = load.original_colors(Image.fromarray(g["origcol_content"]), Image.fromarray(g["origcol_generated"]))
    assert np.array_equal(np.asarray(out), g["origcol_out"])
