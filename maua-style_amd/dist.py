"""Multi-GPU layout: independent frames / images sharded over one process per GPU.

The hot path does not shard inside one image (it would need a halo exchange at all 13 convs and an all-reduce
of five Gram matrices per iteration - SURVEY.md §8e), and the reference has no distributed code at all.  What
shards naturally is the list of independent optimisation problems: video frames without optical flow
(style.vid_img minus flow) or separate images.  Each rank therefore owns a contiguous block of frames and a
full replica of the network; the only communication is ONE broadcast at start-up (conv weights, and the style
Gram targets computed once on rank 0) over RCCL/xGMI - `backend="nccl"` on ROCm - and nothing per iteration.
On CPU-only hosts the same code runs over gloo (tests/test_dist_cpu.py).
"""
import os

import torch
import torch.distributed as td


def env_rank():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


_LONG_GROUP = None  # a second communicator over the same ranks whose collectives may wait for hours (end_of_job_barrier)


def init(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's env).
    Returns (rank, local_rank, world).  One process per GPU; the device is chosen before any collective.

    Two timeouts.  The default group - rendezvous, the FIRST broadcast of a job, the benchmark's barriers - gives up after
    MAUA_DIST_TIMEOUT_S (default 600 s): a rank that died at start-up must not leave the others holding their GPUs.  Every
    collective that ranks reach after uneven amounts of work - vid_img's per-scale weight broadcast from the second scale on,
    `gather_frames`, `end_of_job_barrier` - runs on a second group with MAUA_DIST_JOB_TIMEOUT_S (default one week).  Only rank
    `src` needs a device-resident copy per group: RCCL creates the second communicator lazily at its first collective."""
    global _LONG_GROUP
    rank, local_rank, world = env_rank()
    if world > 1 and not td.is_initialized():
        if backend is None:  # MAUA_DIST_BACKEND=gloo: several ranks sharing one GPU (testing the N > 1 paths on a 1-GPU box)
            backend = os.environ.get("MAUA_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            if local_rank >= torch.cuda.device_count():
                raise RuntimeError(f"rank {rank}: LOCAL_RANK {local_rank} but {torch.cuda.device_count()} device(s) visible; RCCL "
                                   "needs one GPU per rank (MAUA_DIST_BACKEND=gloo shares GPUs between ranks)")
            torch.cuda.set_device(local_rank)
        elif torch.cuda.is_available():
            torch.cuda.set_device(local_rank % torch.cuda.device_count())
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        import datetime
        short = datetime.timedelta(seconds=float(os.environ.get("MAUA_DIST_TIMEOUT_S", 600)))
        td.init_process_group(backend=backend, rank=rank, world_size=world, timeout=short)
        long = datetime.timedelta(seconds=float(os.environ.get("MAUA_DIST_JOB_TIMEOUT_S", 7 * 24 * 3600)))
        _LONG_GROUP = td.new_group(ranks=list(range(world)), timeout=long, backend=backend)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank if local_rank < torch.cuda.device_count() else 0)
    return rank, local_rank, world


def end_of_job_barrier():
    """Where the ranks of a sharded job wait for each other after their (uneven, possibly hours-long) shards: a barrier on the
    long-timeout group.  Everything before it uses the default group's short timeout."""
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        if _LONG_GROUP is not None:
            t = torch.zeros(1, device=_coll_device())
            td.all_reduce(t, group=_LONG_GROUP)
            if t.is_cuda:
                torch.cuda.synchronize()
        else:
            td.barrier()


def shard_range(n_items, rank, world):
    """Contiguous block [lo, hi) of `n_items` owned by `rank`: the first n % world ranks get one extra item."""
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_owner(index, n_items, world):
    for r in range(world):
        lo, hi = shard_range(n_items, r, world)
        if lo <= index < hi:
            return r
    raise IndexError(index)


def job_group():
    """The long-timeout communicator (None outside a multi-rank job): for collectives that ranks reach after uneven amounts of
    work - anything past the start-up phase of a sharded job."""
    return _LONG_GROUP


def broadcast_tensors(tensors, src=0, group=None):
    """Broadcast a list of same-dtype tensors as ONE flat buffer (one collective instead of one per layer).  `group`: the
    communicator (default group = short timeout; `job_group()` for a collective in the middle of a long job)."""
    if not (td.is_available() and td.is_initialized()):
        return  # (a one-rank group still runs the collective: the code path of the 8-GPU job on a 1-GPU box)
    tensors = [t for t in tensors if t is not None and t.numel() > 0]
    if not tensors:
        return
    flat = torch.cat([t.detach().reshape(-1) for t in tensors])
    td.broadcast(flat, src=src, group=group)
    off = 0
    with torch.no_grad():
        for t in tensors:
            n = t.numel()
            t.copy_(flat[off:off + n].reshape(t.shape))
            off += n


def broadcast_network(net, src=0, mid_job=False):
    """Rank `src`'s conv weights/biases -> every rank (51.8 MB for VGG-19 through conv5_1).  `mid_job`: the ranks arrive after
    uneven shards of work (vid_img rebuilds the network per image size; a resumed rank may be a whole scale ahead of rank 0),
    so the collective runs on the long-timeout group - the 10-minute default group is for start-up only."""
    broadcast_tensors([p.data for p in net.parameters()], src, group=_LONG_GROUP if mid_job else None)


class ReplicaWeights:
    """vid_img rebuilds its loss network per image size; the replicas need ONE broadcast per MODEL and job, not one per size.  The kept
    weights are keyed by the model (`key`: the caller passes what selects the checkpoint and the architecture - model file, pooling -; a
    `--scaling_args` table may switch models between sizes, the reference's own config/scaling-img.json goes vgg19 -> prune -> nin).  The
    first call for a key broadcasts rank `src`'s parameters and keeps them, conv layer by conv layer; later calls with that key copy the
    kept tensors into the new network locally - no collective a rank that is a scale ahead or behind could block on.  A later network of the
    same key with MORE conv layers (deeper loss layers at a later size) broadcasts its new tail only; one whose layer shapes do not match
    the kept ones (the key did not tell two architectures apart) drops the entry and broadcasts whole again.  Every rank takes the same
    branch: the decision depends on the key and the layer shapes only, which the ranks share.  The very first broadcast of a job runs on
    the short-timeout start-up group, everything later on the long-timeout one.  Outside a process group `sync` does nothing."""

    def __init__(self):
        self.kept = {}  # key -> [(weight, bias or None)] of the conv layers, in network order
        self.calls = 0

    @staticmethod
    def _convs(net):
        import torch.nn as nn
        return [m for m in net.modules() if isinstance(m, nn.Conv2d)]

    @staticmethod
    def _keep(convs):
        return [(m.weight.data.clone(), None if m.bias is None else m.bias.data.clone()) for m in convs]

    def sync(self, net, src=0, key=None):
        if not (td.is_available() and td.is_initialized()):
            return "single process"
        first_of_job = self.calls == 0
        self.calls += 1
        group = None if first_of_job else _LONG_GROUP
        convs = self._convs(net)
        kept = self.kept.get(key)
        n = 0 if kept is None else min(len(convs), len(kept))
        if kept is not None and any(m.weight.shape != w.shape or (m.bias is None) != (b is None) for m, (w, b) in zip(convs[:n], kept[:n])):
            kept = None  # another architecture under the same key: start over for it
        if kept is None:
            broadcast_tensors([p.data for p in net.parameters()], src, group=group)
            self.kept[key] = self._keep(convs)
            return "broadcast"
        for m, (w, b) in zip(convs[:n], kept[:n]):
            m.weight.data.copy_(w)
            if b is not None:
                m.bias.data.copy_(b)
        if len(convs) > n:  # deeper than anything broadcast so far for this model: the new tail only
            broadcast_tensors([t for m in convs[n:] for t in ([m.weight.data] + ([] if m.bias is None else [m.bias.data]))], src, group=group)
            kept += self._keep(convs[n:])
            return "broadcast of the new tail"
        return "local copy"


def broadcast_style_targets(net, src=0):
    """Style Gram targets captured on rank `src` -> every rank (2.4 MB for the default five layers), so the style
    forward passes run once per job instead of once per rank (and once per frame, as the reference does:
    style.py:178 keeps the hoisting commented out)."""
    if not (td.is_available() and td.is_initialized()):
        return
    for mod in net.style_losses:
        for name in ("target", "video_target"):
            t = getattr(mod, name)
            shape = torch.tensor(list(t.shape) + [0] * (4 - t.dim()), dtype=torch.int64, device=_coll_device())
            td.broadcast(shape, src=src)
            dims = [int(v) for v in shape.tolist() if v > 0]
            if td.get_rank() != src:  # targets live where the network computes (the GPU), whatever the backend moves them with
                t = torch.empty(dims, dtype=torch.float32, device=_compute_device()) if dims else torch.Tensor()
                setattr(mod, name, t)
    broadcast_tensors([getattr(m, n) for m in net.style_losses for n in ("target", "video_target")], src)


def _compute_device():
    return torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")


def _coll_device():
    return torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")


def barrier():
    if td.is_available() and td.is_initialized() and td.get_world_size() > 1:
        td.barrier()


def max_over_ranks(value):
    """max of a python float over all ranks (used for the benchmark's wall time)."""
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device())
    td.all_reduce(t, op=td.ReduceOp.MAX)
    return float(t.item())


def gather_floats(value):
    """python float of every rank, in rank order, on every rank."""
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return [float(value)]
    t = torch.tensor([value], dtype=torch.float64, device=_coll_device())
    out = [torch.zeros_like(t) for _ in range(td.get_world_size())]
    td.all_gather(out, t)
    return [float(v.item()) for v in out]


def group_size():
    """Number of ranks the communicator itself reports (1 outside a process group)."""
    return td.get_world_size() if (td.is_available() and td.is_initialized()) else 1


def backend_name():
    """"nccl" (= RCCL on ROCm), "gloo", or "none" for a single process."""
    return str(td.get_backend()) if (td.is_available() and td.is_initialized()) else "none"


def gather_frames(local, n_items):
    """Collect {index: CPU tensor} dictionaries on rank 0 (final outputs; the reference writes files instead)."""
    if not (td.is_available() and td.is_initialized()) or td.get_world_size() == 1:
        return local
    out = [None] * td.get_world_size()
    td.all_gather_object(out, local, group=_LONG_GROUP)  # the ranks arrive here after their whole shards: long timeout
    merged = {}
    for d in out:
        merged.update(d)
    assert len(merged) == n_items
    return merged
