"""CPU oracle for the maua-style image-optimisation hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, on the CPU, the arithmetic the
reference performs per iteration (reference optim.py:111-255, loss.py, models.py:351-453
and torch.optim.LBFGS/Adam as configured at optim.py:180-196).  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it, and
only as the checker; the product (`maua-style_amd/`) never imports it and fails loudly
when its HIP library is missing.

Parity status: PINNED.  `tests/test_oracle_golden.py` checks this restatement against
fixtures in `tests/golden/` that were produced by running the unmodified reference in
the build container (`tools/make_golden.py`): per-module losses, total loss, pixel
gradient, style/content targets, and L-BFGS/Adam trajectories in fp32 and fp64 - for
VGG-19, NIN and (round 6) the reference's other VGG stacks, VGG-16 and the channel-pruned
VGG-16, chosen by the checkpoint's name as models.py:248-327 does.

Where the arithmetic lives in a third-party dependency: all FLOPs of the reference are
PyTorch ops (requirements.txt pins torch==1.8.1; this image has torch 2.10).  The oracle
therefore uses the same torch CPU primitives for the dense contractions (conv2d, mm) but
spells out the backward pass, the loss algebra and both optimizers explicitly instead of
using autograd / torch.optim, so it is an independent statement of what the HIP path
must compute.
"""
from .style_oracle import (  # noqa: F401
    LayerSpec,
    OracleNet,
    adam_run,
    build_spec,
    gram_matrix,
    deprocess_u8,
    lbfgs_run,
    match_histogram,
    optimize,
    resize_bilinear,
)
