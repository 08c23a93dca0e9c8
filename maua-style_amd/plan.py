"""The planner's configuration: every routing switch and threshold of the host side AND of libmaua_hip in one table.

Until round 4 some 45 `MAUA_*` environment variables were read where they were used (engine.py, models.py, optim.py, style.py and `getenv`
calls in csrc/): a benchmark line did not say which of them were in effect, and a stale one changed it silently.  Now

  * `FIELDS` below is the one list of settings, each with its default and what it decides;
  * `get(name)` is the one accessor (host code), `forward_to_library(lib)` hands the library's share to `maua_set_tuning` when the
    library is loaded (csrc reads them with `maua::tuning(name, default)`, no `getenv`);
  * ten environment variables are honoured (`ENV_VARS`): the library path, the distributed-run settings, the host thread count, four
    switches the reference-facing tools and tests use every day - and `MAUA_PLAN="field=value,field=value"`, which sets any field
    (what tools/ab_*.sh use for A/B runs);
  * any OTHER `MAUA_*` variable in the environment is ignored and reported: `env_overrides()` lists what is in effect and what was
    ignored (bench.py prints it in every line as `env_overrides`; hip.lib() warns once on stderr).

Nothing here touches the GPU; the module imports without torch.
"""
import os
import sys

# name -> (default, who reads it, what it decides).  Values are kept as strings (what an environment variable would hold); get_int /
# get_float / on() convert.  "lib" fields are forwarded to libmaua_hip (numbers only).
FIELDS = {
    # ---- kernel families (host routing, models.py)
    "conv_x6": ("1", "host", "3x3 stride-1 layers on the reduced-width matrix cores with fp32-accurate operand splits: 1 both passes, fwd / bwd one pass, 0 fp32 MFMA"),
    "conv_x3": ("1", "host", "1: fp16x3 arithmetic (22 significand bits, conv_x3*.hip); 0: bf16x6 everywhere (the strictly 24-bit route bench.py reports beside the headline)"),
    "conv_x3w": ("1", "host", "layers whose consumed channel count is a multiple of 16 run conv_x3w.hip instead of conv_x3.hip"),
    "conv_x3q": ("256", "host", "smallest consumed channel count from which a 3x3 layer runs conv_x3q.hip (0: never)"),
    "conv_x3p": ("64", "host", "smallest consumed channel count from which a 3x3 layer may run the persistent conv_x3p.hip (0: never); conv_x3p_preferred decides per launch"),
    "conv_x3p_max": ("100000", "host", "largest consumed channel count for conv_x3p.hip"),
    "x3p_gram_min_mb": ("700", "host", "the Gram backward rides in conv_x3p's launch from this many MB of maps per launch (below: conv_x3w's fused form)"),
    "conv_image": ("1", "host", "the 3-channel image layer's forward pass on conv_img.hip"),
    "split_min_produced": ("16", "host", "a 3x3 stride-1 pass runs the split-precision kernels (64-channel tiles) when it PRODUCES at least this many channels, else the fp32-MFMA / direct kernels (VGG-19 / NIN: only the 64 -> 3 image gradient is below; the pruned VGG-16 has widths of 22 and 24)"),
    "x3w_min_pixels": ("4096", "host", "planes smaller than this run conv_x3.hip's 4-row tiles"),
    "few_mfma": ("1", "host", "backward-data of the image layer (64 -> 3 channels) on the matrix cores (conv_few_mfma.hip) instead of conv3x3_few_out's vector-ALU kernel"),
    "strided_fwd_3x3": ("1", "host", "forward pass of a strided, unpadded layer (NIN's stem) as space to depth + a stride-1 3x3 convolution over sites (conv_x3w)"),
    "strided_bwd_3x3": ("1", "host", "backward-data of a strided, unpadded layer (NIN's 11x11 / 4 stem) as a stride-1 3x3 convolution over the output sites + depth to space"),
    # ---- fusions (engine.py)
    "pool_codes": ("1", "host", "2x2 / 2 max pools keep one decision byte per window for the backward pass"),
    "loss_ledger": ("1", "host", "loss partial sums go to a ledger summed once per evaluation"),
    "gram_batch": ("1", "host", "one finishing launch for the Gram / loss chains of all style layers"),
    "gram_partial_batch": ("1", "host", "one partial-product launch for the Gram-form layers"),
    "fuse_pool": ("1", "host", "conv + ReLU + pool in one launch where the pool is the activation's only consumer"),
    "fuse_pool_split": ("1", "host", "... also where the small grid splits the channel loop (the adding pass pools)"),
    "fuse_unpool": ("1", "host", "backward-data staged straight from the pooled map's gradient and the decision bytes"),
    "fuse_gram_max_c": ("256", "host", "style layers of at most this many channels have their Gram backward ride in the next convolution's backward launch (0: never)"),
    "image_gram": ("1", "host", "the image layer's launch leaves the Gram slabs of relu1_1"),
    "finish_in_launch": ("0", "host", "1: the engine arms its split-K workspace - small splits are added up by the last workgroup to arrive at a tile, no finishing launch (bit-identical; measured neutral at 512 x 512: the last arriver's serial chain costs what the launch cost, profiles/probes_r06.md section 1)"),
    "dmat_pack_batch": ("1", "host", "one launch packs the D matrices of all fused style layers"),
    "style_stream": ("auto", "host", "Gram / loss chains of a single image on a side stream: 0, 1, or auto (from 1536 x 1536 pixels)"),
    "side_streams": ("4", "host", "side streams for the per-frame kernels of independent frames (0: none)"),
    # ---- optimiser / job level
    "hip_graph": ("1", "host", "replay iterations from captured hipGraphs"),
    "graph_bundles": ("1", "host", "L-BFGS iterations captured as whole bundles (evaluation + update)"),
    "frame_batch": ("0", "host", "vid_img: frames per batch (0: planned from memory; 1: the reference's frame-by-frame loop)"),
    "frame_cache_mb": ("4096", "host", "vid_img: host cache for decoded frames"),
    "debug_poison": ("0", "host", "tests: every engine buffer starts as NaN"),
    # ---- libmaua_hip (csrc/: maua::tuning)
    "conv_few_out": ("1", "lib", "3x3 passes that produce <= 4 channels run conv3x3_few_out (0: the general MFMA kernel)"),
    "few_out_ks4_below": ("1024", "lib", "conv3x3_few_out: launches of fewer workgroups than this split the channel loop four ways"),
    "x3w_ks": ("0", "lib", "conv_x3w: force this K split (0: cost model)"),
    "x3w_stagger": ("7", "lib", "conv_x3w: start delay of odd-slot workgroups, units of 512 cycles"),
    "finish_in_launch_max_ks": ("4", "lib", "split channel loops of at most this many slabs are finished inside the producing launch when the workspace is armed (conv_x3q, conv_x3w)"),
    "x3q_ks": ("0", "lib", "conv_x3q: force this K split (0: cost model)"),
    "x3q_min_fill": ("0.85", "lib", "conv_x3q_preferred: smallest grid fill x plane cover"),
    "x3q_min_chunks": ("4", "lib", "conv_x3q_preferred: smallest number of 32-channel chunks per workgroup"),
    "x3p_ks": ("0", "lib", "conv_x3p: force this K split (0: cost model)"),
    "cot_inner": ("0", "lib", "conv_x3w / conv_x3q workgroup order: 1 = the channel tiles of a pixel tile start side by side on one XCD (shared patch in its L2), 0 = blockIdx.y is the channel tile"),
    "x3p_order": ("1", "lib", "conv_x3p's work list: 0 channel tile outermost (an XCD's band = part of one channel tile's plane), 1 pixel tile outermost (the channel tiles of a pixel tile side by side on an XCD's CUs: the plane leaves memory once)"),
    "x3p_groups": ("256", "lib", "conv_x3p: workgroups per launch (a multiple of 8)"),
    "x3p_min_fill": ("0.8", "lib", "conv_x3p_preferred: smallest list fill x plane cover"),
    "x3p_min_items": ("512", "lib", "conv_x3p_preferred: smallest number of work items (and a quarter of the smallest items x chunks)"),
    "x6_persist": ("0", "lib", "conv_x6: several tiles per workgroup (experiment)"),
    "gram_x3": ("1", "lib", "Gram partial products in fp16x3 (0: fp32 MFMA)"),
    "gram_bwd_x3": ("1", "lib", "Gram backward in fp16x3 (0: fp32 MFMA)"),
    "gram_t128": ("1", "lib", "128 x 128 two-role Gram blocks: 0 never, 1 whole 64-pixel stages, 2 ragged maps too"),
    "gram_t128_min_hw": ("1024", "lib", "128 x 128 Gram blocks from this many pixels"),
    "p1_order": ("1", "lib", "conv1x1_x3: (pixel tile, channel tile) items in XCD bands, channel tiles adjacent (0: dispatch order)"),
    "lbfgs_vec": ("1", "lib", "lbfgs: 16-byte loads in the history sweeps"),
    "lbfgs_tri": ("1", "lib", "lbfgs: the triangular form of the coefficient kernel"),
}

# The environment variables that are honoured, and the field each one sets (None: read where it is used - library path, distributed run,
# host threads).  Everything else goes through MAUA_PLAN.
ENV_VARS = {
    "MAUA_HIP_LIB": None,
    "MAUA_PLAN": None,
    "MAUA_CONV_X3": "conv_x3",
    "MAUA_CONV_X6": "conv_x6",
    "MAUA_HIP_GRAPH": "hip_graph",
    "MAUA_DEBUG_POISON": "debug_poison",
    "MAUA_DIST_BACKEND": None,
    "MAUA_DIST_TIMEOUT_S": None,
    "MAUA_DIST_JOB_TIMEOUT_S": None,
    "MAUA_HOST_THREADS": None,
}
_FIELD_VAR = {field: var for var, field in ENV_VARS.items() if field is not None}
_TOOL_VARS = {"MAUA_FUZZ_ASSUME_NO_GPU"}  # tools/fuzz_abi_host.py's own switch: not a setting of the product

OVERRIDES = {}  # programmatic overrides (tests: monkeypatch.setitem(plan.OVERRIDES, "fuse_pool", "0")); win over the environment


_plan_cache = (None, {})  # (the MAUA_PLAN string it was parsed from, the parsed fields): get() is called several times per launch


def _planned():
    """MAUA_PLAN's fields, parsed once per distinct value of the variable."""
    global _plan_cache
    text = os.environ.get("MAUA_PLAN")
    if _plan_cache[0] != text or text is None:
        _plan_cache = (text, _parse_plan(text))
    return _plan_cache[1]


def _parse_plan(text):
    out = {}
    for part in (text or "").replace(";", ",").split(","):
        part = part.strip()
        if not part:
            continue
        if "=" not in part:
            raise ValueError(f"MAUA_PLAN: expected field=value, got {part!r}")
        k, v = part.split("=", 1)
        k = k.strip().lower()
        if k.startswith("maua_"):
            k = k[5:]
        if k not in FIELDS:
            raise ValueError(f"MAUA_PLAN: unknown field {k!r} (plan.FIELDS lists them)")
        out[k] = v.strip()
    return out


def get(name):
    """The value of a field as a string: programmatic override, MAUA_PLAN, the field's own environment variable, default."""
    default = FIELDS[name][0]
    if name in OVERRIDES:
        return str(OVERRIDES[name])
    planned = _planned()
    if name in planned:
        return planned[name]
    var = _FIELD_VAR.get(name)
    if var is not None and var in os.environ:
        return os.environ[var]
    return default


def get_int(name):
    return int(float(get(name)))


def get_float(name):
    return float(get(name))


def on(name):
    """Switch fields: anything but "0" is on."""
    return get(name) != "0"


def in_effect():
    """{field: value} for every field whose value is not its default."""
    return {k: get(k) for k in FIELDS if get(k) != FIELDS[k][0]}


def ignored_variables(environ=None):
    """MAUA_* variables in the environment that nothing reads any more (they used to be switches: now they must go through MAUA_PLAN)."""
    environ = os.environ if environ is None else environ
    return sorted(k for k in environ if k.startswith("MAUA_") and k not in ENV_VARS and k not in _TOOL_VARS)


def env_overrides(environ=None):
    """What a benchmark line records: the non-default fields in effect, and the MAUA_* variables that were found and ignored."""
    return {"in_effect": in_effect(), "ignored": ignored_variables(environ)}


_warned = False


def warn_ignored():
    global _warned
    ign = ignored_variables()
    if ign and not _warned:
        _warned = True
        hint = ",".join(f"{k[5:].lower()}={os.environ[k]}" for k in ign if k[5:].lower() in FIELDS)
        sys.stderr.write(f"maua: ignoring {', '.join(ign)} (not among the honoured variables {sorted(ENV_VARS)})"
                         + (f"; use MAUA_PLAN=\"{hint}\"" if hint else "") + "\n")
    return ign


def forward_to_library(lib):
    """Hand the library's fields to maua_set_tuning (every one, so that a programmatic override going back to the default is seen too)."""
    for k, (default, who, _) in FIELDS.items():
        if who == "lib":
            rc = lib.maua_set_tuning(k.encode(), float(get(k)))
            if rc != 0:
                raise RuntimeError(f"libmaua_hip does not know the tuning constant {k!r}")
