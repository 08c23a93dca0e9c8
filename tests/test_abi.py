"""The C-ABI library builds for gfx950, loads without a GPU and exports every symbol include/maua_hip.h declares."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import PKG, REPO

HEADER = os.path.join(REPO, "include", "maua_hip.h")


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(maua_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def libpath():
    import build_native
    return build_native.build()


def test_header_symbols_exported(libpath):
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    exported = set(re.findall(r" T (maua_[a-z0-9_]+)", out))
    want = declared_symbols()
    assert len(want) >= 40
    missing = [s for s in want if s not in exported]
    assert not missing, f"declared in maua_hip.h but not exported: {missing}"
    extra = sorted(exported - set(want))
    assert not extra, f"exported but not declared in maua_hip.h: {extra}"


def test_header_preamble_names_every_setter_with_its_scope(libpath):
    """The header's Conventions paragraph says which state is per host thread and which is process-wide: every
    exported setter (`nm`: a symbol with `_set_` in its name) must be named there, the per-thread one in the per-thread sentence, the
    process-wide ones in the PROCESS-WIDE sentence - and the sources must agree (thread_local for the former, plain statics for the
    latter, no other mutable file-scope state outside the diagnostic *_STAMP builds)."""
    out = subprocess.check_output(["nm", "-D", "--defined-only", libpath], text=True)
    setters = sorted(set(re.findall(r" T (maua_[a-z0-9_]*set_[a-z0-9_]+)", out)))
    assert setters == ["maua_conv_x3p_set_max_groups", "maua_set_split_batch_hint", "maua_set_tuning"], setters
    head = open(HEADER).read().split("#ifndef MAUA_HIP_H")[0]
    per_thread = head[head.index("per host thread"):head.index("PROCESS-WIDE")]
    process_wide = head[head.index("PROCESS-WIDE"):head.index("reductions are fixed-order")]
    assert "maua_set_split_batch_hint" in per_thread and "maua_last_error" in head
    assert "exactly two setters" in process_wide and "maua_set_tuning" in process_wide and "maua_conv_x3p_set_max_groups" in process_wide
    assert "no process-wide mutable state" not in head
    csrc = os.path.join(PKG, "csrc")
    mutable = {}
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".hpp")):
            stamp = 0
            for line in open(os.path.join(csrc, f)):
                if re.match(r"#if(def)? .*STAMP", line):
                    stamp += 1
                elif line.startswith("#endif") and stamp:
                    stamp -= 1
                m = re.match(r"static (thread_local )?([^;=()]*?)\b(g_[a-z_0-9]+)\b(\[[^\]]*\])?\s*(=|;)", line)
                if m and not stamp and "const " not in m.group(2) and "constexpr" not in m.group(2):
                    mutable[m.group(3)] = bool(m.group(1))
    assert mutable.get("g_split_batch_hint") is True and mutable.get("g_err") is True      # per host thread
    assert sorted(k for k, tl in mutable.items() if not tl) == ["g_tuning_set", "g_tuning_values", "g_xp_max_groups"], mutable  # the two documented setters' state


def test_binding_table_matches_header(libpath):
    import hip
    assert sorted(hip.SIGNATURES) == declared_symbols()
    L = hip.lib()  # dlopen + argtypes for every symbol; no compute call
    assert L.maua_abi_version() == 2


def test_host_only_entry_points(libpath):
    import hip
    L = hip.lib()
    # pooling output sizes = ATen's pooling_output_shape (floor and ceil mode, last window must start inside)
    import torch.nn.functional as F
    import torch
    for n in range(2, 40):
        for k, s in ((2, 2), (3, 2), (3, 1), (5, 3)):
            if n < k:
                continue
            for ceil in (False, True):
                want = F.max_pool2d(torch.zeros(1, 1, n, n), k, s, 0, ceil_mode=ceil).shape[-1]
                assert L.maua_pool_out_size(n, k, s, int(ceil)) == want, (n, k, s, ceil)
    assert L.maua_pool_out_size(1, 2, 2, 0) == 0
    assert L.maua_gram_workspace_bytes(64, 1 << 20) > 0
    assert L.maua_gram_workspace_bytes(0, 5) == 0
    assert L.maua_gram_block(512, 1 << 14) == 128 and L.maua_gram_block(64, 1 << 20) == 64 and L.maua_gram_block(512, 256) == 64
    assert L.maua_lbfgs_state_bytes(3 * 64 * 64, 100) > 2 * 101 * 3 * 64 * 64 * 4
    assert L.maua_reduce_workspace_bytes(10) >= 8


def test_invalid_arguments_are_rejected_without_touching_the_gpu(libpath):
    import hip
    L = hip.lib()
    assert L.maua_conv2d_fwd(None, None, None, None, None, 1, 3, 8, 8, 4, 3, 3, 1, 1, 0, 0, None, 0, None) == -1
    assert b"null" in L.maua_last_error()
    assert L.maua_fill(None, 10, 0.0, None) == -1
    assert L.maua_lbfgs_iterate(None, None, None, None, 10, 5, 1.0, -1.0, -1.0, None) == -1


def test_missing_library_fails_loudly(monkeypatch):
    import hip
    monkeypatch.setattr(hip, "_lib", None)
    monkeypatch.setattr(hip, "LIB_PATH", os.path.join(PKG, "no_such_lib.so"))
    with pytest.raises(hip.HipError):
        hip.lib()


def test_cpu_tensors_are_refused(libpath):
    import torch
    import hip
    with pytest.raises(hip.HipError):
        hip.relu_(torch.zeros(4))


def test_host_boundary_under_address_and_ub_sanitizers():
    """SURVEY.md section 5 (sanitizers): the host half of every entry point - argument checks, workspace / bank sizing, split-K
    cost models, grid arithmetic - built with -fsanitize=address,undefined (device code plain: GPU ASan is not available on
    this pool) and driven by a shape fuzzer (tools/fuzz_abi_host.py: extents from -7 to INT_MAX, odd sizes, every kernel
    family) in the GPU-less container.  A sanitizer report aborts the subprocess.  (The first run of this build found the
    signed overflows of `h + 2 * pad - 2` near INT_MAX that common.hpp's conv_dims_ok now rules out before any arithmetic.)"""
    import subprocess
    import sys
    import build_native
    asan = build_native.asan_runtime()
    if asan is None:
        pytest.skip("clang's asan runtime not found under /opt/rocm")
    lib = build_native.build(sanitize=True)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               MAUA_FUZZ_ASSUME_NO_GPU="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    for seed in (0, 1):
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(PKG), "tools", "fuzz_abi_host.py"), lib, "400", str(seed)],
                             capture_output=True, text=True, env=env, timeout=900, cwd="/tmp")
        assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
        assert "no sanitizer report" in out.stdout and "runtime error" not in out.stderr


def test_no_packed_fp32_instruction_in_any_kernel_that_issues_mfma(libpath, tmp_path):
    """profiles/probes_r05.md section 4: a packed fp32 vector instruction (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) can lose its result in
    lanes 48-63 while another wave of the SIMD issues MFMAs (the root cause of round 4's 128 x 128 Gram fault).  Every source with an MFMA
    kernel is compiled without the packed-fp32 target feature (`// hipcc-flags:` line); this disassembles the built library and checks that no
    kernel contains both kinds of instruction."""
    import shutil
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not available")
    lib = shutil.copy(libpath, str(tmp_path / "lib.so"))
    subprocess.run([objdump, "--offloading", lib], cwd=str(tmp_path), capture_output=True, check=True)  # writes lib.so.N.hipv4-...-gfx950
    objs = sorted(p for p in os.listdir(str(tmp_path)) if "gfx950" in p)
    assert objs, "no device code objects found in the library"
    cur, stats = None, {}
    for name in objs:
        text = subprocess.run([objdump, "-d", name], cwd=str(tmp_path), capture_output=True, text=True, check=True).stdout
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = m.group(1)
                stats.setdefault(cur, [0, 0])
            elif cur is not None:
                if re.search(r"\bv_pk_\w+_f32\b", line):
                    stats[cur][0] += 1
                if "v_mfma" in line:
                    stats[cur][1] += 1
    mfma = [k for k, v in stats.items() if v[1]]
    assert len(mfma) >= 60, len(mfma)  # (the scan sees the kernels)
    both = [k for k, v in stats.items() if v[0] and v[1]]
    assert not both, both[:5]
    # Round 6 (tools/soak_streams.py, profiles/probes_r06.md section 2): a kernel WITH packed fp32 instructions beside an MFMA kernel of
    # another stream does lose results.  The kernels that keep them are a closed list - the L-BFGS sweeps, Adam, the bilinear resize: the
    # update kernels run between an evaluation's last join and the next evaluation's first launch (optim.PixelOptimizer._lbfgs_move,
    # tests/test_engine_gpu.py::test_packed_fp32_kernels_never_share_the_gpu_with_matrix_kernels), the resize between two scales.
    packed = sorted(subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
                    for k, v in stats.items() if v[0])
    allowed = ("maua::lbfgs_pair_dots_kernel<", "maua::lbfgs_combine_kernel<", "maua::lbfgs_combine_v4_kernel", "maua::adam_kernel",
               "maua::resize_bilinear_kernel")
    assert packed and all(k.startswith(allowed) for k in packed), [k for k in packed if not k.startswith(allowed)]
