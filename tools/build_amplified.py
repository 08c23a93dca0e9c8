"""Build the library's sources with LLVM's MFMA padding (`-mllvm --amdgpu-mfma-padding-ratio=100`: s_nop between all MFMAs) into
tools/_build/libmaua_pad.so - the AMPLIFIER of profiles/probes_r05.md section 4: it pushes the MFMA-issuing waves of a CU out of step and made
the faulty 128 x 128 Gram form fail in every launch instead of one in 10^5.  Same sources, same per-file flags as the product build
(maua-style_amd/build_native.py); results are bit-identical to the product's (padding changes no arithmetic).
    python tools/build_amplified.py ; MAUA_HIP_LIB=tools/_build/libmaua_pad.so python tools/soak_streams.py"""
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "maua-style_amd"))
import build_native  # noqa: E402

out_dir = os.path.join(REPO, "tools", "_build")
obj_dir = os.path.join(out_dir, "pad_obj")
os.makedirs(obj_dir, exist_ok=True)
lib = os.path.join(out_dir, "libmaua_pad.so")
procs, objs = [], []
for src in build_native._sources():
    obj = os.path.join(obj_dir, os.path.basename(src)[:-4] + ".o")
    objs.append(obj)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "--amdgpu-mfma-padding-ratio=100", "-c", src,
           "-o", obj] + build_native._file_flags(src)
    procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
for src, p in procs:
    out, _ = p.communicate()
    if p.returncode:
        sys.exit(f"hipcc failed on {src}:\n{out}")
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
print(lib)
