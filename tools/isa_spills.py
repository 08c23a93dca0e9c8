"""Where a kernel's scratch (spill) traffic sits relative to its MFMAs: python tools/isa_spills.py file.s [substring of the kernel name]
(hipcc -S --cuda-device-only ... -o file.s).  Prints per kernel the scratch loads / stores and the MFMA ordinal in front of each."""
import bisect
import re
import sys

src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else ""
starts = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
for k, (i0, name) in enumerate(starts):
    if want not in name:
        continue
    i1 = starts[k + 1][0] if k + 1 < len(starts) else len(src)
    lines = src[i0:i1]
    end = next((j for j, l in enumerate(lines) if "s_endpgm" in l and "; -- End" in "".join(lines[j:j + 3])), len(lines))
    lines = lines[:end]
    mf = [j for j, l in enumerate(lines) if "v_mfma" in l]
    ld = [j for j, l in enumerate(lines) if "scratch_load" in l]
    st = [j for j, l in enumerate(lines) if "scratch_store" in l]
    rl = [j for j, l in enumerate(lines) if "v_readlane" in l]
    wl = [j for j, l in enumerate(lines) if "v_writelane" in l]
    print(f"{name[-48:]}: {len(lines)} lines, {len(mf)} MFMAs, scratch loads {len(ld)}, stores {len(st)}, readlane {len(rl)}, writelane {len(wl)}")
    print("   loads behind MFMA #:", [bisect.bisect(mf, j) for j in ld])
    print("   stores behind MFMA #:", [bisect.bisect(mf, j) for j in st])
    hist = {}
    for j in rl + wl:
        b = bisect.bisect(mf, j) // 48
        hist[b] = hist.get(b, 0) + 1
    print("   lane moves per 48-MFMA tap:", sorted(hist.items()))
