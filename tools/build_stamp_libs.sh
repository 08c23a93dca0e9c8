#!/bin/bash
# The diagnostic builds the *_clock.py tools load (in-kernel s_memtime stamps; tools/profile_round.sh runs those tools): rebuild after any
# change to csrc/ or to the entry points, here in the container (tools/_build/ travels to the GPU box with the snapshot).
set -e
root=$(cd "$(dirname "$0")/.." && pwd); mkdir -p "$root/tools/_build"
# (the last two: conv_x3q's timing-only XQ_NO_SPLIT form - the patch by LDS-DMA as if it came pre-split, wrong numbers - plain and stamped:
#  tools/nosplit_ceiling.py)
for v in "-DXW_STAMP stamp" "-DXQ_STAMP qstamp" "-DLB_STAMP lbstamp" "-DXP_STAMP pstamp" "-DXQ_NO_SPLIT qnosplit" "-DXQ_STAMP@-DXQ_NO_SPLIT qstamp_nosplit"; do
  set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared ${1//@/ } -I "$root/include" -o "$root/tools/_build/libmaua_$2.so" "$root"/maua-style_amd/csrc/*.hip 2>/dev/null &
done
wait; ls -la "$root"/tools/_build/*.so
