#!/bin/bash
# The diagnostic builds the *_clock.py tools load (in-kernel s_memtime stamps; tools/profile_round.sh runs those tools): rebuild after any
# change to csrc/ or to the entry points, here in the container (tools/_build/ travels to the GPU box with the snapshot).
set -e
root=$(cd "$(dirname "$0")/.." && pwd); mkdir -p "$root/tools/_build"
for v in "XW_STAMP stamp" "XQ_STAMP qstamp" "LB_STAMP lbstamp" "XP_STAMP pstamp"; do
  set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -fPIC -shared -D$1 -I "$root/include" -o "$root/tools/_build/libmaua_$2.so" "$root"/maua-style_amd/csrc/*.hip 2>/dev/null &
done
wait; ls -la "$root"/tools/_build/*.so
