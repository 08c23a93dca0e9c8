#!/bin/bash
# Build the library from a git revision's csrc (default HEAD) into tools/_build/libmaua_<name>.so for A/B timing on one box:
#   tools/build_variant.sh NAME [REV] ; then MAUA_HIP_LIB=tools/_build/libmaua_NAME.so python tools/bench_x6_one.py ...
set -e
name=$1; rev=${2:-HEAD}
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$root/tools/_build"
git -C "$root" archive "$rev" maua-style_amd/csrc include | tar -x -C "$tmp"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I "$tmp/include" -o "$root/tools/_build/libmaua_$name.so" "$tmp"/maua-style_amd/csrc/*.hip 2>/dev/null
rm -rf "$tmp"
ls -la "$root/tools/_build/libmaua_$name.so"
