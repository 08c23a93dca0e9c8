#!/bin/bash
# Build the Winograd experiment (tools/wino/conv_wino.hip) into tools/_build/libmaua_wino.so, linked against the product library
# for the helpers it takes from there (maua::set_error).
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p "$root/tools/_build"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -o "$root/tools/_build/libmaua_wino.so" "$root/tools/wino/conv_wino.hip" \
  -L"$root/maua-style_amd" -l:libmaua_hip.so -Wl,-rpath,"$root/maua-style_amd"
ls -la "$root/tools/_build/libmaua_wino.so"
