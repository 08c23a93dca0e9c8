#!/bin/bash
# A library with other schedule constants of conv_x3p.hip (everything else: the objects of the last build_native.build()):
#   tools/build_x3p_variant.sh NAME "-DXP_W1_PIECES=9 -DXP_EARLY_LOADS=0"   ->   tools/_build/libmaua_NAME.so
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd); mkdir -p "$root/tools/_build"
objs=$(ls "$root"/maua-style_amd/csrc/build/*.o | grep -v conv_x3p.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize $flags -c "$root/maua-style_amd/csrc/conv_x3p.hip" -o "$root/tools/_build/x3p_$name.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/tools/_build/libmaua_$name.so" $objs "$root/tools/_build/x3p_$name.o"
ls -la "$root/tools/_build/libmaua_$name.so"
