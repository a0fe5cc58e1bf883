#!/bin/bash
# Build a variant of the library for A/B runs: tools/lab/variant.sh <name> <file.hip> [-DFLAG=...]
# -> tools/lab/lib_<name>.so (re-uses the objects of the regular build for the other sources)
set -e
here=$(cd "$(dirname "$0")/../.." && pwd)
csrc=$here/vae-gslm_amd/csrc
name=$1; src=$2; shift 2
obj=/tmp/vg_variant_${name}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $csrc/$src -o $obj
others=$(ls $csrc/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $here/tools/lab/lib_${name}.so $obj $others
echo built tools/lab/lib_${name}.so
