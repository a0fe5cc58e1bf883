#!/bin/bash
# Like variant.sh, but the variant's source is ANOTHER file that replaces <file.hip> of the regular build:
# tools/lab/variant_file.sh <name> <file.hip it replaces> <path of the replacement source> [-DFLAG=...]
set -e
here=$(cd "$(dirname "$0")/../.." && pwd)
csrc=$here/vae-gslm_amd/csrc
name=$1; src=$2; repl=$3; shift 3
obj=/tmp/vg_variant_${name}.o
cp $repl $csrc/_variant_${name}.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $csrc/_variant_${name}.hip -o $obj
rm -f $csrc/_variant_${name}.hip
others=$(ls $csrc/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $here/tools/lab/lib_${name}.so $obj $others
echo built tools/lab/lib_${name}.so
