#!/bin/bash
# Build a variant of the library with several sources recompiled under extra flags:
#   tools/lab/variant_multi.sh <name> "<a.hip b.hip ...>" [-DFLAG=...]   -> tools/lab/lib_<name>.so
set -e
here=$(cd "$(dirname "$0")/../.." && pwd)
csrc=$here/vae-gslm_amd/csrc
name=$1; srcs=$2; shift 2
objs=""; skip=""
for src in $srcs; do
  obj=/tmp/vg_variant_${name}_${src%.hip}.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -w "$@" -c $csrc/$src -o $obj &
  objs="$objs $obj"; skip="$skip -e /${src%.hip}.o"
done
wait
others=$(ls $csrc/build/*.o | grep -v $skip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $here/tools/lab/lib_${name}.so $objs $others
echo built tools/lab/lib_${name}.so
