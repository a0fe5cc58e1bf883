#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c27; mkdir -p $O
for rep in 1 2; do
  for v in cur prio1 prio2; do
    if [ $v = cur ]; then unset VG_LIB; else export VG_LIB=$PWD/tools/lab/lib_attn_$v.so; fi
    echo "== $v"; bash tools/lab/attn_kernels.sh "VG_ATTN_SKIP=20" | grep "us x"
  done
done 2>&1 | tee $O/prio.txt
