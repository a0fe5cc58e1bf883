#!/bin/bash
# forward attention ablations (VG_LAB_ATTN bits; results wrong by construction) at 2 blocks / 1 block per CU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c5; mkdir -p $O
for extra in 0 48000; do
for v in cur lab1 lab2 lab4 lab8 lab16 lab32 lab51 lab64 lab72; do
  if [ $v = cur ]; then unset VG_LIB; else export VG_LIB=$PWD/tools/lab/lib_$v.so; fi
  echo -n "extra=$extra $v: "
  VG_ATTN_LDS_EXTRA=$extra STD=0.3 SHAPES=16x1000 python tools/attn_bench.py 2>&1 | grep "B=16" | cut -c1-40
done; done | tee $O/attn_fwd_ablation.txt
