#!/bin/bash
# attention: scattered launch order (VG_ATTN_SCHED=4, lab) against the longest-first order
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c7; mkdir -p $O
for rep in 1 2; do for sc in 0 4; do
  echo "== VG_ATTN_SCHED=$sc"
  VG_ATTN_SCHED=$sc SHAPES=16x1000 python tools/attn_bench.py 2>&1 | grep "B=16"
  VG_ATTN_SCHED=$sc STD=0.3 SHAPES=16x1000,8x2000 python tools/attn_bench.py 2>&1 | grep "B="
done; done | tee $O/attn_sched4.txt
