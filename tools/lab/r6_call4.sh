#!/bin/bash
# occupancy A/B of the attention kernels: extra dynamic LDS per block -> 3 / 2 / 1 blocks per CU
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c4; mkdir -p $O
for extra in 0 24000 48000 100000; do
  echo "== VG_ATTN_LDS_EXTRA=$extra"
  VG_ATTN_LDS_EXTRA=$extra SHAPES=16x1000 python tools/attn_bench.py 2>&1 | grep "B=16"
  VG_ATTN_LDS_EXTRA=$extra STD=0.3 SHAPES=16x1000 python tools/attn_bench.py 2>&1 | grep "B=16"
done | tee $O/attn_occupancy.txt
