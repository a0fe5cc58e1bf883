#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c14; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_round6_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "conv_block or bottleneck" > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt; tail -3 $O/pytest.txt
for rep in 1 2; do for v in "" 1; do
  echo "== VG_ATTN_V1=$v"
  VG_ATTN_V1=$v SHAPES=16x1000,16x640,8x2000 python tools/attn_bench.py 2>&1 | grep "B="
  VG_ATTN_V1=$v STD=0.3 SHAPES=16x1000 python tools/attn_bench.py 2>&1 | grep "B="
done; done | tee $O/attn_v1_vs_v2.txt
