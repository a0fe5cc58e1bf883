#!/bin/bash
# Backward row-image swizzle (VG_ATTN_BSW 1 vs 0): parity first, then kernel times and LDS conflict counters, one call.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c17; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_round5_gpu.py tests/test_parity_round2_gpu.py tests/test_kernels_gpu.py tests/test_parity_round6_gpu.py tests/test_parity_round4_gpu.py -m gpu -x -q -k "attn or attention or packed or layer or window" 2>&1 | tail -5 | tee $O/parity.txt
for rep in 1 2; do
  for v in bsw0 cur; do
    if [ $v = cur ]; then unset VG_LIB; else export VG_LIB=$PWD/tools/lab/lib_attn_bsw0.so; fi
    echo "== $v"; bash tools/lab/attn_kernels.sh "VG_ATTN_SKIP=20"
  done
done 2>&1 | tee $O/kernels.txt
for v in bsw0 cur; do
  if [ $v = cur ]; then unset VG_LIB; else export VG_LIB=$PWD/tools/lab/lib_attn_bsw0.so; fi
  PMC_GROUPS=0,1 SHAPES=16x1000 SCALES=0.3 ITERS=5 python3 tools/lab/pmc_any.py $O/pmc_$v attn_bwd_dq,attn_bwd_dkv,attn2_fwd -- python3 tools/lab/attn_window.py > $O/pmc_$v.txt 2>&1
  tail -40 $O/pmc_$v.txt
  rm -rf $O/pmc_$v/pass* 
done
