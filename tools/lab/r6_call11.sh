#!/bin/bash
# round 6, call 11: attention backward with the additive terms inside the products (fifth k-step) -- parity tests + A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c11; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_round5_gpu.py tests/test_parity_round2_gpu.py tests/test_kernels_gpu.py tests/test_packed_rows_gpu.py -x -q -m gpu -k "attn or attention or window or packed" > $O/pytest_attn.txt 2>&1; echo "rc=$?" >> $O/pytest_attn.txt
tail -8 $O/pytest_attn.txt
for rep in 1 2 3; do for v in cur attn_before_ext; do
  if [ $v = cur ]; then unset VG_LIB; else export VG_LIB=$PWD/tools/lab/lib_$v.so; fi
  echo "== $v"
  SHAPES=16x1000 python tools/attn_bench.py 2>&1 | grep "B="
  STD=0.3 SHAPES=16x1000,8x2000 python tools/attn_bench.py 2>&1 | grep "B="
done; done | tee $O/attn_ab.txt
unset VG_LIB
SHAPES=16x1000 SCALES=0.3 bash tools/lab/attn_kernels.sh "VG_ATTN_SKIP=20" 2>&1 | grep -v amdgpu | tee $O/attn_kernels_cur.txt
