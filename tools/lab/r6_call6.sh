#!/bin/bash
# backward attention: empty tile body / no DMA builds under rocprofv3 (per-kernel times)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c6; mkdir -p $O
for v in cur lab256 lab512 lab768; do
  if [ $v = cur ]; then unset VG_LIB; else export VG_LIB=$GRAFT_REPO_ROOT/tools/lab/lib_$v.so; fi
  echo "== $v"
  SHAPES=16x1000 SCALES=0.3 bash tools/lab/attn_kernels.sh "VG_ATTN_SKIP=20" 2>&1 | grep -v amdgpu
done | tee $O/attn_bwd_ablation.txt
