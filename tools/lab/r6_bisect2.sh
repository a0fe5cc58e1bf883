#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6bisect; mkdir -p $O
FILES="tests/test_dp_gpu.py tests/test_kernels_gpu.py tests/test_model_parity_gpu.py tests/test_packed_rows_gpu.py tests/test_packed_step_gpu.py tests/test_parity_round2_gpu.py tests/test_parity_round3_gpu.py tests/test_parity_round4_gpu.py tests/test_parity_round5_gpu.py"
i=0
for cfg in "VG_SMALL_LINEAR=0 VG_DW_FUSED=1" "VG_SMALL_LINEAR=1 VG_DW_FUSED=0"; do
  i=$((i+1))
  ( export $cfg; timeout 900 python -m pytest $FILES -m gpu -x -q > $O/full$i.txt 2>&1; echo "== $cfg : rc=$? segv=$(grep -c 'Segmentation' $O/full$i.txt) $(grep -v '^  File' $O/full$i.txt | tail -1 | cut -c1-80)" )
done
