#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6bisect; mkdir -p $O
i=0
for cfg in "VG_X=1" "VG_DW_FUSED=0" "VG_SMALL_LINEAR=0"; do
  i=$((i+1))
  ( export $cfg; timeout 900 python -m pytest tests/ -m gpu -x -q --ignore tests/test_parity_round6_gpu.py > $O/exact$i.txt 2>&1; echo "== $cfg : rc=$? segv=$(grep -c 'Segmentation' $O/exact$i.txt) $(grep -v '^  File' $O/exact$i.txt | grep 'passed\|failed' | tail -1 | cut -c1-80)"; grep "File \"/root" $O/exact$i.txt | head -2 )
done
