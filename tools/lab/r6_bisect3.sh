#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6bisect; mkdir -p $O
i=0
for cfg in "VG_SMALL_LINEAR=1 VG_DW_FUSED=1" "VG_SMALL_LINEAR=0 VG_DW_FUSED=1" "VG_SMALL_LINEAR=1 VG_DW_FUSED=0"; do
  i=$((i+1))
  ( export $cfg; export VG_DEBUG_STREAMS=1; timeout 900 python -m pytest tests/ -m gpu -x -q -s --deselect tests/test_dp_gpu.py --ignore tests/test_parity_round6_gpu.py > $O/all$i.txt 2>&1; echo "== $cfg : rc=$? segv=$(grep -c 'Segmentation' $O/all$i.txt) $(grep -v '^  File' $O/all$i.txt | tail -1 | cut -c1-80)"; grep "vg_streams" $O/all$i.txt | tail -3 )
done
