#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6bisect; mkdir -p $O
ulimit -a > $O/ulimit.txt
export VG_DEBUG_RESOURCES=1 VG_DEBUG_STREAMS=1
timeout 1200 python -m pytest tests/ -m gpu -x -q -s --ignore tests/test_parity_round6_gpu.py > $O/res.txt 2>&1; echo "rc=$? segv=$(grep -c 'Segmentation' $O/res.txt)"
grep "vg_res" $O/res.txt | awk 'NR%12==0' | cut -c1-220 | tail -32
grep "vg_res" $O/res.txt | tail -3 | cut -c1-220
grep "vg_streams" $O/res.txt | tail -2
grep -n "max user processes\|open files\|stack" $O/ulimit.txt
