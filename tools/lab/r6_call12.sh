#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c12; mkdir -p $O
python tools/lab/attn_order_sweep.py 2>&1 | grep -v amdgpu | tee $O/attn_order_sweep.txt
python tools/lab/attn_order_sweep.py 2>&1 | grep -v amdgpu | tee -a $O/attn_order_sweep.txt
