#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c3; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_round6_gpu.py -x -q -m gpu > $O/pytest_r6.txt 2>&1; echo "rc=$?" >> $O/pytest_r6.txt
tail -8 $O/pytest_r6.txt
bash tools/lab/run_attnv.sh 2>&1 | grep -v amdgpu | tee $O/attn_noskip.txt
