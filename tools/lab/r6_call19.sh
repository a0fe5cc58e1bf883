#!/bin/bash
# One-launch dwnorm backward: tests, kernel A/B (cold), step A/B.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c19; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_round6_gpu.py tests/test_packed_step_gpu.py tests/test_kernels_gpu.py -m gpu -x -q -k "dwnorm or conv or packed" 2>&1 | tail -4 | tee $O/test.txt
python tools/lab/dw_bwd_fused_ab.py 2>&1 | grep -v amdgpu | tee $O/kernel_ab.txt
T=640 B=16 python tools/lab/dw_bwd_fused_ab.py 2>&1 | grep -v amdgpu | tee -a $O/kernel_ab.txt
for rep in 1 2 3; do for v in 0 1; do
  echo "== VG_DW_FUSED=$v"
  VG_DW_FUSED=$v timeout 300 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e3,1),'k tok/s', round(d['ms_per_step'],3),'ms', d['roofline']['hbm_kernels'].get('dwnorm_bwd'))"
done; done | tee $O/step_ab.txt
