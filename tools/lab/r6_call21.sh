#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c21; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_round6_gpu.py tests/test_packed_step_gpu.py tests/test_kernels_gpu.py tests/test_model_parity_gpu.py -m gpu -x -q 2>&1 | grep "passed\|failed\|Error\|assert" | tail -6 | tee $O/test.txt
for rep in 1 2 3; do for v in 0 1; do
  echo "== VG_DW_FUSED=$v"
  VG_DW_FUSED=$v timeout 300 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e3,1),'k tok/s', round(d['ms_per_step'],3),'ms', d['roofline']['hbm_kernels'].get('dwnorm_bwd'))"
done; done | tee $O/step_ab.txt
VG_MAIN_PRIO=0 timeout 300 python -m pytest tests/test_parity_round6_gpu.py -m gpu -x -q -k "uneven_stream" 2>&1 | grep "passed\|failed\|returncode\|Segmentation" | head -5 | tee $O/rule_off.txt
