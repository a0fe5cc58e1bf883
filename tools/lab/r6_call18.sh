#!/bin/bash
# SmallLinearFn (the few-row fp32 Linears of the diffusion-step embedding): test, then the step with / without, alternating.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c18; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_round6_gpu.py -m gpu -x -q -k "small_linear" 2>&1 | tail -3 | tee $O/test.txt
timeout 900 python -m pytest tests/test_model_parity_gpu.py tests/test_parity_round3_gpu.py -m gpu -x -q 2>&1 | tail -3 | tee -a $O/test.txt
for rep in 1 2 3; do for v in 0 1; do
  echo "== VG_SMALL_LINEAR=$v"
  VG_SMALL_LINEAR=$v timeout 300 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e3,1),'k tok/s', round(d['ms_per_step'],3),'ms', d['roofline'].get('probe_pflops', d['roofline'].get('peak')))"
done; done | tee $O/step_ab.txt
python tools/lab/mm_shapes.py 2>&1 | grep "aten::" | tee $O/mm_shapes_after.txt
