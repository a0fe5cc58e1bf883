#!/bin/bash
# The launch-stream rule: the new test, the raw reproducer, the exact full-suite command, then the step with a normal- / high-priority compute stream.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c20; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_round6_gpu.py -m gpu -x -q -k "uneven_stream" 2>&1 | tail -15 | tee $O/test_rule.txt
for m in normal high; do echo "== MODE=$m"; MODE=$m timeout 120 python tools/lab/hipgraph_queue_collision.py 2>&1 | grep -v amdgpu | tail -2; done 2>&1 | tee $O/reproducer.txt
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt
grep -v "^  File" $O/pytest_gpu.txt | tail -8
for rep in 1 2 3; do for v in 0 -1; do
  echo "== VG_MAIN_PRIO=$v"
  VG_MAIN_PRIO=$v timeout 300 python bench.py --no-cpu-baseline --steps 60 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e3,1),'k tok/s', round(d['ms_per_step'],3),'ms')"
done; done | tee $O/prio_ab.txt
