#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c23; mkdir -p $O
for m in normal high mask; do echo "== MODE=$m"; MODE=$m timeout 120 python tools/lab/hipgraph_queue_collision.py 2>&1 | grep -v amdgpu | tail -2; done 2>&1 | tee $O/reproducer.txt
for k in mask normal; do echo "== regression test, VG_LAUNCH_STREAM=$k"; VG_LAUNCH_STREAM=$k timeout 300 python -m pytest tests/test_parity_round6_gpu.py -m gpu -x -q -k "uneven_stream" 2>&1 | grep "passed\|failed\|returncode" | head -3 | cut -c1-200; done 2>&1 | tee $O/test_rule.txt
run() { echo "== $*"; ( for kv in "$@"; do export $kv; done; timeout 300 python bench.py --no-cpu-baseline --steps 30 $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e3,1),'k tok/s', round(d['ms_per_step'],3),'ms')" ); }
for ARGS in "" "--coalesce 0" "--single-rank-rccl --comm abi" "--single-rank-rccl"; do
  echo "#### bench.py $ARGS"
  for rep in 1 2; do
  run VG_LAUNCH_STREAM=normal
  run VG_LAUNCH_STREAM=mask
  done
  run VG_LAUNCH_STREAM=prio
done 2>&1 | tee $O/kinds.txt
