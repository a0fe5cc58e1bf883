#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c24; mkdir -p $O
run() { echo "== $*"; ( for kv in "$@"; do export $kv; done; timeout 300 python bench.py --no-cpu-baseline --steps 30 $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e3,1),'k tok/s', round(d['ms_per_step'],3),'ms')" ); }
for ARGS in "--coalesce 0" "--single-rank-rccl --comm abi" ""; do
  echo "#### bench.py $ARGS"
  run VG_LAUNCH_STREAM=normal
  run VG_LAUNCH_STREAM=mask
  run VG_LAUNCH_STREAM=mask VG_LAUNCH_WRAP=1
  run VG_LAUNCH_STREAM=prio VG_LAUNCH_WRAP=1
  run VG_LAUNCH_STREAM=mask DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run VG_LAUNCH_STREAM=normal DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
done 2>&1 | tee $O/wrap.txt
