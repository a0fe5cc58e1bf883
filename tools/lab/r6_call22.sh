#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c22; mkdir -p $O
run() { echo "== $*"; ( for kv in "$@"; do export $kv; done; timeout 300 python bench.py --no-cpu-baseline --steps 30 $ARGS 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e3,1),'k tok/s', round(d['ms_per_step'],3),'ms')" ); }
for ARGS in "--coalesce 0" "--single-rank-rccl --comm abi"; do
  echo "#### $ARGS"
  run VG_X=1
  run VG_MAIN_PRIO=0
  run VG_DW_FUSED=0
  run VG_SMALL_LINEAR=0
  run VG_MAIN_PRIO=0 VG_DW_FUSED=0 VG_SMALL_LINEAR=0
done 2>&1 | tee $O/regress.txt
