#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c29; mkdir -p $O
run() { echo "== bench.py $*"; timeout 600 python bench.py --no-cpu-baseline --steps 20 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']/1e3,1),'k tok/s', round(d['ms_per_step'],3),'ms')"; }
run
for g in 1024 512 256; do run --ragged --packed-granule $g; done
for g in 1024 256; do run --ragged --packed-step 1 --packed-granule $g; done
run
