#!/bin/bash
# Per-kernel times of the attention launches under rocprofv3 for a list of environment settings:
#   bash tools/lab/attn_kernels.sh "VG_ATTN_SKIP=20" "VG_ATTN_SKIP=0" ...      (each argument: space-separated VAR=value)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/attn_kernels; mkdir -p $O
i=0
for cfg in "$@"; do
  i=$((i+1))
  ( cd /tmp; export TMPDIR=/tmp; for kv in $cfg; do export $kv; done
    SHAPES=${SHAPES:-16x1000} SCALES=${SCALES:-0.3} ITERS=10 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/run$i -- python3 $R/tools/lab/attn_window.py > $O/run$i.log 2>&1 )
  echo "== $cfg"
  grep -v amdgpu.ids $O/run$i.log | grep "B=" 
  python3 - <<PY
import csv, glob
f = glob.glob("$O/run$i/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "attn" in r["Name"]:
        print(f"   {float(r['AverageNs'])/1e3:8.1f} us x{r['Calls']:>4}  {r['Name'][:60]}")
PY
  rm -rf $O/run$i
done
