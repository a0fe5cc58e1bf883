#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6c25; mkdir -p $O
for k in normal mask; do
  ( cd /tmp; export TMPDIR=/tmp VG_LAUNCH_STREAM=$k; timeout 400 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$k -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --coalesce 0 > $O/bench_$k.txt 2>&1 )
  echo "#### $k"; tail -1 $O/bench_$k.txt | cut -c1-120
  python3 $R/tools/lab/gaps_steps.py $O/tr_$k -3 2>&1 | head -40
  rm -rf $O/tr_$k
done
