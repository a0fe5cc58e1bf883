#!/bin/bash
# the DP GPU test module N times in one call (first-attempt failures are fatal now): VERDICT r05 item 5
cd $GRAFT_REPO_ROOT
N=${N:-15}
O=gpurun_out/r6dp; mkdir -p $O
rm -f gpurun_out/dp_first_attempt_failures.log
for i in $(seq 1 $N); do
  timeout 900 python -m pytest tests/test_dp_gpu.py -q -m gpu -p no:cacheprovider > $O/run_$i.txt 2>&1
  echo "run $i: rc=$? $(tail -1 $O/run_$i.txt)"
done | tee $O/summary.txt
[ -f gpurun_out/dp_first_attempt_failures.log ] && cp gpurun_out/dp_first_attempt_failures.log $O/ || echo "no first-attempt failure in $N runs" | tee -a $O/summary.txt
