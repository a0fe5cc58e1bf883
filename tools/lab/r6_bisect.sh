#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6bisect; mkdir -p $O
T="tests/test_parity_round5_gpu.py::test_decoder_on_the_side_branch_changes_nothing_but_the_schedule"
i=0
for pre in "tests/test_kernels_gpu.py" "tests/test_model_parity_gpu.py" "tests/test_packed_rows_gpu.py tests/test_packed_step_gpu.py" "tests/test_parity_round2_gpu.py tests/test_parity_round3_gpu.py" "tests/test_dp_gpu.py" "tests/test_parity_round4_gpu.py tests/test_parity_round5_gpu.py"; do
  i=$((i+1))
  timeout 900 python -m pytest $pre $T -m gpu -x -q > $O/run$i.txt 2>&1
  echo "== $pre : rc=$? $(grep -c 'Segmentation' $O/run$i.txt) $(tail -1 $O/run$i.txt | cut -c1-80)"
done
