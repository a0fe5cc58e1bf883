#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c26; mkdir -p $O
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt
grep -v "^  File" $O/pytest_gpu.txt | grep "passed\|failed\|rc=\|Segmentation\|real" | tail -5
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee $O/smoke.txt
bash tools/bench_variants.sh $O/variants 2>&1 | tail -22 | tee $O/variants.txt
