#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6suite; mkdir -p $O
( time timeout 3000 python -m pytest tests/ -x -q -m gpu ) > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt
tail -15 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -5 | tee $O/smoke.txt
