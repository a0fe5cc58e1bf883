#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6gdb; mkdir -p $O
timeout 1800 /opt/rocm/bin/rocgdb -batch -ex "handle SIGUSR1 nostop noprint pass" -ex "handle SIG34 nostop noprint pass" -ex "run" -ex "bt 12" -ex "info registers" -ex "x/40i \$pc-80" -ex "thread apply all bt 6" --args python -m pytest tests/ -x -q -m gpu -p no:faulthandler > $O/gdb2.txt 2>&1
grep -n "SIGSEGV" $O/gdb2.txt | head -3
grep -c "New Thread" $O/gdb2.txt
tail -5 $O/gdb2.txt | cut -c1-200
