cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/sl
rocprofv3 --kernel-trace --output-format csv -d /tmp/sl -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/sl.log 2>&1
python3 $R/tools/lab/step_listing.py /tmp/sl $R/gpurun_out/step_listing.txt > $R/gpurun_out/step_table_now.txt
head -3 $R/gpurun_out/step_table_now.txt
