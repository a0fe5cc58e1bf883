# Lab: --ragged with the Transformer stack packed (A) against the whole step packed (B): per-kernel difference per step
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/ra /tmp/rb
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ra -- python3 $R/bench.py --ragged --packed-step 0 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/ra.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rb -- python3 $R/bench.py --ragged --packed-step 1 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/rb.log 2>&1
python3 $R/tools/lab/diff_kstats.py /tmp/ra /tmp/rb 1 > $R/gpurun_out/ragged_ab.txt
tail -1 /tmp/ra.log | cut -c1-200 >> $R/gpurun_out/ragged_ab.txt
tail -1 /tmp/rb.log | cut -c1-200 >> $R/gpurun_out/ragged_ab.txt
cat $R/gpurun_out/ragged_ab.txt | cut -c1-200
