cd $GRAFT_REPO_ROOT
LT=1 bash tools/lab/kstat.sh lt Cijk python3 $GRAFT_REPO_ROOT/tools/lab/ffn_in_shape.py > gpurun_out/ffn_in_lt_kernels.txt 2>&1
grep group_m /tmp/ks_lt.log >> gpurun_out/ffn_in_lt_kernels.txt
cat gpurun_out/ffn_in_lt_kernels.txt
