cd $GRAFT_REPO_ROOT
for rep in 1 2; do
unset VG_LIB; bash tools/lab/kstat.sh base dwnorm python3 $GRAFT_REPO_ROOT/tools/lab/dw_bwd_probe.py
export VG_LIB=$GRAFT_REPO_ROOT/tools/lab/lib_dwocc2.so; bash tools/lab/kstat.sh occ2 dwnorm python3 $GRAFT_REPO_ROOT/tools/lab/dw_bwd_probe.py
done
