cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for v in cur noskip; do
  if [ $v = cur ]; then unset VG_LIB; else export VG_LIB=$PWD/tools/lab/lib_$v.so; fi
  echo "== $v"; SHAPES=16x1000 python tools/attn_bench.py 2>&1 | grep "B=16"
  STD=0.3 SHAPES=16x1000 python tools/attn_bench.py 2>&1 | grep "B=16"
done; done
