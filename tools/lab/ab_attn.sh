for rep in 1 2; do
for v in attn_r04 attn_nopair cur; do
  if [ $v = cur ]; then unset VG_LIB; else export VG_LIB=$PWD/tools/lab/lib_$v.so; fi
  echo "== $v"; SHAPES=16x1000,8x2000 python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids
done; done
unset VG_LIB
echo "== window lab, skip thresholds"
for thr in 20 6; do echo "-- VG_ATTN_SKIP=$thr"; VG_ATTN_SKIP=$thr SHAPES=16x1000 SCALES=0.3 python tools/lab/attn_window.py 2>&1 | grep -v amdgpu.ids; done
