#!/bin/bash
# round 6, GPU call 1: the 8-bit stored GELU derivative -- conversion probe, tests, epilogue A/B, step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c1; mkdir -p $O
tools/lab/u8_cvt_probe > $O/u8_cvt_probe.txt 2>&1
timeout 900 python -m pytest tests/test_parity_round6_gpu.py tests/test_parity_round5_gpu.py -x -q -m gpu > $O/pytest_r6.txt 2>&1; echo "rc=$?" >> $O/pytest_r6.txt
tail -5 $O/pytest_r6.txt
for lib in default u8cvt1 u8cvt2; do
  if [ $lib = default ]; then unset VG_LIB; else export VG_LIB=$GRAFT_REPO_ROOT/tools/lab/lib_$lib.so; fi
  echo "== $lib" >> $O/epi_sweep.txt
  CFGS=13 M=16000 timeout 300 python tools/lab/epi_sweep.py >> $O/epi_sweep.txt 2>&1
  # which rounding does this build produce?  (max decode error over a grid, through the C ABI)
  timeout 120 python - >> $O/epi_sweep.txt 2>&1 <<'PY'
import sys, os, math, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "vae-gslm_amd"))
import hipvg
from hipvg import functional as F
d = torch.device("cuda:0")
N = 256
u = torch.linspace(-6, 6, 256 * N).view(256, N).to(d).bfloat16()
eye = torch.eye(N, device=d).bfloat16()
c = torch.zeros(256, N, dtype=torch.uint8, device=d)
F.gemm(u, eye, 256, N, N, act=hipvg.ACT_GELU | hipvg.ACT_SAVE_DERIV | hipvg.ACT_DERIV_U8, aux_out=c, tile_cfg=13)
ud = u.double()
g = 0.5 * torch.erfc(-ud / math.sqrt(2)) + ud * torch.exp(-0.5 * ud * ud) / math.sqrt(2 * math.pi)
e = (c.double() * 0.005 - 0.13 - g)
print(f"decode error: max |e| {float(e.abs().max()):.5f}, mean e {float(e.mean()):+.6f}")
PY
done
unset VG_LIB
cat $O/epi_sweep.txt
for i in 1 2; do
  for v in 0 1; do
    VG_DERIV_U8=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_u8_${v}_$i.json 2> $O/bench_u8_${v}_$i.err
    python - <<PY
import json
try:
    j = json.loads(open("$O/bench_u8_${v}_$i.json").read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print("VG_DERIV_U8=$v run $i:", j["value"], j["ms_per_step"], "family", r.get("frac"), "path", r.get("attn_ffn_path_frac"), "probe", r.get("peak_measured"))
except Exception as e:
    print("bench $v $i failed", e)
PY
  done
done | tee $O/bench_ab.txt
