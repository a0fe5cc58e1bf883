#!/bin/bash
# round 6, GPU call 2: 8-bit derivative with 16-byte paired accesses -- tests, epilogue A/B, step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c2; mkdir -p $O
timeout 1200 python -m pytest tests/test_parity_round6_gpu.py -x -q -m gpu > $O/pytest_r6.txt 2>&1; echo "rc=$?" >> $O/pytest_r6.txt
tail -15 $O/pytest_r6.txt
CFGS=13 M=16000 timeout 300 python tools/lab/epi_sweep.py > $O/epi_sweep.txt 2>&1
CFGS=13 M=16000 timeout 300 python tools/lab/epi_sweep.py >> $O/epi_sweep.txt 2>&1
cat $O/epi_sweep.txt
for i in 1 2 3; do
  for v in 0 1; do
    VG_DERIV_U8=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_u8_${v}_$i.json 2> $O/bench_u8_${v}_$i.err
    python - <<PY
import json
try:
    j = json.loads(open("$O/bench_u8_${v}_$i.json").read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print("VG_DERIV_U8=$v run $i:", round(j["value"]), round(j["ms_per_step"], 3), "family", round(r.get("frac"), 4), "path", round(r.get("attn_ffn_path_frac"), 4), "probe", round(r.get("peak_measured")))
except Exception as e:
    print("bench $v $i failed", e)
PY
  done
done | tee $O/bench_ab.txt
