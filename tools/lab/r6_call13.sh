#!/bin/bash
# round 6, call 13: the 8-bit derivative tile through the LDS-DMA ring (FFN-in dgrad) -- tests, cold A/B, step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c13; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_round6_gpu.py tests/test_parity_round2_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -6 $O/pytest.txt
for rep in 1 2; do for v in 1 0; do
  echo "== VG_AUX_RING=$v"
  VG_AUX_RING=$v CFGS=13 M=16000 timeout 300 python tools/lab/epi_sweep.py 2>&1 | grep cfg13
done; done | tee $O/epi_sweep_ring.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print(sys.argv[2], round(j["value"]), "tok/s", round(j["ms_per_step"], 3), "ms family", round(r.get("frac", 0), 4), "path", round(r.get("attn_ffn_path_frac", 0), 4), "probe", round(r.get("peak_measured", 0)))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2 3; do for v in 1 0; do
  VG_AUX_RING=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/full_ring${v}_$rep.json 2>$O/full_ring${v}_$rep.err; line $O/full_ring${v}_$rep.json "aux_ring=$v"
done; done | tee $O/bench_ab.txt
