#!/bin/bash
# round 6, call 10: conditioning merged into the conv blocks' first 1x1 convolution -- tests, step A/B
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c10; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_round6_gpu.py -x -q -m gpu > $O/pytest_r6.txt 2>&1; echo "rc=$?" >> $O/pytest_r6.txt
tail -12 $O/pytest_r6.txt
timeout 1500 python -m pytest tests/test_kernels_gpu.py tests/test_packed_step_gpu.py tests/test_model_parity_gpu.py -x -q -m gpu -k "bottleneck or packed or step_c1 or step_full or oracle" > $O/pytest_conv.txt 2>&1; echo "rc=$?" >> $O/pytest_conv.txt
tail -6 $O/pytest_conv.txt
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
  VG_COND_MERGE=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/full_cm${v}_$rep.json 2>$O/full_cm${v}_$rep.err; line $O/full_cm${v}_$rep.json "cond_merge=$v"
done; done | tee $O/bench_ab.txt
