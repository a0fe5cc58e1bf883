#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c15; mkdir -p $O
timeout 900 python bench.py --steps 2000 --warmup 20 --no-cpu-baseline > $O/bench_sustained_2000_steps.json 2> $O/sustained.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r6c15/bench_sustained_2000_steps.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("sustained:", round(j["value"]), "tok/s", round(j["ms_per_step"], 3), "ms over", round(j["ms_per_step"] * j["steps"] / 1e3, 1), "s; family", round(r["frac"], 4), "path", round(r["attn_ffn_path_frac"], 4), "probe", round(r["peak_measured"]))
PY
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_default_again.json 2>/dev/null
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r6c15/bench_default_again.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("default:", round(j["value"]), "tok/s", round(j["ms_per_step"], 3), "ms; family", round(r["frac"], 4), "path", round(r["attn_ffn_path_frac"], 4), "probe", round(r["peak_measured"]), "cpu", j.get("cpu_baseline", {}).get("value"))
PY
