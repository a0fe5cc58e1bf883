#!/bin/bash
# DP readiness: one-rank RCCL step variants (plain / torch / abi / abi + CU mask / bf16 wire) + the DP GPU test module, timed
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c8; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS} > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
try:
    j = json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print("$name:", round(j["value"]), "tok/s", round(j["ms_per_step"], 3), "ms  path", round(r.get("attn_ffn_path_frac", 0), 4), "comm_exposed", j.get("comm_exposed_ms"), "probe", round(r.get("peak_measured", 0)))
except Exception as e:
    print("$name failed", e)
PY
}
for rep in 1 2; do
BENCH_ARGS="" run plain_$rep X=1
BENCH_ARGS="--single-rank-rccl --comm torch" run rccl_torch_$rep X=1
BENCH_ARGS="--single-rank-rccl --comm abi" run rccl_abi_$rep X=1
BENCH_ARGS="--single-rank-rccl --comm abi" run rccl_abi_mask16_$rep VG_COMM_CU_MASK=16
BENCH_ARGS="--single-rank-rccl --comm abi" run rccl_abi_mask32_$rep VG_COMM_CU_MASK=32
BENCH_ARGS="--single-rank-rccl --comm abi" run rccl_abi_bf16wire_$rep VG_COMM_DTYPE=bf16
BENCH_ARGS="--single-rank-rccl --comm torch" run rccl_torch_bf16wire_$rep VG_COMM_DTYPE=bf16
done | tee $O/dp_variants.txt
( time timeout 1500 python -m pytest tests/test_dp_gpu.py -x -q -m gpu ) > $O/pytest_dp.txt 2>&1
tail -6 $O/pytest_dp.txt
