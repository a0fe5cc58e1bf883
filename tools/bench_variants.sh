#!/bin/bash
# Every bench line profiles/rNN/README.md quotes, in one call on the GPU box:  bash tools/bench_variants.sh <outdir>
O=${1:-gpurun_out/variants}; mkdir -p $O
timeout 400 python bench.py > $O/bench_default.json 2>$O/bench_default.err
timeout 200 python bench.py --seq-len 640 --no-cpu-baseline > $O/bench_T640.json 2>/dev/null
timeout 300 python bench.py --seq-len 2000 --no-cpu-baseline > $O/bench_T2000.json 2>/dev/null
timeout 200 python bench.py --ragged --no-cpu-baseline > $O/bench_ragged.json 2>/dev/null
timeout 200 python bench.py --ragged --packed-rows 0 --no-cpu-baseline > $O/bench_ragged_padded.json 2>/dev/null
timeout 200 python bench.py --ragged --packed-step 1 --no-cpu-baseline > $O/bench_ragged_packed_step.json 2>/dev/null
timeout 200 python bench.py --coalesce 0 --no-cpu-baseline > $O/bench_nocoalesce.json 2>/dev/null
timeout 200 python bench.py --host-batches --no-cpu-baseline > $O/bench_hostbatches.json 2>/dev/null
timeout 200 python bench.py --mode decode > $O/decode_B8.json 2>/dev/null
timeout 200 python bench.py --mode decode --decode-batch 1 > $O/decode_B1.json 2>/dev/null
timeout 200 python bench.py --mode decode --decode-batch 64 > $O/decode_B64.json 2>/dev/null
timeout 200 python bench.py --mode decode --decode-batch 32 > $O/decode_B32.json 2>/dev/null
timeout 300 python bench.py --single-rank-rccl --no-cpu-baseline > $O/bench_single_rank_rccl_torch.json 2>/dev/null
timeout 300 python bench.py --single-rank-rccl --comm abi --no-cpu-baseline > $O/bench_single_rank_rccl_abi.json 2>/dev/null
timeout 100 python tools/attn_bench.py > $O/attn.txt 2>&1
STD=0.3 timeout 100 python tools/attn_bench.py > $O/attn_std0.3.txt 2>&1
VG_DEBUG_GEMM=3 timeout 200 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph 0 2>&1 >/dev/null | python tools/gemm_step_table.py 2 > $O/gemm_step_table.txt
timeout 200 python tools/lab/blaslt_compare.py > $O/gemm_vs_hipblaslt.txt 2>&1
for f in $O/*.json; do
  python - $f <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], '|', d["metric"][:44], '|', round(d["value"], 1), d["unit"], '|', round(d["ms_per_step"], 3), 'ms |',
          d.get("roofline", {}).get("achieved"))
except Exception as e:
    print(sys.argv[1], 'unreadable:', e)
PY
done
