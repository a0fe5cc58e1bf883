#!/bin/bash
# round 6, call 9: split-M + lean packed-row mask -- tests, piece timings, ragged / T=640 A/B, and who launches colsum_final
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c9; mkdir -p $O
timeout 1500 python -m pytest tests/test_parity_round6_gpu.py tests/test_parity_round2_gpu.py tests/test_packed_rows_gpu.py tests/test_packed_step_gpu.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "rc=$?" >> $O/pytest.txt
tail -6 $O/pytest.txt
timeout 600 python tools/lab/split_m_sweep.py 2>&1 | grep -v amdgpu | tee $O/split_m_sweep.txt
line() { python - "$1" "$2" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = j.get("roofline", {})
    print(sys.argv[2], round(j["value"]), "tok/s", round(j["ms_per_step"], 3), "ms family", round(r.get("frac", 0), 4), "path", round(r.get("attn_ffn_path_frac", 0), 4))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
for rep in 1 2; do
  for sp in 1 0; do
    VG_GEMM_SPLIT_M=$sp timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/full_sp${sp}_$rep.json 2>/dev/null; line $O/full_sp${sp}_$rep.json "full split_m=$sp"
    VG_GEMM_SPLIT_M=$sp timeout 300 python bench.py --ragged --steps 20 --warmup 5 --no-cpu-baseline > $O/ragged_sp${sp}_$rep.json 2>/dev/null; line $O/ragged_sp${sp}_$rep.json "ragged split_m=$sp"
    VG_GEMM_SPLIT_M=$sp timeout 300 python bench.py --ragged --packed-step 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/raggedps_sp${sp}_$rep.json 2>/dev/null; line $O/raggedps_sp${sp}_$rep.json "ragged packed-step split_m=$sp"
    VG_GEMM_SPLIT_M=$sp timeout 300 python bench.py --seq-len 640 --steps 20 --warmup 5 --no-cpu-baseline > $O/T640_sp${sp}_$rep.json 2>/dev/null; line $O/T640_sp${sp}_$rep.json "T=640 split_m=$sp"
  done
done | tee $O/bench_ab.txt
( cd /tmp; export TMPDIR=/tmp; timeout 600 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph 0 > /dev/null 2>&1 )
python - <<'PY' | tee gpurun_out/r6c9/colsum_final_dispatches.txt
import csv, glob, collections
f = glob.glob("gpurun_out/r6c9/trace/**/*kernel_trace.csv", recursive=True)
rows = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if "colsum_final" in r["Kernel_Name"]:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        rows[(r["Kernel_Name"][:60], r.get("Grid_Size_X", r.get("Grid_Size")), r.get("Grid_Size_Y"), r.get("Workgroup_Size_X"), round(d, -1))] += 1
for k, n in sorted(rows.items(), key=lambda kv: -kv[0][4]):
    print(n, k)
PY
rm -rf gpurun_out/r6c9/trace
