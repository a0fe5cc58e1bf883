cd $GRAFT_REPO_ROOT
one() { python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$1', round(d['value']), round(d['ms_per_step'],3), 'tn', k['gemm_bf16_tn']['launches'], round(k['gemm_bf16_tn']['avg_us'],1), 'path', round(d['roofline']['attn_ffn_path_frac'],4))"; }
for i in 1 2 3; do
VG_WDEFER_MIN_ROUNDS=2 one r2
VG_WDEFER_MIN_ROUNDS=3 one r3
VG_WDEFER_MIN_ROUNDS=6 one r6
VG_WDEFER_MIN_ROUNDS=12 one r12
done > gpurun_out/wdefer_ab.txt 2>&1
sort gpurun_out/wdefer_ab.txt
