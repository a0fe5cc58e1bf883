cd $GRAFT_REPO_ROOT
one() { python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels']; print('$1', round(d['value']), round(d['ms_per_step'],3), 'nt', round(k['gemm_bf16_nt']['avg_us'],2), 'nn', round(k['gemm_bf16_nn']['avg_us'],2), 'path', round(d['roofline']['attn_ffn_path_frac'],4))"; }
export VG_GEMM_WIDE_NN=1
for i in 1 2 3; do
VG_GEMM_GROUP_M_WIDE=0 VG_GEMM_WIDE_N=3072 one base2
VG_GEMM_GROUP_M_WIDE=0 VG_GEMM_WIDE_N=512 one all0
VG_GEMM_GROUP_M_WIDE=0 VG_GEMM_WIDE_N=2048 one n2048
VG_GEMM_GROUP_M_WIDE=2 VG_GEMM_WIDE_N=3072 one g2wide
VG_GEMM_GROUP_M_WIDE=1 VG_GEMM_WIDE_N=3072 one g1wide
VG_GEMM_GROUP_M=8 VG_GEMM_GROUP_M_WIDE=0 VG_GEMM_WIDE_N=3072 one gm8
VG_GEMM_GROUP_M=2 VG_GEMM_GROUP_M_WIDE=0 VG_GEMM_WIDE_N=3072 one gm2
done > gpurun_out/ffn_in_bench_ab3.txt 2>&1
sort gpurun_out/ffn_in_bench_ab3.txt
