cd $GRAFT_REPO_ROOT
one() { python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3), 'path', round(d['roofline']['attn_ffn_path_frac'],4))"; }
for i in 1 2 3; do
VG_SIDE_UNET=0 one base
VG_SIDE_UNET=1 one unet_side
done
VG_SIDE_UNET=1 python -m pytest tests/test_parity_round2_gpu.py tests/test_model_parity_gpu.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
