cd $GRAFT_REPO_ROOT
one() { python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3), 'fam', round(d['roofline']['frac'],4), 'path', round(d['roofline']['attn_ffn_path_frac'],4))"; }
for i in 1 2 3; do
VG_SIDE_UNET=0 one base
VG_SIDE_UNET=1 one unet_side
VG_SIDE_UNET=1 VG_MAIN_PRIO=-1 one unet_side_mainprio
done
