cd $GRAFT_REPO_ROOT
one() { python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['ms_per_step'],3), 'path', round(d['roofline']['attn_ffn_path_frac'],4))"; }
for i in 1 2 3; do
VG_SIDE_STREAM=0 one off
one early
VG_SIDE_FORK=late one late
VG_MAIN_PRIO=-1 one early_prio
VG_MAIN_PRIO=-1 VG_SIDE_FORK=late one late_prio
done > gpurun_out/side_ab2.txt 2>&1
sort gpurun_out/side_ab2.txt
