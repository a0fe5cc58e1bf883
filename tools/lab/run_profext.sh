# Lab: bench.py's per-kernel durations -- recorded event pairs (VG_PROF_EXT=0) vs the dispatch's own start / stop events
# (VG_PROF_EXT=1) -- next to one REPLAYED step of the same build on the same box from a rocprofv3 kernel trace
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; k=r['kernels']
print('$1', round(d['value']), 'fam', round(r['frac'],4), 'path', round(r['attn_ffn_path_frac'],4), 'path_ms', round(r['attn_ffn_path_ms'],3), ' '.join(f\"{n}:{v['launches']}x{v['avg_us']:.1f}\" for n,v in k.items()))"; }
for i in 1 2; do
VG_PROF_EXT=0 one recorded-pairs
VG_PROF_EXT=1 one dispatch-events
done
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/pe
rocprofv3 --kernel-trace --output-format csv -d /tmp/pe -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/pe.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/lab/step_listing.py /tmp/pe | grep -i "kernels in one step\|attn\|gemm_ring_group\|2, 2, 256\|2, 3, 256\|2, 1, 256" | cut -c1-150
