cd $GRAFT_REPO_ROOT
one() { python3 bench.py --ragged --ragged-range $2 --packed-step $3 --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 range=$2 packed_step=$3', round(d['value']), round(d['ms_per_step'],3))"; }
for r in 0.5,1.0 0.2,0.7 0.1,0.5 0.05,0.3; do
for i in 1 2; do
one fill $r 0
one fill $r 1
done; done
