# Lab: clocks and power while the training step replays (is the box power-limited?)
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 900 --warmup 3 --no-cpu-baseline > /tmp/b.log 2>&1 &
BP=$!
sleep 32
for i in 1 2 3 4 5 6; do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk\|junction\|edge" | head -8 | tr '\n' ';'
  echo
  sleep 1.5
done
wait $BP
tail -1 /tmp/b.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), d['ms_per_step'], d['roofline']['peak_measured'], d['roofline']['attn_ffn_path_frac'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i power | head -3
