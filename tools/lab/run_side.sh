cd $GRAFT_REPO_ROOT
python -m pytest tests/test_parity_round2_gpu.py tests/test_model_parity_gpu.py tests/test_parity_round3_gpu.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['ms_per_step'],3), 'path', round(d['roofline']['attn_ffn_path_frac'],4))"
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/sl
rocprofv3 --kernel-trace --output-format csv -d /tmp/sl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /tmp/sl.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/lab/step_listing.py /tmp/sl $GRAFT_REPO_ROOT/gpurun_out/step_listing.txt | head -2
python3 $GRAFT_REPO_ROOT/tools/lab/small_runs.py /tmp/sl | head -1
grep -c mask_rows $GRAFT_REPO_ROOT/gpurun_out/step_listing.txt
