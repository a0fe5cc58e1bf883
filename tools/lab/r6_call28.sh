#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6c28; mkdir -p $O
echo "== torch.distributed.run, 1 rank"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400 | tee $O/launcher_1rank.json
echo "== two ranks on one device over gloo (functional dry run of the N > 1 path, timings meaningless)"
VG_BENCH_ONE_DEVICE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 2 --no-cpu-baseline 2>$O/two_rank.err | tail -1 | cut -c1-600 | tee $O/two_rank_one_device.json
tail -3 $O/two_rank.err | cut -c1-300
echo "== sustained"
timeout 600 python bench.py --steps 2000 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_sustained_2000_steps.json; python -c "
import json; d=json.loads(open('$O/bench_sustained_2000_steps.json').read()); print(round(d['value']/1e3,1), round(d['ms_per_step'],3), round(d['roofline']['frac'],4), round(d['roofline']['attn_ffn_path_frac'],4), round(d['roofline']['peak_measured']))"
