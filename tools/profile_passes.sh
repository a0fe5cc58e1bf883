#!/bin/bash
# The rocprofv3 passes behind profiles/rNN/ (GPU box):  bash tools/profile_passes.sh <outdir>
# --pmc passes are separate runs with --kernel-trace only (MI355X_MICROARCH.md, HBM / rocprofv3 section).
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/prof}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph 0"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- $B > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- $B > /dev/null 2>&1
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- $B > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 > $O/bench_under_rocprof.json 2>/dev/null
export SHAPES=8x2000,16x1000 ITERS=2
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_attn_fetch -- python3 $R/tools/attn_bench.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_attn_write -- python3 $R/tools/attn_bench.py > /dev/null 2>&1
cd $R; du -sh $O/* | tail -8
