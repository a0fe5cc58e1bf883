#!/bin/bash
# Everything profiles/rNN/ quotes, produced on the GPU box in one call and summarised there (only the summaries and the
# kernel statistics travel back):  bash tools/profile_round.sh gpurun_out/prof_rNN
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/prof_round}; mkdir -p $O
bash $R/tools/bench_variants.sh $O/variants > $O/variants.txt 2>&1
bash $R/tools/profile_passes.sh ${1:-gpurun_out/prof_round}/raw > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py traffic $O/raw/pmc_fetch $O/raw/pmc_write > $O/pmc_traffic.json 2>$O/pmc_traffic.err
python3 tools/pmc_summary.py mfma $O/raw/pmc_mfma > $O/pmc_mfma.json 2>$O/pmc_mfma.err
python3 tools/pmc_summary.py traffic $O/raw/pmc_attn_fetch $O/raw/pmc_attn_write > $O/pmc_attn_traffic.json 2>/dev/null
cp $(find $O/raw/stats -name "*kernel_stats.csv" | head -1) $O/bench_n1_kernel_stats.csv 2>/dev/null
cp $O/raw/bench_under_rocprof.json $O/ 2>/dev/null
python3 tools/kstats.py $O/raw/stats 3 40 > $O/kernel_table.txt 2>&1
for T in 640 2000; do
  ( cd /tmp; export TMPDIR=/tmp; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw/stats_T$T -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --graph 0 --seq-len $T > /dev/null 2>&1 )
  cp $(find $O/raw/stats_T$T -name "*kernel_stats.csv" | head -1) $O/kernel_stats_T$T.csv 2>/dev/null
done
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/raw/dec -- python3 $R/bench.py --mode decode > /dev/null 2>&1 )
python3 tools/lab/decode_timeline.py $O/raw/dec > $O/decode_timeline.txt 2>&1
SHAPES=16x1000 ITERS=5 python3 tools/lab/pmc_any.py $O/raw/pmc_attn attn2_fwd,attn_bwd_dq,attn_bwd_dkv -- python3 tools/attn_bench.py > $O/pmc_attn_sq.txt 2>&1
cp $O/raw/pmc_attn/summary.json $O/pmc_attn_sq.json 2>/dev/null
SHAPE=ffn_out PMC_GROUPS=0,1 python3 tools/lab/pmc_any.py $O/raw/pmc_gemm_k4096 gemm_ph_kernel -- python3 tools/lab/one_gemm.py > $O/pmc_gemm_sq_k4096.txt 2>&1
SHAPE=qkv PMC_GROUPS=0,1 python3 tools/lab/pmc_any.py $O/raw/pmc_gemm_k1024 gemm_ph_kernel -- python3 tools/lab/one_gemm.py > $O/pmc_gemm_sq_k1024.txt 2>&1
( cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/raw/listing -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2>&1 )
python3 tools/lab/step_listing.py $O/raw/listing $O/step_listing.txt > $O/step_table_replayed_graph.txt 2>&1
python3 tools/lab/small_runs.py $O/raw/listing > $O/small_launches.txt 2>&1
rm -rf $O/raw
ls -la $O
