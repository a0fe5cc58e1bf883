cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sh in dgrad_model ffn_out wgrad; do
  export SHAPE=$sh
  timeout 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmc_g3_$sh -- python3 $R/tools/lab/one_gemm.py > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for sh in ('dgrad_model', 'ffn_out', 'wgrad'):
    acc = collections.defaultdict(float); disp = set()
    for fn in glob.glob(f'gpurun_out/pmc_g3_{sh}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(fn)):
            if 'gemm_ph_kernel' in r['Kernel_Name']:
                acc[r['Counter_Name']] += float(r['Counter_Value']); disp.add(r['Dispatch_Id'])
    n = max(len(disp), 1)
    print(sh, n, {k: round(v / n) for k, v in sorted(acc.items())})
PY
