cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for sh in dgrad_model ffn_out; do
  export SHAPE=$sh
  timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $R/gpurun_out/pmc_g1_$sh -- python3 $R/tools/lab/one_gemm.py > /dev/null 2>&1
  timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d $R/gpurun_out/pmc_g2_$sh -- python3 $R/tools/lab/one_gemm.py > /dev/null 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for sh in ('dgrad_model', 'ffn_out'):
    acc = collections.defaultdict(float); disp = set(); dur = []
    for pas in ('g1', 'g2'):
        for fn in glob.glob(f'gpurun_out/pmc_{pas}_{sh}/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(fn)):
                if 'gemm_ph_kernel' in r['Kernel_Name']:
                    acc[r['Counter_Name']] += float(r['Counter_Value']); disp.add((pas, r['Dispatch_Id']))
        for fn in glob.glob(f'gpurun_out/pmc_{pas}_{sh}/**/*kernel_trace.csv', recursive=True):
            for r in csv.DictReader(open(fn)):
                if 'gemm_ph_kernel' in r['Kernel_Name']:
                    dur.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
    n = len(disp) / 2
    print(sh, 'launches per pass', n, 'median us', sorted(dur)[len(dur) // 2])
    for k, v in sorted(acc.items()):
        print(f'   {k:28s} {v / n:14.0f}')
PY
