#!/usr/bin/env python3
"""Lab: per-kernel hardware counters of any command.

    python3 tools/lab/pmc_any.py <outdir> <kernel-name substring[,substring...]> -- python3 prog.py args...

One `rocprofv3 --pmc` pass per counter group (kernel trace only, the program directly after `--`), counters the
profiler of this box does not list are dropped; prints and writes <outdir>/summary.json: per kernel (matched by
substring) the per-launch average of every counter and the average duration.  Environment variables are inherited.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

args = sys.argv[1:]
out, pats = os.path.abspath(args[0]), args[1].split(",")
cmd = [os.path.abspath(a) if (a.endswith(".py") and os.path.exists(a)) else a for a in args[args.index("--") + 1:]]
os.makedirs(out, exist_ok=True)
env = dict(os.environ, TMPDIR="/tmp")
listing = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, cwd="/tmp").stdout
GROUPS = [
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES",
     "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS", "GRBM_GUI_ACTIVE"],
    ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_SALU", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT",
     "SQ_LDS_IDX_ACTIVE"],
    ["SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_FLAT", "SQ_ACTIVE_INST_MISC", "SQ_INST_CYCLES_VMEM", "SQ_WAVES",
     "SQ_INSTS_SMEM", "SQ_WAIT_INST_ANY"],
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum"],
    ["TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
    ["FETCH_SIZE"],
    ["WRITE_SIZE"],
]
if os.environ.get("PMC_GROUPS"):
    GROUPS = [GROUPS[int(i)] for i in os.environ["PMC_GROUPS"].split(",")]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for gi, grp in enumerate(GROUPS):
    grp = [c for c in grp if c in listing]
    if not grp:
        continue
    d = os.path.join(out, f"g{gi}")
    try:
        subprocess.run(["rocprofv3", "--pmc", *grp, "--kernel-trace", "--output-format", "csv", "-d", d, "--", *cmd],
                       env=env, cwd="/tmp", capture_output=True, text=True, timeout=400)
    except subprocess.TimeoutExpired:
        continue
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            for p in pats:
                if p in r["Kernel_Name"]:
                    acc[p][r["Counter_Name"]] += float(r["Counter_Value"])
                    cnt[p][r["Counter_Name"]] += 1
    for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(fn)):
            for p in pats:
                if p in r["Kernel_Name"]:
                    acc[p][f"_us_g{gi}"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
                    cnt[p][f"_us_g{gi}"] += 1
res = {p: {k: v / max(cnt[p][k], 1) for k, v in sorted(acc[p].items())} for p in pats}
for p in pats:
    res[p]["_launches_per_pass"] = cnt[p].get("_us_g0", 0)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
for p in pats:
    print(p, json.dumps(res[p]))
