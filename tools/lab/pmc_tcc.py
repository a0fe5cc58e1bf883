#!/usr/bin/env python3
"""Lab (round 3): where do the operand requests of the 256x256 long-phase GEMM loop go?

Runs tools/lab/one_gemm.py under `rocprofv3 --pmc` (kernel trace only, one counter group per pass, the program directly
after `--`) for the regular build and for the -DVG_LAB_SAMETILE build (every block reads tile 0: an L2-resident
operand set), and prints per-launch sums of the L2 (TCC) and vector-L1 (TCP) counters that the profiler of this box
offers.  Usage on the GPU box:  python3 tools/lab/pmc_tcc.py <outdir>
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.abspath(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "pmc_tcc"))
os.makedirs(out, exist_ok=True)
os.environ["TMPDIR"] = "/tmp"
os.chdir("/tmp")

listing = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True).stdout
open(os.path.join(out, "counters_available.txt"), "w").write(listing)
have = lambda n: n in listing
GROUPS = [
    ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_READ_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_TAG_STALL_sum", "TCC_BUBBLE_sum"],
    ["TCP_TCC_READ_REQ_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_GATE_EN1_sum"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TA_TCP_STATE_READ_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"],
    ["TA_BUSY_avr", "TA_TA_BUSY_sum", "TA_BUFFER_LOAD_WAVEFRONTS_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum"],
    ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "GRBM_GUI_ACTIVE"],
    ["FETCH_SIZE"],
]
result = {}
for libname, lib in (("regular", None), ("sametile", os.path.join(ROOT, "tools/lab/lib_sametile.so"))):
    if lib and not os.path.exists(lib):
        continue
    for shape in ("qkv",):
        env = dict(os.environ, SHAPE=shape)
        if lib:
            env["VG_LIB"] = lib
        acc = collections.defaultdict(float)
        cnt = collections.defaultdict(int)
        for gi, grp in enumerate(GROUPS):
            grp = [c for c in grp if have(c)]
            if not grp:
                continue
            d = os.path.join(out, f"{libname}_{shape}_g{gi}")
            cmd = ["rocprofv3", "--pmc", *grp, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   "python3", os.path.join(ROOT, "tools/lab/one_gemm.py")]
            try:
                subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
            except subprocess.TimeoutExpired:
                continue
            for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(fn)):
                    if "gemm_ph_kernel" in r["Kernel_Name"]:
                        acc[r["Counter_Name"]] += float(r["Counter_Value"])
                        cnt[r["Counter_Name"]] += 1
            durs = []
            for fn in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(fn)):
                    if "gemm_ph_kernel" in r["Kernel_Name"]:
                        durs.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            if durs:
                acc[f"_us_g{gi}"] = sum(durs) / len(durs)
        result[f"{libname}/{shape}"] = {k: (v if k.startswith("_us") else v / max(cnt[k], 1)) for k, v in sorted(acc.items())}
        print(libname, shape, json.dumps(result[f"{libname}/{shape}"]), flush=True)
json.dump(result, open(os.path.join(out, "summary.json"), "w"), indent=1)
