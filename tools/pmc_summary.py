#!/usr/bin/env python3
"""Turn rocprofv3 --pmc passes over `bench.py --graph 0` into the per-kernel JSON summaries under profiles/.

  traffic: python tools/pmc_summary.py traffic <fetch_dir> <write_dir> > profiles/rNN/pmc_traffic_vX.json
           (one pass with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE; corrected as MI355X_MICROARCH.md
            prescribes: FETCH_SIZE in KiB and x2 on gfx950, WRITE_SIZE in KiB)
  mfma:    python tools/pmc_summary.py mfma <dir> > profiles/rNN/pmc_mfma_vX.json
           (--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE: MFMA-busy share of the launch's SIMD-cycles =
            busy / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs))

Each directory is what `rocprofv3 --pmc ... --kernel-trace --output-format csv -d <dir> -- python3 bench.py ...`
wrote; every *_counter_collection.csv below it is read."""
import collections
import csv
import glob
import json
import os
import sys

csv.field_size_limit(1 << 30)


def read(directory, counter):
    """-> {(kernel, grid): [values per dispatch]} summed over the counter's dimensions of one dispatch"""
    per_dispatch = collections.defaultdict(float)
    meta = {}
    files = glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *_counter_collection.csv under {directory}")
    for fn in files:
        with open(fn, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                key = (fn, row["Dispatch_Id"])
                per_dispatch[key] += float(row["Counter_Value"])
                meta[key] = (row["Kernel_Name"], int(row["Grid_Size"]))
    out = collections.defaultdict(list)
    for key, v in per_dispatch.items():
        out[meta[key]].append(v)
    return out


def is_bf16_gemm(name):
    return any(k in name for k in ("gemm_dma_kernel", "gemm_dma32_kernel", "gemm_ph_kernel", "gemm_ring_group_kernel"))


def short(name):
    return name.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")


def traffic(fetch_dir, write_dir):
    fetch, write = read(fetch_dir, "FETCH_SIZE"), read(write_dir, "WRITE_SIZE")
    kernels, fam_f, fam_w, fam_n = [], 0.0, 0.0, 0
    for key in sorted(fetch, key=lambda k: -sum(fetch[k])):
        name, grid = key
        f = [v * 1024 * 2 for v in fetch[key]]
        w = [v * 1024 for v in write.get(key, [])]
        n = len(f)
        if is_bf16_gemm(name):
            fam_f += sum(f)
            fam_w += sum(w)
            fam_n += n
        kernels.append({"kernel": short(name)[:110], "grid_threads": grid, "launches": n,
                        "fetch_mb_per_launch": round(sum(f) / n / 1e6, 2),
                        "write_mb_per_launch": round(sum(w) / max(len(w), 1) / 1e6, 2)})
    return {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes, --kernel-trace), "
                      "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph 0",
            "correction": "FETCH_SIZE is reported in KiB and counts 64 B per 128-B request on gfx950: "
                          "bytes = value * 1024 * 2; WRITE_SIZE bytes = value * 1024",
            "bf16_gemm_family": {"launches": fam_n, "fetch_bytes_per_launch": fam_f / max(fam_n, 1),
                                 "write_bytes_per_launch": fam_w / max(fam_n, 1),
                                 "traffic_bytes_per_launch": (fam_f + fam_w) / max(fam_n, 1)},
            "kernels": kernels[:40]}


def mfma(directory):
    busy, active = read(directory, "SQ_VALU_MFMA_BUSY_CYCLES"), read(directory, "GRBM_GUI_ACTIVE")
    rows, fam_b, fam_a = [], 0.0, 0.0
    for key in sorted(busy, key=lambda k: -sum(active.get(k, [0]))):
        b, a = sum(busy[key]), sum(active.get(key, [0]))
        if a <= 0:
            continue
        simd_cycles = a / 8 * 1024
        if is_bf16_gemm(key[0]):
            fam_b += b
            fam_a += simd_cycles
        rows.append({"kernel": short(key[0])[:110], "grid_threads": key[1], "launches": len(busy[key]),
                     "mfma_busy_share": round(b / simd_cycles, 4)})
    return {"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace, "
                      "python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph 0",
            "normalisation": "GRBM_GUI_ACTIVE is summed over the 8 XCDs: SIMD-cycles = GRBM_GUI_ACTIVE / 8 * 1024",
            "bf16_gemm_family_mfma_busy_share": fam_b / max(fam_a, 1), "kernels": rows[:40]}


def sq(directory, counters):
    """Per kernel: each SQ counter summed over the launches, as a share of SQ_WAVE_CYCLES where that makes sense."""
    data = {c: read(directory, c) for c in counters}
    base = data.get("SQ_WAVE_CYCLES") or next(iter(data.values()))
    rows = []
    for key in sorted(base, key=lambda k: -sum(base[k])):
        wc = sum(base[key])
        row = {"kernel": short(key[0])[:90], "grid_threads": key[1], "launches": len(base[key])}
        for c in counters:
            v = sum(data[c].get(key, [0.0]))
            row[c] = v
            if c != "SQ_WAVE_CYCLES" and wc > 0:
                row[c + "/WAVE_CYCLES"] = round(v / wc, 4)
        rows.append(row)
    return {"note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; SQ_VALU_MFMA_BUSY_CYCLES counts cycles "
                    "(MI355X_MICROARCH.md, cycle constants)", "kernels": rows[:30]}


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "sq":
        json.dump(sq(sys.argv[2], sys.argv[3].split(",")), sys.stdout, indent=1)
    elif len(sys.argv) >= 4 and sys.argv[1] == "traffic":
        json.dump(traffic(sys.argv[2], sys.argv[3]), sys.stdout, indent=1)
    elif len(sys.argv) >= 3 and sys.argv[1] == "mfma":
        json.dump(mfma(sys.argv[2]), sys.stdout, indent=1)
    else:
        raise SystemExit(__doc__)
