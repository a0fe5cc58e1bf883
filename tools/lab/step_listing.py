#!/usr/bin/env python3
"""One hipGraph-replayed training step from a rocprofv3 kernel trace, in launch order and aggregated by kernel:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python tools/lab/step_listing.py <dir> [listing.txt]
The step is the stretch between the second-to-last and the third-to-last group of adamw launches."""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = n.replace("void at::native::", "").replace("(anonymous namespace)::", "").replace("_GLOBAL__N_1", "")
    n = re.sub(r"^_ZN\d*", "", n)
    return n[:100]


def main(d, out=None):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
    groups = []
    for i in idx:
        if groups and i - groups[-1][-1] <= 2:
            groups[-1].append(i)
        else:
            groups.append([i])
    seg = rows[groups[-3][-1] + 1: groups[-2][-1] + 1]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    span = (int(seg[-1]["End_Timestamp"]) - int(seg[0]["Start_Timestamp"])) / 1e6
    tot = sum(dur(r) for r in seg) / 1e3
    agg = collections.OrderedDict()
    for r in seg:
        a = agg.setdefault(short(r["Kernel_Name"]), [0, 0.0])
        a[0] += 1
        a[1] += dur(r)
    print(f"{len(seg)} kernels in one step (optimizer launches included), span {span:.2f} ms, kernel time {tot:.2f} ms")
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / 1e3:8.3f} ms  x{c:4d}  avg {t / c:8.1f} us  {n}")
    if out:
        with open(out, "w") as fo:
            t0 = int(seg[0]["Start_Timestamp"])
            for r in seg:
                fo.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {dur(r):8.1f}  {short(r['Kernel_Name'])}\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
