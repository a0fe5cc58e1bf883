#!/usr/bin/env python3
"""Census of the tiny kernels inside one hipGraph-replayed step from a rocprofv3 kernel trace:
    rocprofv3 --kernel-trace --output-format csv -d <dir> -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline
    python tools/lab/small_runs.py <dir>
Isolates one step (between two groups of adamw launches), counts kernels under 9 us and prints the longest runs of
them with the large kernels before / after each run (that is how the runs are located in the model)."""
import collections
import csv
import glob
import sys


def main(d):
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
    seg = rows[groups[-3][-1] + 1: groups[-2][0]]
    dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    clean = lambda n: n.replace("void at::native::", "").replace("(anonymous namespace)::", "")
    names = [(clean(r["Kernel_Name"])[:70], dur(r)) for r in seg]
    small = [(n, t) for n, t in names if t < 9]
    print(f"kernels in one replayed step: {len(names)} | under 9 us: {len(small)} launches = {sum(t for _, t in small) / 1e3:.2f} ms")
    c = collections.Counter(n for n, _ in small)
    for n, k in c.most_common(14):
        print(f"  {k:4d} x {n}")
    runs, cur, prev = [], [], "<start>"
    for n, t in names:
        if t < 9:
            cur.append(n)
        else:
            if cur:
                runs.append((prev, cur, n))
                cur = []
            prev = n
    print(f"{len(runs)} runs; the longest:")
    for prev, cur, nxt in sorted(runs, key=lambda x: -len(x[1]))[:12]:
        cc = collections.Counter(k.split("<")[0][:24] + ("<" + k.split("<")[1][:30] if "<" in k else "") for k in cur)
        print(f"  {len(cur):3d} after {prev[:38]} | before {nxt[:38]} | {dict(cc.most_common(3))}")


if __name__ == "__main__":
    main(sys.argv[1])
