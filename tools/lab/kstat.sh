#!/bin/bash
# Lab: per-kernel average durations of a python command under rocprofv3 (kernel trace + stats, csv).
#   bash tools/lab/kstat.sh <tag> <name-filter> python3 script.py ...   (environment inherited; run on the GPU box)
tag=$1; filt=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/ks_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$tag -- "$@" > /tmp/ks_$tag.log 2>&1
python3 - "$tag" "$filt" <<'PY'
import csv, glob, sys
tag, filt = sys.argv[1], sys.argv[2]
for f in glob.glob(f"/tmp/ks_{tag}/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if filt in r["Name"]:
            print(tag, r["Name"][:70], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2))
PY
