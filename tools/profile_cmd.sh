#!/bin/bash
# Run on the GPU box (through gpurun): rocprofv3 kernel-trace stats of one python tool.
#   tools/profile_cmd.sh <tag> tools/bench_bwd.py    -> gpurun_out/prof/<tag>/kernel_stats.txt
set -e -o pipefail
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $REPO/"$@" > $OUT/trace.log 2>&1
cd $REPO
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
with open(out + "/kernel_stats.txt", "w") as w:
    for r in rows[:25]:
        line = "%-90s calls %6s avg_us %9.2f min %9.2f max %9.2f pct %6s" % (
            r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3,
            float(r["MaxNs"]) / 1e3, r["Percentage"])
        print(line)
        w.write(line + "\n")
PY
