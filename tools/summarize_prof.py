#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats + per-kernel PMC averages)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
# which kernels the PMC table lists: the aggregation's by default, the matrix-core ones with "mfma"
KEEP = ("mfma", "linear_rows", "knn", "cosine") if len(sys.argv) > 2 and sys.argv[2] == "mfma" else \
    ("agg", "adj", "bwd", "normalize", "pack_kept", "gather_floor", "blend", "clear_words")


def find(pattern):
    return sorted(glob.glob(os.path.join(root, "**", pattern), recursive=True))


def short(name):
    name = name.split("(")[0]
    for pre in ("void sngnn::", "sngnn::", "void "):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:70]


print("== kernel stats (kernel-trace --stats) ==")
for f in find("*kernel_stats.csv"):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: -float(r.get("TotalDurationNs", 0) or 0))
    print(f"{'kernel':70s} {'calls':>7s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}")
    for r in rows[:18]:
        print(f"{short(r['Name']):70s} {r['Calls']:>7s} {float(r['AverageNs'])/1e3:10.2f} "
              f"{float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} {float(r['Percentage']):7.2f}")

print("\n== PMC (average per dispatch, sngnn kernels) ==")
for f in find("*counter_collection.csv"):
    acc = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not any(w in k for w in KEEP):
            continue
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        for c, v in cs.items():
            print(f"{k:60s} {c:32s} n={len(v):4d} avg={sum(v)/len(v):16.1f}")
