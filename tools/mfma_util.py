#!/usr/bin/env python3
"""MFMA utilisation per matrix-core kernel from a tools/profile_mfma.sh run:
    util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles of the dispatch),
    cycles of the dispatch = GRBM_GUI_ACTIVE / 8   (rocprofv3 sums the counter over the 8 XCDs;
    MI355X_MICROARCH.md, 'DVFS give-back': the quotient reads high on dispatches under ~0.3 ms)
and, beside it, the same busy cycles over duration x 2.4 GHz (the nominal clock) from the
kernel-trace pass.  Grouped per kernel NAME and per distinct grid size (one shape each)."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
KEEP = ("k_cosine_mfma", "k_knn_mfma", "k_linear_rows", "k_wgrad_mfma")


def short(name):
    name = name.split("(")[0]
    for pre in ("void sngnn::", "sngnn::", "void "):
        if name.startswith(pre):
            name = name[len(pre):]
    return name[:64]


acc = defaultdict(lambda: defaultdict(list))       # (kernel, grid) -> counter -> values
for f in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(w in k for w in KEEP):
            acc[(k, r.get("Grid_Size", "?"))][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = defaultdict(list)
for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if any(w in k for w in KEEP):
            grid = str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) if "Grid_Size_X" in r \
                else r.get("Grid_Size", "?")
            dur[(k, grid)].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-9)
print("\n== MFMA utilisation (busy cycles of the matrix pipes / SIMD cycles of the dispatch) ==")
print(f"{'kernel':64s} {'grid':>10s} {'avg_us':>9s} {'MFMA busy':>13s} {'GUI_ACTIVE/8':>13s} {'util':>7s} {'util@2.4GHz':>11s} {'MOPS_BF16':>13s} {'MOPS_F32':>12s}")
for key in sorted(acc):
    c = acc[key]
    mean = lambda n: (sum(c[n]) / len(c[n])) if c.get(n) else float("nan")      # noqa: E731
    busy, gui = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE") / 8.0
    d = sum(dur[key]) / len(dur[key]) if dur.get(key) else float("nan")
    print(f"{key[0]:64s} {key[1]:>10s} {d * 1e6:9.1f} {busy:13.0f} {gui:13.0f} {busy / (gui * 1024):7.1%} "
          f"{busy / (d * 2.4e9 * 1024):11.1%} {mean('SQ_INSTS_VALU_MFMA_MOPS_BF16'):13.0f} {mean('SQ_INSTS_VALU_MFMA_MOPS_F32'):12.0f}")
