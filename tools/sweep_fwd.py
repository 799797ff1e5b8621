#!/usr/bin/env python3
"""Measurement aid: device time of the forward's launches (HIP events inside the library,
event-pair overhead taken off) and the wall time per call of a back-to-back loop, per
(top_k, thr) regime.  Run on the GPU box:  python tools/sweep_fwd.py [workload] [channels]"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sngnn_amd import _lib, ops  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
workload = sys.argv[1] if len(sys.argv) > 1 else "arxiv"
channels = int(sys.argv[2]) if len(sys.argv) > 2 else None
n, c, ei, x, h, lin = bench.make_rank_inputs(workload, 0, 1, 1234, dev, channels, scale=float(os.environ.get("SCALE", 1.0)))
g = Graph(ei, n, True, True)
print(f"workload {workload} C={c} N={n} E'={g.num_edges} unit-row table {n * c * 4 / 2**20:.0f} MiB")
m, f, z, e0 = C.c_float(), C.c_float(), C.c_float(), C.c_float()
if os.environ.get("MODE") == "normalized":      # the caller holds the unit rows: no normalisation pass
    un, unrm = ops.normalize_rows(h)
    call = lambda k, thr: ops.aggregate_forward_normalized(g, un, unrm, k, thr)
else:
    call = lambda k, thr: ops.aggregate_forward(g, h, k, thr)
if os.environ.get("FILTER") is not None:          # A/B of the fp16 filter (csrc/agg_fwd_filter.h)
    lib.sngnn_filter_enable(int(os.environ["FILTER"]))
    print("filter mode", os.environ["FILTER"], "(0 never, 1 auto, 2 always, 3 wave rows and tasks always)")
if os.environ.get("MIN_DEG") is not None:         # wave rows below this in-degree skip the filter (knob 8)
    lib.sngnn_tuning_set(8, int(os.environ["MIN_DEG"]))
    print("filter only for wave rows with in-degree >=", os.environ["MIN_DEG"])
if os.environ.get("FIN") is not None:             # knob 9: 0 = the split rows' finalize as a launch of its own, 1 = the library's rule, v > 1 = inside the main launch on v workgroups
    lib.sngnn_tuning_set(9, int(os.environ["FIN"]))
    print("finalize inside the main launch:", os.environ["FIN"])
if os.environ.get("ROLES") is not None:           # only some row classes of the main kernel (timing only)
    lib.sngnn_tuning_set(0, int(os.environ["ROLES"]))
    print("role mask", os.environ["ROLES"], "(1 tasks, 2 wave rows, 4 small rows)")
if os.environ.get("TABLE") is not None:           # how sngnn_agg_forward scores: 0 auto, 1 unit-row table always, 2 on the fly always
    lib.sngnn_tuning_set(2, int(os.environ["TABLE"]))
    print("scoring mode", os.environ["TABLE"], "(0 auto, 1 table, 2 on the fly)")
REGIMES = ((16, 0.0), (16, 0.9), (1, 0.99), (None, 0.0))
if os.environ.get("REGIMES"):                     # e.g. REGIMES="1:0.0,16:0.0,none:0.0"
    REGIMES = tuple((None if a.lower() == "none" else int(a), float(b))
                    for a, b in (item.split(":") for item in os.environ["REGIMES"].split(",")))
for k, thr in REGIMES:
    res, wall = [], []
    for rnd in range(int(os.environ.get("ROUNDS", 4))):
        lib.sngnn_profile_enable(1)
        for rep in range(12):
            call(k, thr)
            lib.sngnn_profile_last_forward(C.byref(z), C.byref(m), C.byref(f), C.byref(e0))
            if rep >= 2:
                res.append(((z.value - e0.value) * 1e3, (m.value - e0.value) * 1e3, (f.value - e0.value) * 1e3))
        lib.sngnn_profile_enable(0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(50):
            call(k, thr)
        torch.cuda.synchronize()
        wall.append((time.perf_counter() - t0) / 50 * 1e6)
    med = np.median(np.array(res), axis=0)
    print(f"top_k={k} thr={thr}: normalize {med[0]:6.1f}  main {med[1]:6.1f}  finalize {med[2]:5.1f} us (events)   "
          f"wall per call {np.median(wall):6.1f} us", flush=True)
