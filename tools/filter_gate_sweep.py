#!/usr/bin/env python3
"""Measurement aid: where does the fp16 filter pay at thr 0 (nothing pruned by the threshold, only by
top_k)?  For a family of synthetic graphs - arxiv's size at 1x / 2x / 4x its edges, uniform in-degrees,
products' degree law at several sizes - the forward's wall time per call with the filter off for such
calls (sngnn_filter_enable(1), round 4's rule) and forced on for the rows above the small class (3), in
ONE process, alternating; beside each the graph's PRUNABLE share P = sum over rows with in-degree >
max(top_k, 16) of (in-degree - top_k), over E' - the quantity the library's rule (agg_fwd.hip:use_filter)
is written in.   python tools/filter_gate_sweep.py [C]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sngnn_amd import _lib, ops, synth  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
C = int(sys.argv[1]) if len(sys.argv) > 1 else 40
if os.environ.get("MIN_DEG") is not None:             # wave rows below this in-degree skip the filter (knob 8)
    lib.sngnn_tuning_set(8, int(os.environ["MIN_DEG"]))
    print("filter only for wave rows with in-degree >=", os.environ["MIN_DEG"])
CASES = [("arxiv 1x", 169343, 1166243, 13000, False), ("arxiv 2x edges", 169343, 2332486, 13000, False),
         ("arxiv 4x edges", 169343, 4664972, 13000, False), ("uniform deg 8", 200000, 1600000, 0, True),
         ("uniform deg 24", 200000, 4800000, 0, True), ("uniform deg 50", 200000, 10000000, 0, True),
         ("products / 16", 153064, 7732392, 17000, False), ("products / 4", 612257, 30929570, 17000, False)]
if os.environ.get("CASES"):
    CASES = [c for c in CASES if any(k in c[0] for k in os.environ["CASES"].split(","))]
for name, n, e, max_deg, uniform in CASES:
    rng = np.random.default_rng(1234)
    ei = torch.from_numpy(synth.make_edges(rng, n, e, max_deg, uniform=uniform)).to(dev)
    g = Graph(ei, n, True, True)
    h = torch.randn(n, C, generator=torch.Generator().manual_seed(1)).to(dev)
    deg = torch.bincount(ei[1][ei[0] != ei[1]], minlength=n)
    for k in (16, 4):
        rank = deg > max(k, 16)
        P = float((deg[rank] - k).sum()) / g.num_edges
        R = float(deg[rank].sum()) / g.num_edges
        t = {1: [], 3: []}
        for rnd in range(3):
            for mode in (1, 3):
                lib.sngnn_filter_enable(mode)
                for _ in range(3):
                    ops.aggregate_forward(g, h, k, 0.0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(30):
                    ops.aggregate_forward(g, h, k, 0.0)
                torch.cuda.synchronize()
                t[mode].append((time.perf_counter() - t0) / 30 * 1e6)
        lib.sngnn_filter_enable(1)
        a, b = float(np.median(t[1])), float(np.median(t[3]))
        print(f"{name:16s} C={C} top_k={k:2d} table {n * C * 4 / 2**20:5.0f} MiB  ranking-row edges R={R:.2f} prunable P={P:.2f}  "
              f"off {a:8.1f} us  on {b:8.1f} us  on/off {b / a:.3f}", flush=True)
    del g, ei, h
