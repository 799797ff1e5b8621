#!/usr/bin/env python3
"""Tuning aid: time the forward's main kernel per row class and per grid size
(HIP events inside the library).  Run on the GPU box."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["SNGNN_DEBUG_LIVE"] = "1"
import bench  # noqa: E402
from sngnn_amd import _lib, ops  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
n, c, ei, x, h, lin = bench.make_rank_inputs("arxiv", 0, 1, 1234, dev)
g = Graph(ei, n, True, True)
k, thr = int(os.environ.get("K", 16)), float(os.environ.get("THR", 0.0))


def timeit(reps=60):
    lib.sngnn_profile_enable(1)
    m, f = C.c_float(), C.c_float()
    ms, fs = [], []
    for _ in range(reps):
        ops.aggregate_forward(g, h, k, thr)
        lib.sngnn_profile_last_forward(C.byref(m), C.byref(f))
        ms.append(m.value)
        fs.append(f.value)
    lib.sngnn_profile_enable(0)
    return float(np.median(ms[10:])) * 1e3, float(np.median(fs[10:])) * 1e3


os.environ["SNGNN_XCD_AFFINITY"] = "0"
for dyn, fin in ((0, 0), (1, 0), (1, 1)):
    os.environ["SNGNN_DYNAMIC"] = str(dyn)
    os.environ["SNGNN_INKERNEL_FIN"] = str(fin)
    for bpc in (5, 6, 7):
        os.environ["SNGNN_DEBUG_BLOCKS_PER_CU"] = str(bpc)
        os.environ["SNGNN_DEBUG_CLASSES"] = "7"
        med, fm = timeit()
        print(f"dynamic={dyn} inkernel_fin={fin} blocks/CU={bpc}: main {med:7.1f} us  "
              f"separate finalize {fm:6.1f} us  total {med + fm:7.1f}", flush=True)
