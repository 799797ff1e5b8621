#!/usr/bin/env python3
"""Tuning aid: the kNN similarity-graph builder (fused fp32-MFMA cosine + per-row top-k)
against materialising S and torch.topk, and at arxiv size where S (115 GB) cannot exist."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sngnn_amd import toolbox as T  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


from sngnn_amd import _lib  # noqa: E402
lib = _lib.load()
for n, f, k in ((2277, 2325, 10), (7600, 932, 10), (2708, 1433, 10), (12000, 500, 16), (20000, 128, 16), (169343, 128, 16)):
    x = torch.randn(n, f, device=dev)
    t = timed(lambda: T.knn_graph(x, k), reps=3 if n > 50000 else 10)
    flop = 2.0 * n * n * f
    line = f"N={n:7d} F={f:5d} k={k:2d}: builder (route by shape) {t:9.2f} ms  ({flop / t / 1e9:6.1f} TFLOP/s effective)"
    if n <= 16384:
        lib.sngnn_tuning_set(6, 1)
        line += f"   fused scan {timed(lambda: T.knn_graph(x, k), reps=5):9.2f} ms"
        lib.sngnn_tuning_set(6, 2)
        line += f"   dense cosine + row selection {timed(lambda: T.knn_graph(x, k), reps=5):9.2f} ms"
        lib.sngnn_tuning_set(6, 0)
    if n <= 20000:
        def dense():
            s = T.cosine_similarity_dense_small(x)
            s.fill_diagonal_(-2.0)
            return torch.topk(s, k, dim=1)
        line += f"   materialise + torch.topk {timed(dense, reps=5):9.2f} ms"
    print(line, flush=True)
