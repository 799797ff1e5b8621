"""On-the-fly scoring against the table form on data that is easy (random rows) and on data that is its worst case
(nearly parallel rows, a deep layer's input: every decision falls within the re-scoring band).  Config 4's graph.
usage: python tools/micro/otf_worst.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sngnn_amd import _lib, ops  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
n, c, ei, x, h, lin = bench.make_rank_inputs("arxiv", 0, 1, 1234, dev)
g = Graph(ei, n, True, True)
base = torch.randn(1, c, device=dev)
data = {"random rows": h,
        "rows within 1e-3 of parallel": base + 1e-3 * torch.randn(n, c, device=dev),
        "rows within 1e-5 of parallel": base + 1e-5 * torch.randn(n, c, device=dev),
        "identical rows": base.expand(n, c).contiguous()}


def wall(hh, k):
    for _ in range(300):
        ops.aggregate_forward(g, hh, k, 0.0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        ops.aggregate_forward(g, hh, k, 0.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / 300 * 1e6


for name, hh in data.items():
    line = f"{name:32s}"
    for k in (16, 1):
        for mode, label in ((0, "table"), (2, "on the fly")):
            lib.sngnn_tuning_set(2, mode)
            line += f"  top_k {k:2d} {label}: {wall(hh, k):6.1f} us"
    lib.sngnn_tuning_set(2, 0)
    print(line, flush=True)
