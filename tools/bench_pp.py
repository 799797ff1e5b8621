#!/usr/bin/env python3
"""Tuning aid: SNGNN_Plus_Plus conv at arxiv size - adjacency-linear branch timings."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sngnn_amd  # noqa: E402
from sngnn_amd import ops  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
n, c, ei, x, h, lin = bench.make_rank_inputs("arxiv", 0, 1, 1234, dev)


def timed(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


g = Graph(ei, n, True, True)
conv = sngnn_amd.SNConv_plus_plus(128, c, n, 16, 0.0, 0.3, True).to(dev)
w, b = conv.w.weight, conv.w.bias
print("adj_linear forward  %.1f us" % timed(lambda: ops.adj_linear(w.detach(), b.detach(), g)))
g0 = torch.randn(n, c, device=dev)
print("adj_linear backward %.1f us" % timed(lambda: ops.adj_linear_backward(g, g0)))
with torch.no_grad():
    print("conv forward        %.1f us" % timed(lambda: conv(x, ei)))


def fb():
    conv.zero_grad()
    conv(x, ei).sum().backward()


print("conv fwd+bwd        %.1f us" % timed(fb, reps=20))
