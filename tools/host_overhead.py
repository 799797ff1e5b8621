#!/usr/bin/env python3
"""Measurement aid: host-side cost of one ops.aggregate forward + backward through autograd on a
graph so small that the GPU work is negligible (what bounds fwd_bwd_ms on a slow host)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sngnn_amd import ops  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
n, c = 2000, 40
ei = torch.randint(0, n, (2, 8000), device=dev)
g = Graph(ei, n, True, True)
h = torch.randn(n, c, device=dev)
gout = torch.randn(n, c, device=dev)
hg = h.clone().requires_grad_(True)
for name, fn in (("forward (inference)", lambda: ops.aggregate_forward(g, h, 16, 0.0)),
                 ("forward + backward (autograd)", lambda: ops.aggregate(hg, g, 16, 0.0).backward(gout))):
    for _ in range(50):
        hg.grad = None
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2000):
        hg.grad = None
        fn()
    t1 = time.perf_counter()          # host time to ENQUEUE (no sync inside the loop)
    torch.cuda.synchronize()
    print(f"{name}: {(t1 - t0) / 2000 * 1e6:.1f} us of host time per call")
