"""Timing experiment (results are WRONG under mask bit 2): config 5's two-pass backward with the per-kept-edge
1/deg_i gathers of pass S replaced by one address (sngnn_tuning_set(4, 7)) against the real thing (3):
what those 22.3 M four-byte gathers cost.   python tools/micro/bwd_products_exp.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sngnn_amd import _lib, ops  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
n, c, ei, _, h, _ = bench.make_rank_inputs("products", 0, 1, 1234, dev, 48, scale=float(os.environ.get("SCALE", 1.0)))
g = Graph(ei, n, True, True)
gout = torch.randn(n, c, generator=torch.Generator().manual_seed(0)).to(dev)
_, wsel, *_ = ops.aggregate_forward(g, h, 16, 0.0, save_for_backward=True)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
for mask in (3, 7, 3, 7):
    lib.sngnn_tuning_set(4, mask)
    ts = []
    for _ in range(5):
        ev[0].record()
        for _ in range(5):
            ops.aggregate_backward(g, h, gout, wsel, 16)
        ev[1].record()
        ev[1].synchronize()
        ts.append(ev[0].elapsed_time(ev[1]) / 5 * 1e3)
    print(f"role mask {mask}: {np.median(ts[1:]):8.1f} us per backward call", flush=True)
lib.sngnn_tuning_set(4, 3)
