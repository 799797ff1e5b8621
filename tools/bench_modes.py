#!/usr/bin/env python3
"""Tuning aid: forward / forward+backward time of the aggregation at arxiv size for the
three selection regimes (no selection = SNConv, published k=1 thr=0.99, bench k=16)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sngnn_amd import ops  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
n, c, ei, x, h, lin = bench.make_rank_inputs("arxiv", 0, 1, 1234, dev)


def timed(fn, reps=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for name, add, rem, k, thr in (("SNConv (no selection, loops kept)", True, False, None, 0.0),
                               ("SNConv_plus k=1 thr=0.99 loops kept", True, False, 1, 0.99),
                               ("SNConv_plus k=16 thr=0.0", True, True, 16, 0.0),
                               ("SNConv_plus k=16 thr=0.9", True, True, 16, 0.9),
                               ("SNConv_plus k=64 thr=0.0", True, True, 64, 0.0)):
    g = Graph(ei, n, add, rem)
    fwd = timed(lambda: ops.aggregate_forward(g, h, k, thr))
    hg = h.clone().requires_grad_(True)
    gout = torch.randn_like(h)

    def fb():
        hg.grad = None
        ops.aggregate(hg, g, k, thr).backward(gout)
    print(f"{name:40s} E'={g.num_edges:8d}  fwd {fwd:7.1f} us   fwd+bwd {timed(fb, reps=50):7.1f} us", flush=True)

# cosine-attention mode (AGNNConv): loops replaced, softmax instead of mean
from sngnn_amd.graph import LOOPS_REPLACE  # noqa: E402
g = Graph(ei, n, True, LOOPS_REPLACE)
fwd = timed(lambda: ops.attention_forward(g, h, save_for_backward=False))
fwd_a = timed(lambda: ops.attention_forward(g, h))
hg = h.clone().requires_grad_(True)
gout = torch.randn_like(h)


def fb_attn():
    hg.grad = None
    ops.attention(hg, g).backward(gout)


print(f"{'AGNNConv attention':40s} E'={g.num_edges:8d}  fwd {fwd:7.1f} us (+alpha {fwd_a:7.1f})   "
      f"fwd+bwd {timed(fb_attn, reps=50):7.1f} us", flush=True)
