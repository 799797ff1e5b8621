#!/usr/bin/env python3
"""Generates the committed golden vectors of the aggregation path from the CPU
oracle (oracle/sngnn_oracle.py - a restatement, NOT the reference itself: the
reference cannot be imported here, see the oracle's header).  Run from the repo
root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sngnn_oracle as O  # noqa: E402
from tests.helpers import random_graph  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (n, e, C, hubs, add_loops, remove_loops, top_k, thr)
    "plus_c40_k16": (600, 6000, 40, ((0, 599), (3, 180), (9, 60)), True, True, 16, 0.0),
    "plus_c5_k10_thr09": (400, 5000, 5, ((1, 300),), True, True, 10, 0.9),
    "plus_keep_loops_k1": (300, 2000, 7, ((2, 150),), True, False, 1, 0.99),
    "snconv_c7": (300, 1500, 7, ((5, 200),), True, False, None, 0.0),
}

for name, (n, e, C, hubs, add, rem, k, thr) in CASES.items():
    ei = random_graph(n, e, seed=len(name) * 101 + n, hubs=hubs)
    gen = torch.Generator().manual_seed(n + C)
    h = torch.randn(n, C, generator=gen)
    h[5] = h[6]
    h[8] = 0.0
    gout = torch.randn(n, C, generator=gen)
    hr = h.clone().requires_grad_(True)
    ref = O.aggregate_reference(hr, ei, add_loops=add, remove_loops=rem, top_k=k, thr=thr)
    (ref["out"] * gout).sum().backward()
    arrays = dict(h=h.numpy(), edge_index=ei.numpy(), gout=gout.numpy(),
                  out=ref["out"].detach().numpy(), grad_h=hr.grad.numpy(),
                  s=ref["s"].detach().numpy(), weight=ref["weight"].detach().numpy(),
                  ei_prime=ref["ei"].numpy(),
                  params=np.array([int(add), int(rem), -1 if k is None else k], np.int64),
                  thr=np.array([thr], np.float64))
    if k is not None:
        arrays["sel_src"] = ref["sel_src"].numpy()
    np.savez_compressed(os.path.join(OUT, f"agg_{name}.npz"), **arrays)
    print(name, "E' =", ref["ei"].size(1))
