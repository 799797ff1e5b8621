#!/usr/bin/env python3
"""Randomised check of the kNN builder against the materialised similarity (fp64), including
adversarial orders (every later node more similar than all earlier ones; many exact
duplicates).  usage: fuzz_knn_gpu.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sngnn_amd import toolbox as T  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
cases = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 3000))
    f = int(rng.choice([1, 2, 3, 7, 16, 32, 33, 64, 96, 100, 128, 129, 200]))
    k = int(rng.integers(1, 33))
    excl = bool(rng.integers(0, 2))
    kind = int(rng.integers(0, 4))
    gen = torch.Generator().manual_seed(seed)
    if kind == 0:
        x = torch.randn(n, f, generator=gen)
    elif kind == 1:                      # few distinct rows: masses of exact ties
        base = torch.randn(max(1, n // 50 + 1), f, generator=gen)
        x = base[torch.randint(0, base.size(0), (n,), generator=gen)]
    elif kind == 2:                      # later nodes ever closer to a common direction
        x = torch.zeros(n, f)
        x[:, 0] = 1.0
        if f > 1:
            x[:, 1] = torch.linspace(1.0, 1e-3, n)
        else:
            x[:, 0] = torch.linspace(1.0, 2.0, n)
    else:                                # earlier nodes best (descending), plus zero rows
        x = torch.randn(n, f, generator=gen) * 0.01
        x[:, 0] += torch.linspace(1e-3, 1.0, n).flip(0)
        x[::17] = 0.0
    tag = f"seed={seed} n={n} f={f} k={k} excl={excl} kind={kind}"
    try:
        idx, sim = T.knn_graph(x.to(dev), k, exclude_self=excl)
        idx, sim = idx.cpu(), sim.cpu()
        xn = torch.nn.functional.normalize(x.double(), dim=1)
        S = xn @ xn.t()
        if excl:
            S.fill_diagonal_(-float("inf"))
        m = min(k, n - (1 if excl else 0))
        assert (idx[:, m:] == -1).all() and (sim[:, m:] == 0).all(), "padding"
        if m > 0:
            got, gs = idx[:, :m], sim[:, :m]
            assert (got >= 0).all() and (got < n).all(), "range"
            srt = got.sort(dim=1).values
            assert (srt[:, 1:] != srt[:, :-1]).all(), "duplicate neighbour"
            if excl:
                assert (got != torch.arange(n).unsqueeze(1)).all(), "self"
            true = S.gather(1, got)
            assert (gs.double() - true).abs().max() <= 3e-6, f"sim err {(gs.double() - true).abs().max():.2e}"
            assert (gs[:, 1:] <= gs[:, :-1] + 1e-7).all(), "order"
            kth = torch.topk(S, m, dim=1).values[:, -1]
            assert (true.min(dim=1).values >= kth - 3e-6).all(), "missed a better neighbour"
            # exact ties inside the returned list are in node-id order
            tie = gs[:, 1:] == gs[:, :-1]
            assert (got[:, 1:][tie] > got[:, :-1][tie]).all(), "tie order"
    except Exception as ex:      # noqa: BLE001
        print("FAIL", tag, "->", repr(ex)[:300], flush=True)
        sys.exit(1)
    cases += 1
    seed += 1
    if cases % 200 == 0:
        print(f"{cases} cases ok (last {tag})", flush=True)
print(f"done: {cases} random cases passed, next seed {seed}")
