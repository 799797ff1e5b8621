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


# ---------------------------------------------------------------------------
# Harness fixture ("Harness row"): the reference trainer's loop
# (train.py:73-160) on the oracle's restated models, CPU, seeded.
# ---------------------------------------------------------------------------
def make_trajectory(kind, args, name, epochs=5):
    import torch.nn.functional as F
    from sngnn_amd import synth
    data = synth.make_dataset("cora", seed=7, scale=0.25)
    torch.manual_seed(1234)
    model = getattr(O, kind)(*args(data.x.size(1), data.x.size(0)))
    init = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    traj = []
    for _ in range(epochs):
        model.train()
        opt.zero_grad()
        out = model(data)
        loss = F.nll_loss(out[data.train_mask], data.y[data.train_mask])
        loss.backward()
        opt.step()
        model.eval()
        with torch.no_grad():
            out = model(data)
            rec = [float(loss)]
            for m in (data.val_mask, data.test_mask):
                rec += [float(F.nll_loss(out[m], data.y[m])),
                        float((out[m].max(1)[1] == data.y[m]).float().mean())]
        traj.append(rec)
    final = {"final." + k: v.clone().numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, f"traj_{name}.npz"), traj=np.array(traj, np.float64),
                        **{"init." + k: v for k, v in init.items()}, **final)
    print(name, "trajectory", [round(t[0], 4) for t in traj])


make_trajectory("SNGNN_Plus", lambda f, n: (f, 16, 7, n, 2, 3, 0.1, 1, 0.0), "plus_2layer")
make_trajectory("SNGNN_Plus_Plus", lambda f, n: (f, 16, 7, n, 1, 4, 0.2, 0.3, 1, 0.0), "plusplus_1layer")
make_trajectory("SNGNN", lambda f, n: (f, 16, 7, 1), "sngnn_1layer")


# ---------------------------------------------------------------------------
# The real Actor topology (derived data, not source): parsed from the raw files the
# reference bundles under datasets/data/Actor/raw with sngnn_amd/datasets.py, stored as
# a compact fixture so the GPU tests can run on a real degree distribution.
# ---------------------------------------------------------------------------
ACTOR_RAW = "/root/reference/datasets/data/Actor/raw"
if os.path.isdir(ACTOR_RAW):
    from sngnn_amd import datasets as DS
    d = DS.load_geom_gcn(ACTOR_RAW, "film")
    ei = d.edge_index.numpy()
    deg = np.bincount(ei[1], minlength=d.x.size(0))
    np.savez_compressed(os.path.join(OUT, "actor_topology.npz"), edge_index=ei.astype(np.int32),
                        y=d.y.numpy().astype(np.int8),
                        train_mask0=d.train_mask[0].numpy(), val_mask0=d.val_mask[0].numpy(),
                        test_mask0=d.test_mask[0].numpy(),
                        stats=np.array([d.x.size(0), ei.shape[1], int((ei[0] == ei[1]).sum()),
                                        int(deg.max()), int((deg == 0).sum()), d.x.size(1),
                                        int(d.y.max()) + 1], np.int64))
    print("actor topology", d.x.shape, ei.shape, "loops", int((ei[0] == ei[1]).sum()),
          "max in-deg", int(deg.max()), "zero in-deg", int((deg == 0).sum()))
