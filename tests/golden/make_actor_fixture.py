#!/usr/bin/env python3
"""Derived-data fixture of the real Actor graph.  (The aggregation / attention / trajectory
fixtures are made by tests/golden/pin_reference.py from runs of the reference's own lines.)
Run from the repo root in the build container:  python tests/golden/make_actor_fixture.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.dirname(os.path.abspath(__file__))

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
