#!/usr/bin/env python3
"""Tuning aid: HIP-graph epoch time (3 forwards, 1 backward, Adam) of the model families at
arxiv size: SNGNN_Plus 1 and 2 layers, SNGNN_Plus_Plus, SNGNN, AGNN."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import sngnn_amd  # noqa: E402
from sngnn_amd.synth import Data  # noqa: E402
from sngnn_amd.train import GraphedEpoch  # noqa: E402

dev = torch.device("cuda:0")
n, c, ei, x, h, lin = bench.make_rank_inputs("arxiv", 0, 1, 1234, dev)
gen = torch.Generator().manual_seed(1)
y = torch.randint(0, c, (n,), generator=gen).to(dev)
r = torch.rand(n, generator=gen)
data = Data(x=x, edge_index=ei, y=y, train_mask=(r < 0.6).to(dev), val_mask=((r >= 0.6) & (r < 0.8)).to(dev),
            test_mask=(r >= 0.8).to(dev))
only = sys.argv[1].split(",") if len(sys.argv) > 1 else None          # e.g. plus_2layer_h32_bn,plusplus_1layer
MODELS = {
    "plus_1layer": lambda: sngnn_amd.SNGNN_Plus(128, 32, c, n, 1, 16, 0.0, 1, 0.5),
    "plus_2layer_h32": lambda: sngnn_amd.SNGNN_Plus(128, 32, c, n, 2, 16, 0.0, 1, 0.5),
    "plus_2layer_h32_bn": lambda: sngnn_amd.SNGNN_Plus(128, 32, c, n, 2, 16, 0.0, 1, 0.5, True),
    # the reference's training scripts: hidden_channels 64, top_k 1 (train_script_SNGNN_plus.sh:13,17)
    "plus_2layer_h64_k1": lambda: sngnn_amd.SNGNN_Plus(128, 64, c, n, 2, 1, 0.0, 1, 0.5),
    "plus_3layer_h64_k1": lambda: sngnn_amd.SNGNN_Plus(128, 64, c, n, 3, 1, 0.0, 1, 0.5),
    # the script's own threshold too (train_script_SNGNN_plus.sh:19: thr 0.99)
    "plus_2layer_h64_k1_thr0.99": lambda: sngnn_amd.SNGNN_Plus(128, 64, c, n, 2, 1, 0.99, 1, 0.5),
    "plusplus_2layer_h64_k1_thr0.99": lambda: sngnn_amd.SNGNN_Plus_Plus(128, 64, c, n, 2, 1, 0.99, 0.5, 1, 0.5),
    "plusplus_1layer": lambda: sngnn_amd.SNGNN_Plus_Plus(128, 32, c, n, 1, 16, 0.0, 0.3, 1, 0.5),
    "plusplus_2layer_h32": lambda: sngnn_amd.SNGNN_Plus_Plus(128, 32, c, n, 2, 16, 0.0, 0.3, 1, 0.5),
    "sngnn_2layer_h32_bn": lambda: sngnn_amd.SNGNN(128, 32, c, 2, True),
    "sngnn_1layer": lambda: sngnn_amd.SNGNN(128, 32, c, 1),
    "agnn_1layer": lambda: sngnn_amd.AGNN(128, 32, c, 1),
    "agnn_2layer_h32": lambda: sngnn_amd.AGNN(128, 32, c, 2),
}
for name, make in MODELS.items():
    if only and name not in only:
        continue
    torch.manual_seed(0)
    model = make().to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    ge = GraphedEpoch(model, data, opt)
    for _ in range(5):
        ge.run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(30):
        ge.run()
    torch.cuda.synchronize()
    print(f"{name:22s} epoch {(time.perf_counter() - t0) / 30 * 1e3:7.3f} ms", flush=True)
