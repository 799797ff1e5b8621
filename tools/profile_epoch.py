#!/usr/bin/env python3
"""Tuning aid: replay the HIP-graph epoch of the bench workload (for rocprofv3)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from sngnn_amd.train import epoch_time_ms  # noqa: E402

dev = torch.device("cuda:0")
n, c, ei, x, h, lin = bench.make_rank_inputs("arxiv", 0, 1, 1234, dev)
share = os.environ.get("SHARE_EVAL", "0") == "1"          # default: the reference's three forwards
print("epoch ms (graphed, share_eval_forward=%s):" % share,
      epoch_time_ms("arxiv", x, ei, n, c, 16, 0.0, graphed=True, epochs=20, share_eval_forward=share))
