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
print("epoch ms (graphed):", epoch_time_ms("arxiv", x, ei, n, c, 16, 0.0, graphed=True, epochs=20))
