"""Does the step time depend on how long the device has been busy?  Windows of 200 forward calls back to back
(config 4), wall time per call of each window, from a cold start; then the same after a pause.
usage: python tools/micro/ramp.py [windows]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sngnn_amd import ops  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
n, c, ei, x, h, lin = bench.make_rank_inputs("arxiv", 0, 1, 1234, dev)
g = Graph(ei, n, True, True)
ops.aggregate_forward(g, h, 16, 0.0)
torch.cuda.synchronize()
time.sleep(2.0)
windows = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for phase in ("cold start", "after a 2 s pause"):
    out = []
    t_begin = time.perf_counter()
    for w in range(windows):
        t0 = time.perf_counter()
        for _ in range(200):
            ops.aggregate_forward(g, h, 16, 0.0)
        torch.cuda.synchronize()
        out.append(((time.perf_counter() - t0) / 200 * 1e6, (time.perf_counter() - t_begin) * 1e3))
    print(phase + ": us per call (ms since the phase began)")
    print("  " + "  ".join(f"{u:.1f}({t:.0f})" for u, t in out))
    time.sleep(2.0)
