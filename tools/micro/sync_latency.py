"""What the bracket around a SHORT timed loop costs: bench.py's step (one forward, three launches) timed over K = 20 and
K = 200 steps with (a) torch.cuda.synchronize() alone at the end, (b) an event recorded behind the loop and polled
(Event.query) before the same synchronize() - the blocking wait's wake-up latency is the difference."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from sngnn_amd import ops  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402

dev = torch.device("cuda:0")
n, c, ei, x, h, lin = bench.make_rank_inputs("arxiv", 0, 1, 1234, dev)
g = Graph(ei, n, True, True)
step = lambda: ops.aggregate_forward(g, h, 16, 0.0)[0]   # noqa: E731


def run(k, poll):
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        step()
    if poll:
        ev = torch.cuda.Event()
        ev.record()
        while not ev.query():
            pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e6


for k in (20, 200):
    for poll in (False, True):
        ts = [run(k, poll) for _ in range(15)]
        print(f"K={k:3d} {'event poll + synchronize' if poll else 'synchronize only       '}: median {np.median(ts):6.2f} us/step  min {min(ts):6.2f}")
