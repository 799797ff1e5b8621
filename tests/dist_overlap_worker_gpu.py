"""Worker of tests/test_dist_overlap_gpu.py (two ranks on the box's one GPU, collectives over gloo): the ORDER
in which ``sngnn_amd.dist.halo_aggregate`` puts its work on the launch stream - the one property of the
overlap design a one-GPU box can check:

    exchange issued -> interior rows' kernels enqueued -> exchange waited for -> halo rows normalised,
    boundary rows' kernels enqueued

Each point is a HIP event on the launch stream (``dist.TRACE``); the worker reports the labels in host
order, the device time between consecutive events, the interior / boundary row counts, and whether the
rows equal the unpartitioned forward's.

    dist_overlap_worker_gpu.py OUT.json"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sngnn_amd import dist as sd  # noqa: E402
from sngnn_amd import ops  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402


def main():
    out_path = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    # a graph with locality: 80 % of a row's sources inside its own rank's range -> interior AND boundary rows
    n, c, k, thr = 6000, 40, 8, 0.0
    n_loc = n // world
    rng = np.random.default_rng(5)
    dst = rng.integers(0, n, size=60000)
    own = (dst // n_loc) * n_loc
    src = np.where(rng.random(dst.size) < 0.8, own + rng.integers(0, n_loc, size=dst.size), rng.integers(0, n, size=dst.size))
    key = np.unique(src.astype(np.int64) * n + dst)
    ei = torch.from_numpy(np.stack([key // n, key % n])).to(dev)
    h = torch.randn(n, c, generator=torch.Generator().manual_seed(3)).to(dev)
    part = sd.Partition.even(rank, world, n, exchange="halo")
    plan = sd.HaloPlan(ei, part)
    graph = Graph(plan.edge_index, plan.table_rows, True, True, row_range=(0, part.n_local))
    h_loc = h[part.row_begin:part.row_end].contiguous()
    with torch.no_grad():
        sd.halo_aggregate(h_loc, plan, graph, k, thr)              # warm-up: workspaces, allocator
        torch.cuda.synchronize()
        sd.TRACE = []
        out = sd.halo_aggregate(h_loc, plan, graph, k, thr)
        trace, sd.TRACE = sd.TRACE, None
        torch.cuda.synchronize()
        whole = ops.aggregate_forward(Graph(ei, n, True, True), h, k, thr)[0][part.row_begin:part.row_end]
    labels = [lab for lab, _ in trace]
    gaps_ms = [trace[i][1].elapsed_time(trace[i + 1][1]) for i in range(len(trace) - 1)]
    rec = dict(rank=rank, labels=labels, gaps_ms=gaps_ms, n_local=int(part.n_local), n_boundary=int(plan.n_boundary),
               n_halo=int(plan.n_halo), equal_to_whole=bool(torch.equal(out, whole)))
    recs = [None] * world
    dist.all_gather_object(recs, rec)
    if rank == 0:
        json.dump(recs, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
