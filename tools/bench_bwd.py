"""Backward of the aggregation at config 4 (arxiv size, C 40, top_k 16, thr 0; GRAPH=products C=48
for config 5's graph): device time of one backward call, node-centric (sngnn_tuning_set(3, 0),
default) without and with the forward's top_k, against the two passes (3, 1)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sngnn_amd import _lib, ops, synth  # noqa: E402
from sngnn_amd.graph import Graph  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    c = int(os.environ.get("C", 40))
    k = int(os.environ.get("TOP_K", 16))
    thr = float(os.environ.get("THR", 0.0))
    if os.environ.get("INPUTS", "bench") == "bench":          # bench.py's graph and h = lin(x)
        import bench
        n, c, ei, _, h, _ = bench.make_rank_inputs(os.environ.get("GRAPH", "arxiv"), 0, 1, 1234, dev, c,
                                                   scale=float(os.environ.get("SCALE", 1.0)))
        g = Graph(ei, n, True, True)
        gout = torch.randn(n, c, generator=torch.Generator().manual_seed(0)).to(dev)
    else:                                                     # the test generator's graph, Gaussian rows
        d = synth.make_dataset(os.environ.get("GRAPH", "arxiv"))
        n = d.x.shape[0]
        g = Graph(d.edge_index.to(dev), n, True, True)
        gen = torch.Generator().manual_seed(0)
        h = torch.randn(n, c, generator=gen).to(dev)
        gout = torch.randn(n, c, generator=gen).to(dev)
    _, wsel, *_ = ops.aggregate_forward(g, h, k if k > 0 else None, thr, save_for_backward=True)
    print("kept edges:", int((wsel > -3.0).sum()), "of", g.num_edges)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    res = {}
    for mode in (1, 0, 2, 1, 0, 2):                   # 2 = node-centric with the top_k hint
        # FORCE=1: the node-centric forms also above the size rule's limit (knob value 2)
        lib.sngnn_tuning_set(3, 1 if mode == 1 else 2 * int(os.environ.get("FORCE", 0)))
        ts = []
        for _ in range(8):
            ev[0].record()
            for _ in range(10):
                gh = ops.aggregate_backward(g, h, gout, wsel, k if (mode == 2 and k > 0) else None)
            ev[1].record()
            ev[1].synchronize()
            ts.append(ev[0].elapsed_time(ev[1]) * 100)
        res.setdefault(mode, []).append(float(np.mean(ts[2:])))
        res[("g", mode)] = gh
    lib.sngnn_tuning_set(3, 2 * int(os.environ.get("FORCE", 0)))
    for roles in (1, 2, 3):                           # timing only: results incomplete unless 3
        lib.sngnn_tuning_set(4, roles)
        ts = []
        for _ in range(6):
            ev[0].record()
            for _ in range(10):
                ops.aggregate_backward(g, h, gout, wsel, k if k > 0 else None)
            ev[1].record()
            ev[1].synchronize()
            ts.append(ev[0].elapsed_time(ev[1]) * 100)
        print("with top_k, roles %d (1 wave-per-node items, 2 fused items): %.1f us/call" % (roles, float(np.mean(ts[2:]))))
    lib.sngnn_tuning_set(4, 3)
    lib.sngnn_tuning_set(3, 0)
    if k > 0 and ops.kept_bits_supported(g, k, c):        # from the kept bits a training forward writes itself
        _, kb = ops._forward_epilogue(g, h, None, k, thr, True, None, None, True)
        ts = []
        for _ in range(8):
            ev[0].record()
            for _ in range(10):
                gb = ops.aggregate_backward_bits(g, h, gout, kb, k)
            ev[1].record()
            ev[1].synchronize()
            ts.append(ev[0].elapsed_time(ev[1]) * 100)
        print("from the forward's kept bits (no packing launch): %.1f us/call; equal to the weights path: %s"
              % (float(np.mean(ts[2:])), bool(torch.equal(gb, res[("g", 2)]))))
        # the training forward that writes them, against the one that writes per-edge weights
        for name, fn in (("forward saving weights", lambda: ops.aggregate_forward(g, h, k, thr, save_for_backward=True)),
                         ("forward saving bits   ", lambda: ops._forward_epilogue(g, h, None, k, thr, True, None, None, True))):
            ts = []
            for _ in range(8):
                ev[0].record()
                for _ in range(10):
                    fn()
                ev[1].record()
                ev[1].synchronize()
                ts.append(ev[0].elapsed_time(ev[1]) * 100)
            print("%s: %.1f us/call" % (name, float(np.mean(ts[2:]))))
    print("two passes   us/call:", res[1])
    print("node-centric us/call:", res[0])
    print("with top_k   us/call:", res[2])
    print("equal bits:", bool(torch.equal(res[("g", 0)], res[("g", 1)])),
          " hint vs two passes max |diff| / max |g|:",
          float((res[("g", 2)] - res[("g", 1)]).abs().max() / res[("g", 1)].abs().max()))


if __name__ == "__main__":
    main()
