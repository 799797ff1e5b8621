"""GPU: BASELINE config 5's workload on ONE GPU - SNGNN_Plus_Plus forward + backward on an
ogbn-products-sized graph (2 449 029 nodes, ~123.7 M edges, F = 100, 47 classes, the
Linear(num_nodes, C) adjacency table = 460 MB) - checked through size-independent properties
and against the oracle on sampled rows (the oracle cannot run the whole graph in seconds).
models/models.py:116-137 is the layer; the 8-way node partition of the same layer is
tests/test_dist_two_ranks_gpu.py's subject at test size."""
import numpy as np
import pytest
import torch

from oracle import sngnn_oracle as O

pytestmark = pytest.mark.gpu

N, E, F_IN, CLASSES, K, THR, BETA = 2_449_029, 123_718_280, 100, 47, 16, 0.0, 0.3


def _products_like_graph(dev):
    """Power-law in-degrees (a few 17 k hubs, 2 % isolated rows), uniform sources; generated on
    the GPU in a second (multi-edges allowed: the reference treats them as separate edges)."""
    g = torch.Generator(device=dev).manual_seed(5)
    u = torch.rand(N, generator=g, device=dev)
    raw = (1.0 - u).pow(-1.0 / 1.1).clamp_(max=17000.0)
    raw[torch.rand(N, generator=g, device=dev) < 0.02] = 0.0
    deg = torch.floor(raw * (E / raw.sum())).clamp_(max=17000).to(torch.int64)
    deg[7] = 17000
    rem = E - int(deg.sum())                                  # what the floor / clamp lost
    live = torch.nonzero(deg > 0).view(-1)
    deg[live] += rem // live.numel()
    deg[live[torch.randperm(live.numel(), generator=g, device=dev)[: rem % live.numel()]]] += 1
    dst = torch.repeat_interleave(torch.arange(N, device=dev), deg)
    src = torch.randint(0, N, (dst.numel(),), generator=g, device=dev)
    src[:1000] = dst[:1000]                               # some original self-loops
    src[1000] = 0                                         # node 0 has an out-edge (models.py:125)
    order = torch.argsort(src * N + dst)                  # the reference's loaders coalesce: (src, dst) order
    return torch.stack([src[order], dst[order]])


def test_plus_plus_layer_at_products_size(cuda):
    import sngnn_amd
    from sngnn_amd import ops
    from sngnn_amd.graph import GLOBAL_CACHE
    ei = _products_like_graph(cuda)
    e = ei.size(1)
    assert 0.9 * E < e < 1.1 * E
    g = torch.Generator(device=cuda).manual_seed(11)
    x = torch.randn(N, F_IN, generator=g, device=cuda)
    torch.manual_seed(3)
    conv = sngnn_amd.SNConv_plus_plus(F_IN, CLASSES, N, top_k=K, thr=THR, init_beta=BETA,
                                      is_remove_self_loops=True).to(cuda)
    assert conv.w.weight.shape == (CLASSES, N)             # the reference's Linear(num_nodes, C) table
    out = conv(x, ei)
    assert out.shape == (N, CLASSES) and bool(torch.isfinite(out).all())
    gout = torch.randn(N, CLASSES, generator=g, device=cuda) * 1e-3
    (out * gout).sum().backward()

    # --- the layer's two branches, recomputed separately: blend identity and d beta
    graph = GLOBAL_CACHE.get(ei, N, True, True)
    with torch.no_grad():
        from sngnn_amd.conv import _lin_aligned
        h_pad = _lin_aligned(x, conv.lin)[0]          # the layer's own ``lin`` (same bits: a 1-ulp change
        h = h_pad[:, :CLASSES]                        # of h flips near-ties somewhere among 2.4 M rows)
        out_1 = ops.aggregate_forward(graph, h_pad, K, THR)[0][:, :CLASSES]
        out_0 = ops.adj_linear(conv.w.weight, conv.w.bias, graph)
        want = BETA * out_0 + (1 - BETA) * out_1
        assert (out - want).abs().max() <= 1e-5 * want.abs().max() + 1e-6
        gb = (gout.double() * (out_0.double() - out_1.double())).sum()
        assert abs(float(conv.beta.grad) - float(gb)) <= 2e-3 * abs(float(gb)) + 1e-6
        # isolated rows aggregate to zero (mean over max(deg, 1)); their output is beta * bias-branch only
        deg_in = torch.bincount(ei[1][ei[0] != ei[1]], minlength=N)
        iso = torch.nonzero(deg_in == 0).view(-1)[:1000]
        assert float(out_1[iso].abs().max()) == 0.0

    # --- sampled rows against the oracle (hub, its neighbours, random rows)
    gen = np.random.default_rng(0)
    rows = np.unique(np.concatenate([[7, 0, 1], gen.integers(0, N, size=60)]))
    src_c, dst_c = ei[0].cpu(), ei[1].cpu()
    keep = src_c != dst_c
    src_c, dst_c = src_c[keep], dst_c[keep]
    order = torch.argsort(dst_c, stable=True)
    dst_s, src_s = dst_c[order], src_c[order]
    lo = torch.searchsorted(dst_s, torch.from_numpy(rows))
    hi = torch.searchsorted(dst_s, torch.from_numpy(rows) + 1)
    h_cpu_rows = {}
    for i, a, b in zip(rows, lo.tolist(), hi.tolist()):
        nb = src_s[a:b]
        ids = torch.cat([torch.tensor([int(i)]), nb])
        hh = h[ids.to(cuda)].cpu()
        # the row's in-edges as a tiny graph: node 0 = the target, 1.. = its sources in list order
        sub = torch.stack([torch.arange(1, ids.numel()), torch.zeros(nb.numel(), dtype=torch.int64)])
        ref = O.aggregate_reference(hh, sub, add_loops=False, remove_loops=False, top_k=K, thr=THR)
        got = out_1[int(i)].cpu()
        assert (got - ref["out"][0]).abs().max() <= 1e-5 * ref["out"][0].abs().max() + 2e-6, int(i)
        h_cpu_rows[int(i)] = nb

    # adjacency branch and its gradient on sampled nodes: out_0[i] = b + sum_{e: src = i} W[:, dst_e],
    # dW[:, j] = beta * sum_{e: dst = j} gout[src_e]   (loops removed)
    order_s = torch.argsort(src_c, stable=True)
    s_sorted, d_sorted = src_c[order_s], dst_c[order_s]
    lo = torch.searchsorted(s_sorted, torch.from_numpy(rows))
    hi = torch.searchsorted(s_sorted, torch.from_numpy(rows) + 1)
    wt = conv.w.weight.detach().t()
    for i, a, b in zip(rows, lo.tolist(), hi.tolist()):
        tg = d_sorted[a:b].to(cuda)
        ref0 = conv.w.bias.detach().double() + wt[tg].double().sum(0)
        assert (out_0[int(i)].double() - ref0).abs().max() <= 1e-5 * ref0.abs().max() + 1e-6, int(i)
        nb = h_cpu_rows[int(i)].to(cuda)
        dw = BETA * gout[nb].double().sum(0)
        got = conv.w.weight.grad[:, int(i)].double()
        assert (got - dw).abs().max() <= 1e-4 * dw.abs().max() + 1e-9, int(i)
    # lin.weight's gradient exists and is finite (its value is checked at test size elsewhere)
    assert bool(torch.isfinite(conv.lin.weight.grad).all()) and float(conv.lin.weight.grad.abs().max()) > 0


def test_products_eight_way_split_equals_the_whole(cuda):
    """BASELINE config 5's PARTITION at full size on one GPU: the products-sized graph cut into
    the 8 node ranges the 8 ranks would own (``sngnn_graph_create_partition``; all-gather form of the
    exchange, which on one GPU needs no collective - the full feature table is simply there), the
    ++ layer's forward and backward run range by range with ``w`` sharded by node range (a rank's
    adjacency branch gathers W^T rows through the partition of the FLIPPED edge list, as
    sngnn_amd/conv.py does under a partition).  The ranges' outputs, concatenated, are the whole
    graph's output - bit for bit for the aggregation - and the ranges' partial gradients, summed
    (what the reduce-scatter does), are the whole graph's gradients.  models/models.py:116-137;
    SURVEY.md 8e.  Also: edges per rank and time per rank for even and edge-balanced bounds."""
    import time

    from sngnn_amd import ops
    from sngnn_amd.dist import Partition
    from sngnn_amd.graph import Graph
    from tests.helpers import REPORT_LINES
    W, C = 8, 48
    ei = _products_like_graph(cuda)
    g = torch.Generator(device=cuda).manual_seed(21)
    h = torch.randn(N, C, generator=g, device=cuda)
    h[:, 47] = 0.0                                          # 47 classes padded to 16-byte rows
    wt = torch.randn(N, C, generator=g, device=cuda) * 0.01  # W^T of Linear(num_nodes, C)
    bias = torch.randn(C, generator=g, device=cuda) * 0.01
    beta = torch.full((1,), BETA, device=cuda)
    gout = torch.randn(N, C, generator=g, device=cuda) * 1e-3

    def layer(graph_t, graph_adj, rows, whole):
        """forward + backward of the ++ layer on the target rows ``rows`` = (begin, end)."""
        hg, wg, bg, be = (t.clone().requires_grad_(True) for t in (h, wt, bias, beta))
        out1 = ops.aggregate(hg, graph_t, K, THR)
        out0 = ops.adj_linear(wg.t(), bg, graph_t) if whole else ops.gather_sum(wg, bg, graph_adj)
        out = ops.blend(out0, out1, be)
        out.backward(gout[rows[0]:rows[1]])
        return out.detach(), out1.detach(), out0.detach(), hg.grad, wg.grad, bg.grad, be.grad

    gw = Graph(ei, N, True, True)
    whole = layer(gw, None, (0, N), True)
    e_total = gw.num_edges
    del gw
    ei_f = ei.flip(0).contiguous()
    deg_in = torch.bincount(ei[1][ei[0] != ei[1]], minlength=N)
    report = {}
    for name, bounds in (("even", Partition.even_bounds(N, W)),
                         ("edge_balanced", Partition.edge_balanced_bounds(deg_in, W))):
        assert bounds[0] == 0 and bounds[-1] == N and len(bounds) == W + 1
        outs, out1s, out0s = [], [], []
        dh = torch.zeros_like(h)
        dw = torch.zeros_like(wt)
        db = torch.zeros_like(bias)
        dbeta = torch.zeros_like(beta)
        edges, ms = [], []
        for r in range(W):
            rows = (bounds[r], bounds[r + 1])
            gt = Graph(ei, N, True, True, row_range=rows)
            gf = Graph(ei_f, N, True, True, row_range=rows)
            assert gt.num_nodes == rows[1] - rows[0] and gt.num_total_nodes == N
            edges.append(gt.num_edges)
            res = layer(gt, gf, rows, False)               # (also the warm-up of the timing below)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            layer(gt, gf, rows, False)
            torch.cuda.synchronize()
            ms.append((time.perf_counter() - t0) * 1e3)
            outs.append(res[0]); out1s.append(res[1]); out0s.append(res[2])
            dh += res[3]; dw += res[4]; db += res[5]; dbeta += res[6]
            del gt, gf, res
        assert sum(edges) == e_total
        assert torch.equal(torch.cat(out1s), whole[1]), f"{name}: aggregation of the ranges != whole graph"
        out0 = torch.cat(out0s)
        assert float((out0 - whole[2]).abs().max()) <= 1e-6 * float(whole[2].abs().max())
        assert float((torch.cat(outs) - whole[0]).abs().max()) <= 1e-6 * float(whole[0].abs().max())
        for got, want, what in ((dh, whole[3], "grad_h"), (dw, whole[4], "grad_w"), (db, whole[5], "grad_w_bias"),
                                (dbeta, whole[6], "grad_beta")):
            sc = float(want.abs().max())
            tol = 2e-6 * sc if want.numel() > C else 2e-4 * sc      # (the two scalars-ish: signed sums over 117 M terms)
            assert float((got - want).abs().max()) <= tol, (name, what, float((got - want).abs().max()), sc)
        report[name] = (edges, ms)
        REPORT_LINES.append(f"config 5, 8-way {name} split on one GPU: edges per rank max / mean = "
                            f"{max(edges) / (sum(edges) / W):.3f}; ++ layer forward + backward per rank "
                            f"[ms] = {', '.join(f'{t:.2f}' for t in ms)} (max {max(ms):.2f})")
    # the edge-balanced bounds are what they say
    assert max(report["edge_balanced"][0]) <= 1.02 * sum(report["edge_balanced"][0]) / W
