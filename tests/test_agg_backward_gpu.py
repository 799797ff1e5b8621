"""GPU parity of the aggregation backward and of the SNGNN++ adjacency branch."""
import numpy as np
import pytest
import torch

from oracle import sngnn_oracle as O
from tests.helpers import assert_close, random_graph

pytestmark = pytest.mark.gpu


# gradients are sums with cancellation (the F.normalize projection): compare
# against the magnitude of the gradient, not element by element
def assert_grad_close(got, want, what, rel=2e-5):
    got, want = got.detach().cpu(), want.detach().cpu()
    scale = want.abs().max().clamp_min(1e-30)
    err = (got - want).abs().max()
    assert err <= rel * scale, f"{what}: max err {err.item():.3e} vs scale {scale.item():.3e}"


CASES = [
    (64, 300, 5, (), True, 1, 0.0),
    (64, 300, 7, (), False, None, 0.0),
    (200, 1500, 40, (), True, 16, 0.0),
    (200, 1500, 47, ((3, 90), (7, 150)), True, 6, 0.1),
    (500, 4000, 64, ((0, 499), (9, 300)), True, 3, -0.2),
    (500, 4000, 6, ((0, 499),), False, None, 0.0),
    (400, 3000, 128, ((2, 350),), True, 8, 0.0),
    (300, 2000, 129, (), False, 20, -1.5),
    (300, 2000, 300, ((4, 200),), True, 5, 0.0),
    (1500, 20000, 40, ((0, 1499), (1, 900)), True, 16, 0.0),
]


@pytest.mark.parametrize("n,e,C,hubs,rem,k,thr", CASES)
def test_backward_matches_autograd_of_oracle(cuda, n, e, C, hubs, rem, k, thr):
    from sngnn_amd.graph import Graph
    from sngnn_amd import ops
    ei = random_graph(n, e, seed=n + e + C, hubs=hubs)
    # hub SOURCES as well (out-degree skew exercises the split-source path)
    ei = torch.cat([ei, torch.stack([torch.full((n // 2,), 3), torch.arange(n // 2) * 2 + 1])], 1)
    ei = torch.unique(ei, dim=1)
    gen = torch.Generator().manual_seed(C + n)
    h = torch.randn(n, C, generator=gen)
    h[5] = h[6]
    gout = torch.randn(n, C, generator=gen)

    h_ref = h.clone().requires_grad_(True)
    ref = O.aggregate_reference(h_ref, ei, add_loops=True, remove_loops=rem, top_k=k, thr=thr)
    (ref["out"] * gout).sum().backward()

    g = Graph(ei.to(cuda), n, True, rem)
    h_gpu = h.to(cuda).requires_grad_(True)
    out = ops.aggregate(h_gpu, g, k, thr)
    (out * gout.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    assert_close(out, ref["out"])
    assert_grad_close(h_gpu.grad, h_ref.grad, "grad_h")
    # deterministic: a second run gives the same bits
    h2 = h.to(cuda).requires_grad_(True)
    (ops.aggregate(h2, g, k, thr) * gout.to(cuda)).sum().backward()
    assert torch.equal(h2.grad, h_gpu.grad)


def test_eps_clamped_rows(cuda):
    """Rows below F.normalize's eps: n = h / eps, no projection in the Jacobian."""
    from sngnn_amd.graph import Graph
    from sngnn_amd import ops
    n, C = 40, 8
    ei = random_graph(n, 200, seed=5)
    h = torch.randn(n, C, generator=torch.Generator().manual_seed(2))
    h[3] = 0.0
    h[4] = 1e-14
    gout = torch.randn(n, C, generator=torch.Generator().manual_seed(3))
    h_ref = h.clone().requires_grad_(True)
    ref = O.aggregate_reference(h_ref, ei, add_loops=True, remove_loops=False, top_k=None)
    (ref["out"] * gout).sum().backward()
    g = Graph(ei.to(cuda), n, True, False)
    h_gpu = h.to(cuda).requires_grad_(True)
    (ops.aggregate(h_gpu, g, None, 0.0) * gout.to(cuda)).sum().backward()
    assert_grad_close(h_gpu.grad, h_ref.grad, "grad_h (eps rows)", rel=1e-4)


@pytest.mark.parametrize("n,e,C,hubs,src_min", [(120, 900, 5, ((1, 100),), 0),
                                                 (300, 3000, 47, ((0, 299), (2, 150)), 0),
                                                 (300, 3000, 40, ((5, 200),), 3),
                                                 (200, 1500, 130, (), 0)])
def test_adj_linear_branch(cuda, n, e, C, hubs, src_min):
    from sngnn_amd.graph import Graph
    from sngnn_amd import ops
    ei = random_graph(n, e, seed=n + C, hubs=hubs)
    ei = torch.cat([ei, torch.stack([torch.full((n // 2,), 7), torch.arange(n // 2)])], 1)
    ei = torch.unique(ei, dim=1)
    if src_min:
        ei = ei[:, ei[0] >= src_min]          # no edge leaves nodes < src_min
    gen = torch.Generator().manual_seed(n)
    W = torch.randn(C, n, generator=gen)
    b = torch.randn(C, generator=gen)
    gout = torch.randn(n, C, generator=gen)
    ei_p = O.sn_edge_list(ei, n, True, True)
    assert int(ei_p[0].min()) == src_min

    W_ref, b_ref = W.clone().requires_grad_(True), b.clone().requires_grad_(True)
    out_ref = O.adj_linear_reference(W_ref, b_ref, ei_p, n)
    (out_ref * gout).sum().backward()

    g = Graph(ei.to(cuda), n, True, True)
    assert g.src_min == src_min
    Wg = torch.nn.Parameter(W.t().contiguous().to(cuda).t())     # column-major [C, n]
    bg = b.to(cuda).requires_grad_(True)
    out = ops.adj_linear(Wg, bg, g)
    (out * gout.to(cuda)).sum().backward()
    assert_close(out, out_ref, what="out_0", rtol=1e-5, atol=1e-5)
    assert_grad_close(Wg.grad, W_ref.grad.to_dense() if W_ref.grad.is_sparse else W_ref.grad,
                      "dW")
    assert_grad_close(bg.grad, b_ref.grad, "db")


@pytest.mark.parametrize("parts,k,thr", [(2, 8, 0.0), (3, None, 0.0), (4, 2, 0.1)])
def test_node_range_partitions_reproduce_the_whole_graph(cuda, parts, k, thr):
    """What every rank of a multi-GPU run computes: the partition graph of a node range
    (global column ids, all-gathered h) gives exactly the owned rows of the full
    result, and the ranks' partial grad_h sum (the reduce-scatter) to the full gradient."""
    from sngnn_amd.graph import Graph
    from sngnn_amd import ops
    n, C = 1200, 40
    n -= n % parts
    ei = random_graph(n, 12000, seed=21, hubs=((0, n - 1), (n // 2, 300), (n - 1, 200)))
    ei = torch.cat([ei, torch.stack([torch.full((n // 3,), 5), torch.arange(n // 3) * 3])], 1)
    ei = torch.unique(ei, dim=1).to(cuda)
    gen = torch.Generator().manual_seed(9)
    h = torch.randn(n, C, generator=gen).to(cuda)
    gout = torch.randn(n, C, generator=gen).to(cuda)

    g_full = Graph(ei, n, True, True)
    hf = h.clone().requires_grad_(True)
    out_full = ops.aggregate(hf, g_full, k, thr)
    (out_full * gout).sum().backward()

    step = n // parts
    outs, grad_sum, edges = [], torch.zeros_like(h), 0
    for r in range(parts):
        lo, hi = r * step, (r + 1) * step
        g = Graph(ei, n, True, True, row_range=(lo, hi))
        assert (g.num_nodes, g.num_total_nodes, g.row_offset) == (step, n, lo)
        edges += g.num_edges
        hp = h.clone().requires_grad_(True)
        o = ops.aggregate(hp, g, k, thr)
        assert o.shape == (step, C)
        (o * gout[lo:hi]).sum().backward()
        outs.append(o.detach())
        grad_sum += hp.grad
    assert edges == g_full.num_edges
    assert torch.equal(torch.cat(outs), out_full.detach())
    scale = hf.grad.abs().max()
    assert (grad_sum - hf.grad).abs().max() <= 2e-6 * scale


@pytest.mark.parametrize("parts,rem", [(2, True), (3, False)])
def test_partitioned_adjacency_branch_reproduces_the_whole_graph(cuda, parts, rem):
    """Multi-GPU SNGNN++: each rank gathers W^T rows over the out-edges of its own nodes
    (partition of the flipped edge list); concatenated outputs equal the single-GPU
    branch and the ranks' partial weight gradients sum to the full one."""
    from sngnn_amd.graph import Graph
    from sngnn_amd import ops
    n, C = 900, 40
    n -= n % parts
    ei = random_graph(n, 9000, seed=31, hubs=((0, n - 1), (4, 300)))
    ei = torch.cat([ei, torch.stack([torch.zeros(200, dtype=torch.long), torch.arange(200) + 1])], 1)
    ei = torch.unique(ei, dim=1).to(cuda)                 # node 0 has out-edges: src_min == 0
    gen = torch.Generator().manual_seed(4)
    W = torch.randn(C, n, generator=gen)
    b = torch.randn(C, generator=gen).to(cuda)
    gout = torch.randn(n, C, generator=gen).to(cuda)
    Wg = lambda: torch.nn.Parameter(W.t().contiguous().to(cuda).t())

    g_full = Graph(ei, n, True, rem)
    assert g_full.src_min == 0
    w_full, b_full = Wg(), b.clone().requires_grad_(True)
    out_full = ops.adj_linear(w_full, b_full, g_full)
    (out_full * gout).sum().backward()

    flipped = ei.flip(0).contiguous()
    step = n // parts
    outs, dW, db = [], torch.zeros(C, n, device=cuda), torch.zeros(C, device=cuda)
    for r in range(parts):
        lo, hi = r * step, (r + 1) * step
        g_out = Graph(flipped, n, True, rem, row_range=(lo, hi))
        w_r, b_r = Wg(), b.clone().requires_grad_(True)
        o = ops.adj_linear_partition(w_r, b_r, g_out)
        assert o.shape == (step, C)
        (o * gout[lo:hi]).sum().backward()
        outs.append(o.detach())
        dW += w_r.grad
        db += b_r.grad
    assert_close(torch.cat(outs), out_full.detach(), what="out_0", rtol=1e-5, atol=1e-5)
    assert_grad_close(dW, w_full.grad, "dW")
    assert_grad_close(db, b_full.grad, "db")


def _backward_in_mode(lib, mode, g, h, gout, wsel, hint):
    from sngnn_amd import ops
    lib.sngnn_tuning_set(3, mode)
    return ops.aggregate_backward(g, h, gout, wsel, hint)


@pytest.mark.parametrize("n,e,C,hubs,rem,k,thr", CASES + [
    (3000, 20000, 40, ((0, 2999), (5, 400)), True, 16, 0.0),       # mostly nodes small on both sides
    (3000, 9000, 32, (), True, 2, 0.3),                            # G = 8: two trips over the lists
    (2000, 30000, 1, ((1, 1500),), True, 4, 0.0),
    (50, 0, 12, (), False, 3, 0.0),                                # loops only
    (2500, 40000, 40, ((2, 2400),), True, 120, -1.0),              # up to 120 kept edges in a row's list
    (700, 30000, 8, ((1, 650), (4, 300)), True, 128, -1.5),        # the largest top_k the hint serves
    (4, 3, 4, (), True, 1, -1.5),                                  # a handful of edges: mask of one word
    (3, 1, 2, (), False, 2, 0.0),
])
def test_node_centric_backward_equals_the_two_passes(cuda, n, e, C, hubs, rem, k, thr):
    """sngnn_tuning_set(3, mode) (2 = node-centric whatever the graph): a node small both as target and as source does its pass-T and
    pass-S parts in one work item (dnT in registers, kept bits from the packed mask); the sums keep
    the two passes' order, so grad_h is the same bit for bit - whole graphs and node-range
    partitions.  With the forward's top_k as a hint every node is one work item (rows of any
    in-degree scanned by one wave): split rows then sum their kept edges in one chain instead of
    per-task partial rows - equal to rounding; a hint that is NOT true changes nothing but the
    speed."""
    from sngnn_amd import _lib, ops
    from sngnn_amd.graph import Graph
    lib = _lib.load()
    ei = random_graph(n, e, seed=3 * n + e + C, hubs=hubs) if e else torch.zeros(2, 0, dtype=torch.long)
    ei = torch.cat([ei, torch.stack([torch.full((n // 2,), min(3, n - 1)), torch.arange(n // 2) * 2 + 1])], 1)
    ei = torch.unique(ei, dim=1).to(cuda)
    gen = torch.Generator().manual_seed(C + n)
    h = torch.randn(n, C, generator=gen).to(cuda)
    h[min(7, n - 1)] = 0.0
    gout = torch.randn(n, C, generator=gen).to(cuda)

    def close(x, y):
        return float((x - y).abs().max()) <= 2e-6 * max(float(y.abs().max()), 1e-30)

    try:
        g = Graph(ei, n, True, rem)
        _, wsel, *_ = ops.aggregate_forward(g, h, k, thr, save_for_backward=True)
        two = _backward_in_mode(lib, 1, g, h, gout, wsel, None)
        assert torch.equal(_backward_in_mode(lib, 2, g, h, gout, wsel, None), two)
        if k is not None:
            hinted = _backward_in_mode(lib, 2, g, h, gout, wsel, k)
            assert close(hinted, two)
            assert torch.equal(hinted, _backward_in_mode(lib, 2, g, h, gout, wsel, k))      # deterministic
        # a promise that does not hold (rows keep far more than 1 edge)
        assert close(_backward_in_mode(lib, 2, g, h, gout, wsel, 1), two)
        lo, hi = n // 4, n // 4 + n // 3
        gp = Graph(ei, n, True, rem, row_range=(lo, hi))
        _, wsel_p, *_ = ops.aggregate_forward(gp, h, k, thr, save_for_backward=True)
        two_p = _backward_in_mode(lib, 1, gp, h, gout[lo:hi].contiguous(), wsel_p, None)
        assert torch.equal(_backward_in_mode(lib, 2, gp, h, gout[lo:hi].contiguous(), wsel_p, k), two_p)
    finally:
        lib.sngnn_tuning_set(3, 0)


@pytest.mark.parametrize("C,k,thr", [(40, 16, 0.0), (40, 1, 0.0), (32, 3, 0.2), (8, 2, -0.5), (64, 10, 0.9), (7, 16, 0.0),
                                     (130, 5, 0.0), (48, 16, 0.3)])
def test_kept_bits_written_by_the_forward_equal_the_packed_weights(cuda, C, k, thr):
    """Training calls hand the backward WHICH edges were kept as bits the forward packs itself
    (sngnn_epilogue_t.kept_bits: a halfword per small row, 128 bits per wave row / split-row task,
    the finalize sets the winners') instead of per-edge weights + a packing launch: the gradient is
    the weights path's BIT FOR BIT, on a graph with small rows of every degree, wave rows, split
    rows (a 5 000-edge hub) and isolated nodes - and the bits say exactly what the weights say."""
    from sngnn_amd import _lib, ops
    from sngnn_amd.graph import Graph
    lib = _lib.load()
    n = 6000
    hubs = ((5, 5000), (6, 1300), (7, 129), (8, 128), (9, 17), (10, 16))
    ei = random_graph(n, 30000, seed=C + k, hubs=hubs).to(cuda)
    ei = ei[:, ei[1] < n - 50]                                        # the last 50 nodes have no in-edge
    g = Graph(ei, n, True, True)
    assert g.num_fused_nodes * 2 >= n and ops.kept_bits_supported(g, k)
    gen = torch.Generator().manual_seed(3)
    h = torch.randn(n, C, generator=gen).to(cuda)
    h[11] = h[12]                                                      # an exact tie somewhere
    gout = torch.randn(n, C, generator=gen).to(cuda)
    # the weights path, explicitly
    out_w, wsel, *_ = ops.aggregate_forward(g, h, k, thr, save_for_backward=True)
    grad_w = ops.aggregate_backward(g, h, gout, wsel, k)
    # the autograd function: takes the bits path by itself
    hg = h.clone().requires_grad_(True)
    out_b = ops.aggregate(hg, g, k, thr)
    out_b.backward(gout)
    assert torch.equal(out_b.detach(), out_w)
    assert torch.equal(hg.grad, grad_w)
    # the bits themselves against the weights, edge by edge
    out2, kb = ops._forward_epilogue(g, h, None, k, thr, True, None, None, True)
    assert torch.equal(out2, out_w)
    words = kb.view(torch.int32).cpu().numpy().view(np.uint32)
    rowptr, rperm = g.array("rowptr").astype(np.int64), g.array("rperm").astype(np.int64)
    deg = np.diff(rowptr)
    kept = (wsel.cpu().numpy() > -3.0)
    n_split, n_med_end = int((deg > 128).sum()), int((deg > 16).sum())
    wbase = ((n + 1) // 2 + 3) // 4 * 4
    tbase = wbase + 4 * (n_med_end - n_split)
    task0 = np.concatenate([[0], np.cumsum((deg[rperm[:n_split]] + 127) // 128)])

    def bit(b):
        return (words[b >> 5] >> np.uint32(b & 31)) & np.uint32(1)
    for slot in list(range(min(n_med_end + 40, n))) + list(range(n - 60, n)):
        i = int(rperm[slot])
        if slot >= n_med_end:
            base = 16 * i
        elif slot >= n_split:
            base = 32 * wbase + 128 * (slot - n_split)
        else:
            base = 32 * tbase + 128 * int(task0[slot])
        got = np.array([bit(base + t) for t in range(int(deg[i]))], dtype=bool)
        assert np.array_equal(got, kept[rowptr[i]:rowptr[i + 1]]), (slot, i)
    # knob 3 = 1 (two passes for every node) has no bits path: the function falls back to the weights
    try:
        lib.sngnn_tuning_set(3, 1)
        assert not ops.kept_bits_supported(g, k)
        hg2 = h.clone().requires_grad_(True)
        ops.aggregate(hg2, g, k, thr).backward(gout)
    finally:
        lib.sngnn_tuning_set(3, 0)
    assert float((hg2.grad - grad_w).abs().max()) <= 2e-6 * float(grad_w.abs().max())
    assert not ops.kept_bits_supported(g, 17) and not ops.kept_bits_supported(g, None)
