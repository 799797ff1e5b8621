"""GPU: edge cases of the graph build and the aggregation the reference's semantics
imply (duplicate edges, pre-existing loops kept twice, empty inputs) and the error
behaviour of the boundary."""
import numpy as np
import pytest
import torch

from oracle import sngnn_oracle as O
from tests.helpers import assert_close

pytestmark = pytest.mark.gpu


def run(cuda, h, ei, add, rem, k, thr):
    from sngnn_amd import ops
    from sngnn_amd.graph import Graph
    g = Graph(ei.to(cuda), h.size(0), add, rem)
    out, _, _, sel_src, _ = ops.aggregate_forward(g, h.to(cuda), k, thr, want_selection=k is not None and k > 0)
    return g, out, sel_src


def test_duplicate_edges_and_kept_loops_count_twice(cuda):
    """The convs never coalesce: a duplicated edge is two edges (two candidates, degree
    + 2), and with is_remove_self_loops=0 a pre-existing loop ends up twice
    (models.py:234-236)."""
    h = torch.tensor([[1., 0., 0.], [1., 1., 0.], [0., 1., 1.], [1., 0., 1.]])
    ei = torch.tensor([[1, 1, 2, 0, 0, 3, 3], [0, 0, 0, 0, 1, 3, 2]])      # 1->0 twice, loops 0->0, 3->3
    for add, rem, k, thr in ((True, False, 3, 0.0), (True, False, None, 0.0), (True, True, 2, 0.0)):
        ref = O.aggregate_reference(h, ei, add_loops=add, remove_loops=rem, top_k=k, thr=thr)
        g, out, sel = run(cuda, h, ei, add, rem, k, thr)
        assert g.num_edges == ref["ei"].size(1)
        assert_close(out, ref["out"])
        if k:
            assert torch.equal(sel.cpu().long(), ref["sel_src"])
    # node 0, loops kept: in-edges 1,1,2,0(orig loop),0(added loop) -> degree 5
    g, _, _ = run(cuda, h, ei, True, False, None, 0.0)
    rowptr = g.array("rowptr")
    assert rowptr[1] - rowptr[0] == 5


def test_empty_and_degenerate_graphs(cuda):
    h = torch.randn(5, 8, generator=torch.Generator().manual_seed(0))
    empty = torch.zeros((2, 0), dtype=torch.long)
    # no edges, loops added: every node aggregates only itself with cosine 1
    g, out, _ = run(cuda, h, empty, True, False, 2, 0.0)
    ref = O.aggregate_reference(h, empty, add_loops=True, remove_loops=False, top_k=2, thr=0.0)
    assert g.num_edges == 5
    assert_close(out, ref["out"])
    # no edges at all: zeros (the reference would fail on index.max() of an empty tensor)
    g, out, _ = run(cuda, h, empty, True, True, 2, 0.0)
    assert g.num_edges == 0 and bool((out == 0).all())
    # one node
    g, out, _ = run(cuda, h[:1], empty, True, False, None, 0.0)
    assert_close(out, h[:1])
    # top_k = 0 keeps nothing
    ei = torch.tensor([[0, 1, 2], [1, 2, 0]])
    g, out, _ = run(cuda, h, ei, True, False, 0, 0.0)
    assert bool((out == 0).all())


def test_boundary_errors(cuda):
    from sngnn_amd import SNConv_plus, ops
    from sngnn_amd.graph import Graph
    ei = torch.tensor([[0, 1], [1, 7]], device=cuda)
    with pytest.raises(ValueError, match="outside"):
        Graph(ei, 5, True, False)                                  # node id 7 >= N
    with pytest.raises(ValueError, match="int64"):
        Graph(ei.int(), 8, True, False)
    g = Graph(torch.tensor([[0, 1], [1, 2]], device=cuda), 3, True, False)
    with pytest.raises(ValueError, match="float32"):
        ops.aggregate_forward(g, torch.zeros(3, 4, dtype=torch.float64, device=cuda), 1, 0.0)
    with pytest.raises(ValueError, match=r"\[3, C\]"):
        ops.aggregate_forward(g, torch.zeros(4, 4, device=cuda), 1, 0.0)
    with pytest.raises(ValueError, match="C must be"):
        ops.aggregate_forward(g, torch.zeros(3, 513, device=cuda), 1, 0.0)
    conv = SNConv_plus(4, 3, 3).to(cuda)
    with pytest.raises(ValueError, match="same device"):
        conv(torch.zeros(3, 4, device=cuda), torch.tensor([[0], [1]]))


def test_views_and_cache_invalidation(cuda):
    """Non-contiguous inputs are accepted; an in-place edit of edge_index is seen by the
    graph cache (tensor version counter)."""
    from sngnn_amd import SNConv_plus
    torch.manual_seed(0)
    conv = SNConv_plus(6, 4, 50, top_k=3, thr=0.0).to(cuda)
    x_wide = torch.randn(50, 12, device=cuda)
    x = x_wide[:, ::2]                                             # strided view
    ei_t = torch.randint(0, 50, (300, 2), device=cuda)
    ei = ei_t.t()                                                  # [2, E] view of [E, 2]
    out = conv(x, ei)
    ref = conv(x.contiguous(), ei.contiguous())
    assert torch.equal(out, ref)
    ei_c = ei.contiguous()
    a = conv(x, ei_c)
    ei_c[0, :10] = (ei_c[0, :10] + 1) % 50                         # in-place edit
    b = conv(x, ei_c)
    c = conv(x, ei_c.clone())
    assert torch.equal(b, c) and not torch.equal(a, b)


def test_hub_beyond_the_candidate_finalize_budget(cuda):
    """A hub whose chunk-local candidates would not fit the candidate finalize's LDS budget
    (> ~51 k in-edges at top_k = 16): the forward must route it through the streaming
    finalize (scores in HBM scratch) instead of failing, and agree with the C oracle."""
    from oracle import c_oracle as CO
    from sngnn_amd import ops
    from sngnn_amd.graph import Graph
    n, C, k, thr = 70_000, 8, 16, 0.1
    rng = np.random.default_rng(3)
    src = np.concatenate([np.arange(1, 66_001), rng.integers(0, n, size=40_000)])
    dst = np.concatenate([np.zeros(66_000, np.int64), rng.integers(1, n, size=40_000)])
    key = np.unique(src.astype(np.int64) * n + dst)
    ei = torch.from_numpy(np.stack([key // n, key % n]))
    h = torch.randn(n, C, generator=torch.Generator().manual_seed(2))
    ref = CO.aggregate(h.numpy(), ei.numpy(), add_loops=True, remove_loops=True, top_k=k, thr=thr)
    g = Graph(ei.to(cuda), n, True, True)
    assert g.max_in_degree >= 66_000
    out, wsel, _, sel_src, sel_w = ops.aggregate_forward(g, h.to(cuda), k, thr, save_for_backward=True,
                                                         want_selection=True)
    from tests.helpers import assert_close, check_selection
    assert_close(out, torch.from_numpy(ref["out"]))
    res = dict(sel_src=torch.from_numpy(ref["sel_src"]), ei=torch.from_numpy(ref["ei"]),
               s=torch.from_numpy(ref["s"]))
    assert check_selection(res, sel_src, sel_w, k, thr, strict=False, h=h) <= 2
    np.testing.assert_array_equal(sel_src[0].cpu().numpy(), ref["sel_src"][0])      # the hub itself
