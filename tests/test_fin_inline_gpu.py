"""The split rows' finalize INSIDE the main launch (csrc/agg_fwd_impl.h: fin_block_pair / fin_group_batch;
sngnn_tuning_set(9, v): 0 = a launch of its own, 1 = the library's rule, v > 1 = inside the launch on v
workgroups).  The library's rule takes it only where the launch is long enough (arxiv-sized graphs and up), so
the small graphs of the other test files run the separate launch: here the role is FORCED and held against
the oracle, against the separate launch (same selections and weights bit for bit; the rows of at most 128
candidates bit for bit too - the same summation order; bigger rows to rounding: another order), through replays of
a captured graph (the done words are cleared by their consumer), under a row filter, with a store epilogue, on
the fly, and with the fp16 filter in front."""
import numpy as np
import pytest
import torch

from tests.helpers import assert_close, check_selection, oracle_aggregate, random_graph
from tests.test_agg_forward_gpu import CASES

pytestmark = pytest.mark.gpu


@pytest.fixture
def lib(cuda):
    from sngnn_amd import _lib
    handle = _lib.load()
    yield handle
    handle.sngnn_tuning_set(9, 1)
    handle.sngnn_tuning_set(0, 7)
    handle.sngnn_tuning_set(2, 0)
    handle.sngnn_filter_enable(1)


def forward(cuda, g, h, k, thr, train=True):
    from sngnn_amd.ops import aggregate_forward
    res = aggregate_forward(g, h.to(cuda), k, thr, save_for_backward=train, want_selection=True)
    torch.cuda.synchronize()
    return res


@pytest.mark.parametrize("n,e,C,hubs,add,rem,k,thr", [c for c in CASES if c[6] is not None and 0 < c[6] <= 32 and c[3]])
def test_forced_role_matches_oracle(cuda, lib, n, e, C, hubs, add, rem, k, thr):
    from sngnn_amd.graph import Graph
    lib.sngnn_tuning_set(9, 3)
    ei = random_graph(n, e, seed=n + e + C, hubs=hubs)
    gen = torch.Generator().manual_seed(C * 7 + n)
    h = torch.randn(n, C, generator=gen)
    h[5] = h[6]
    h[7] = 2.0 * h[6]
    h[11] = 0.0
    ref = oracle_aggregate(h, ei, add, rem, k, thr)
    g = Graph(ei.to(cuda), n, add, rem)
    out, wsel, inv, sel_src, sel_w = forward(cuda, g, h, k, thr)
    assert_close(out, ref["out"])
    near = check_selection(ref, sel_src, sel_w, k, thr, strict=False, h=h)
    assert near <= max(1, n // 100)


# (n, e, hubs (node, in-degree), C, top_k, thr, workgroups of the role)
AB = [
    (3000, 20000, ((0, 2999), (1, 2500), (2, 1300), (3, 700), (4, 400), (5, 300), (6, 200), (7, 140)), 40, 16, 0.0, 2),
    (3000, 20000, ((0, 2999), (1, 2500), (2, 1300), (3, 700), (4, 400), (5, 300), (6, 200), (7, 140)), 40, 16, 0.2, 5),
    (3000, 20000, ((0, 2999), (1, 2500), (2, 1300), (3, 700), (4, 400)), 32, 1, 0.0, 3),       # G = 8: eight rows per batch
    (3000, 20000, ((0, 2999), (1, 2500), (2, 1300), (3, 700), (4, 400)), 24, 32, -1.0, 3),
    (3000, 20000, ((0, 2999), (1, 2000), (2, 900), (3, 300)), 100, 3, 0.0, 2),                 # G = 32, k not a power of two
    (3000, 20000, ((0, 2999), (1, 2000), (2, 900), (3, 300)), 200, 5, 0.1, 2),                 # G = 64: one row per "batch"
    (3000, 20000, ((0, 2999), (1, 2000), (2, 900), (3, 300)), 7, 16, 0.0, 2),                  # VEC = 1
    (6000, 30000, tuple((v, 129 + 37 * v) for v in range(60)), 40, 16, 0.0, 4),                # many rows, both kinds
    (6000, 30000, tuple((v, 129 + 37 * v) for v in range(60)), 40, 16, 0.0, 64),               # more workgroups than rows
]


@pytest.mark.parametrize("n,e,hubs,C,k,thr,blocks", AB)
def test_forced_role_equals_the_separate_launch(cuda, lib, n, e, hubs, C, k, thr, blocks):
    from sngnn_amd.graph import Graph
    ei = random_graph(n, e, seed=n + C + k, hubs=hubs)
    h = torch.randn(n, C, generator=torch.Generator().manual_seed(C + k))
    h[17] = h[18]                                   # exact ties inside the hubs' candidate lists
    g = Graph(ei.to(cuda), n, True, True)
    res = {}
    for mode in (0, blocks):
        lib.sngnn_tuning_set(9, mode)
        res[mode] = [t.cpu() for t in forward(cuda, g, h, k, thr)]
    a, b = res[0], res[blocks]
    assert torch.equal(a[1], b[1]), "kept weights per edge"
    assert torch.equal(a[3], b[3]) and torch.equal(a[4], b[4]), "selection lists"
    assert torch.equal(a[2], b[2]), "inverse norms"
    deg = np.diff(g.array("rowptr").astype(np.int64))
    moderate = torch.from_numpy(((deg + 127) // 128) * k <= 128)
    assert (deg > 128).sum() >= len(hubs) - 1 and (k < 8 or moderate.sum() < n)      # both forms of the role run
    assert torch.equal(a[0][moderate], b[0][moderate]), "rows of at most 128 candidates: the same summation order"
    assert_close(b[0], a[0], what="big rows", rtol=1e-6, atol=1e-7)
    # and twice the same bits (the role's order does not depend on which wave comes first)
    lib.sngnn_tuning_set(9, blocks)
    again = forward(cuda, g, h, k, thr)[0].cpu()
    assert torch.equal(again, b[0])


def test_replays_of_a_captured_forward(cuda, lib):
    """The done words a launch reads are zero again when it ends: a replayed graph (same arguments, same
    nonce) finds no stale "done"."""
    from sngnn_amd.graph import Graph
    from sngnn_amd.ops import aggregate_forward
    n, C, k = 4000, 40, 16
    ei = random_graph(n, 25000, seed=5, hubs=((0, 3999), (1, 1500), (2, 600), (3, 200), (4, 150)))
    g = Graph(ei.to(cuda), n, True, True)
    lib.sngnn_tuning_set(9, 4)
    h = torch.randn(n, C, device=cuda)
    want = [aggregate_forward(g, h, k, 0.0)[0].clone()]
    h2 = torch.randn(n, C, device=cuda)
    want.append(aggregate_forward(g, h2, k, 0.0)[0].clone())
    torch.cuda.synchronize()
    x = h.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        aggregate_forward(g, x, k, 0.0)                 # warm-up on the capture stream
        s.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=s):
            out = aggregate_forward(g, x, k, 0.0)[0]
    for rep in range(6):
        x.copy_(h if rep % 2 == 0 else h2)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want[rep % 2]), f"replay {rep}"
        assert bool(torch.isfinite(out).all())


def test_row_filter_with_the_role(cuda, lib):
    """sngnn_agg_forward_rows (a rank's interior rows, then its boundary rows): the role skips what its tasks skip."""
    from sngnn_amd.graph import Graph
    from sngnn_amd import ops
    n, C, k = 3000, 40, 16
    ei = random_graph(n, 20000, seed=9, hubs=((0, 2999), (1, 1500), (2, 600), (3, 200), (4, 150), (5, 140)))
    g = Graph(ei.to(cuda), n, True, True)
    h = torch.randn(n, C, device=cuda)
    lib.sngnn_tuning_set(9, 0)
    want = ops.aggregate_forward(g, h, k, 0.1)[0].clone()
    un, nrm = ops.normalize_rows(h)
    flag = (torch.arange(n, device=cuda) % 3 == 0).to(torch.uint8)      # hubs 0 and 3 in one pass, the others in the other
    for mode in (0, 3):
        lib.sngnn_tuning_set(9, mode)
        out = torch.full((n, C), float("nan"), device=cuda)
        ops.aggregate_forward_rows(g, un, nrm, None, k, 0.1, flag, 1, out)
        torch.cuda.synchronize()
        assert bool(torch.isnan(out[flag == 0]).all()) and bool(torch.isfinite(out[flag == 1]).all())
        ops.aggregate_forward_rows(g, un, nrm, None, k, 0.1, flag, 0, out)
        torch.cuda.synchronize()
        assert_close(out, want, what=f"mode {mode}", rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("what", ["on_the_fly", "filter", "epilogue"])
def test_forced_role_in_the_other_instantiations(cuda, lib, what):
    from sngnn_amd.graph import Graph
    from sngnn_amd import ops
    n, C, k = 3000, 40, 16
    ei = random_graph(n, 20000, seed=21, hubs=((0, 2999), (1, 1500), (2, 600), (3, 200), (4, 150), (5, 140)))
    g = Graph(ei.to(cuda), n, True, True)
    h = torch.randn(n, C, device=cuda)
    thr = 0.3 if what == "filter" else 0.0
    if what == "on_the_fly":
        lib.sngnn_tuning_set(2, 2)
    if what == "filter":
        lib.sngnn_filter_enable(2)
    outs = []
    for mode in (0, 3):
        lib.sngnn_tuning_set(9, mode)
        if what == "epilogue":
            bias = torch.linspace(-0.1, 0.1, C, device=cuda)
            hn = h.clone().requires_grad_(True)
            epi = ops.HiddenEpilogue(True, 0.0, False)
            y = ops.aggregate(hn, g, k, thr, epilogue=epi, bias=bias)
            y.sum().backward()
            outs.append((y.detach().cpu(), hn.grad.cpu()))
        else:
            o = ops.aggregate_forward(g, h, k, thr, save_for_backward=True, want_selection=True)
            torch.cuda.synchronize()
            outs.append((o[0].cpu(), o[1].cpu(), o[3].cpu()))
    assert_close(outs[1][0], outs[0][0], what=what, rtol=1e-6, atol=1e-7)
    for x, y in zip(outs[0][1:], outs[1][1:]):
        if what == "epilogue":
            assert_close(y, x, what=what + " gradient", rtol=1e-5, atol=1e-6)
        else:
            assert torch.equal(x, y)



@pytest.mark.parametrize("C,hubs,blocks,relu", [
    (40, ((0, 2999), (1, 2500), (2, 1300), (3, 700), (4, 400), (5, 300), (6, 200), (7, 140)), 2, False),
    (7, ((0, 2999), (1, 2100), (2, 129)), 3, False),
    (200, ((0, 2999), (1, 2100), (2, 129)), 2, True),
    (64, tuple((v, 129 + 37 * v) for v in range(60)), 5, True),
])
def test_forced_role_without_selection(cuda, lib, C, hubs, blocks, relu):
    """top_k None (SNConv): the role adds the tasks' partial rows in the finalize launch's own order - every row
    bit for bit, with and without a store epilogue, table mode and on the fly."""
    from sngnn_amd.graph import Graph
    from sngnn_amd import ops
    n = 6000 if len(hubs) > 10 else 3000
    ei = random_graph(n, 20000, seed=C + len(hubs), hubs=hubs)
    g = Graph(ei.to(cuda), n, True, True)
    h = torch.randn(n, C, generator=torch.Generator().manual_seed(C)).to(cuda)
    for table_mode in (0, 1):
        lib.sngnn_tuning_set(2, table_mode)
        outs = []
        for mode in (0, blocks):
            lib.sngnn_tuning_set(9, mode)
            if relu and C % 4 == 0:
                bias = torch.linspace(-0.2, 0.2, C, device=cuda)
                y = ops.aggregate(h, g, None, 0.0, epilogue=ops.HiddenEpilogue(True, 0.0, False), bias=bias)
            else:
                y = ops.aggregate_forward(g, h, None, 0.0)[0]
            torch.cuda.synchronize()
            assert lib.sngnn_last_forward_finalize_workgroups() == mode
            outs.append(y.cpu())
        assert torch.equal(outs[0], outs[1]), f"table mode {table_mode}"
    ref = oracle_aggregate(h.cpu(), ei, True, True, None, 0.0)
    lib.sngnn_tuning_set(9, blocks)
    assert_close(ops.aggregate_forward(g, h, None, 0.0)[0], ref["out"])


@pytest.mark.parametrize("k", [16, None])
def test_the_wait_is_bounded(cuda, lib, k):
    """Tasks that never come (masked out: knob 0) - the role gives up after its bounded number of polls and writes the
    rows it could not finalize as NaN: a loud wrong answer within seconds, never a hung GPU.  Every other row is
    computed as usual, and the next call (tasks back) is clean: no stale state."""
    import time
    from sngnn_amd.graph import Graph
    from sngnn_amd import ops
    n, C = 3000, 40
    hubs = ((0, 2999), (1, 1500), (2, 600), (3, 200), (4, 150))
    ei = random_graph(n, 20000, seed=3, hubs=hubs)
    g = Graph(ei.to(cuda), n, True, True)
    h = torch.randn(n, C, device=cuda)
    lib.sngnn_tuning_set(9, 0)
    want = ops.aggregate_forward(g, h, k, 0.0)[0].clone()
    deg = torch.from_numpy(np.diff(g.array("rowptr").astype(np.int64))).to(cuda)
    split = deg > 128
    assert int(split.sum()) >= 5
    lib.sngnn_tuning_set(9, 2)
    lib.sngnn_tuning_set(0, 6)                      # wave rows and small rows only: no task ever says "done"
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = ops.aggregate_forward(g, h, k, 0.0)[0]
    torch.cuda.synchronize()
    took = time.perf_counter() - t0
    assert took < 30.0, f"the bounded wait took {took:.1f} s"
    assert bool(torch.isnan(out[split]).all()), "rows without their tasks must come back as NaN"
    assert torch.equal(out[~split], want[~split])
    lib.sngnn_tuning_set(0, 7)
    again = ops.aggregate_forward(g, h, k, 0.0)[0]
    torch.cuda.synchronize()
    assert bool(torch.isfinite(again).all())
    assert_close(again, want, rtol=1e-6, atol=1e-7)
