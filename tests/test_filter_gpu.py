"""The fp16 filter of the selecting forward (csrc/agg_fwd_filter.h): its error bound on the
hardware, and that it never changes a result - selections, weights, outputs and saved
selections with the filter are bit-identical to the unfiltered path."""
import ctypes as C

import numpy as np
import pytest
import torch

from tests.helpers import random_graph

pytestmark = pytest.mark.gpu
FILT_EPS = 1.1e-3          # csrc/agg_fwd_filter.h


def _rows(n, c, seed, kind):
    g = torch.Generator().manual_seed(seed)
    if kind == "normal":
        h = torch.randn(n, c, generator=g)
    elif kind == "parallel":           # nearly parallel rows: every cosine within 1e-3 of 1
        h = torch.randn(1, c, generator=g) + 1e-2 * torch.randn(n, c, generator=g)
    elif kind == "tiny":               # components far below fp16's range before the scaling
        h = torch.randn(n, c, generator=g) * torch.logspace(-30, 0, c).view(1, -1)
    else:                              # sparse non-negative (bag-of-words after a ReLU)
        h = torch.relu(torch.randn(n, c, generator=g) - 1.0)
    h[5] = h[6]
    h[9] = 0.0
    return h


@pytest.mark.parametrize("c", [36, 40, 48, 64, 100, 128, 256, 512])
@pytest.mark.parametrize("kind", ["normal", "parallel", "tiny", "sparse"])
def test_filter_error_bound_on_the_hardware(cuda, c, kind):
    from sngnn_amd import _lib, ops
    n = 3000
    h = _rows(n, c, c, kind).to(cuda)
    un, _, filt = ops.normalize_rows_filter(h)
    assert filt is not None and filt.shape == (n, ops.filter_row_bytes(c)) and filt.size(1) % 128 == 0
    g = torch.Generator().manual_seed(1)
    pa = torch.randint(0, n, (200_000,), generator=g).to(cuda)
    pb = torch.randint(0, n, (200_000,), generator=g).to(cuda)
    out = torch.empty(pa.numel(), dtype=torch.float32, device=cuda)
    _lib.check(_lib.load().sngnn_filter_pair_scores(filt.data_ptr(), c, pa.data_ptr(), pb.data_ptr(), pa.numel(),
                                                    out.data_ptr(), torch.cuda.current_stream().cuda_stream), "pairs")
    exact = (un[pa].double() * un[pb].double()).sum(-1)
    err = (out.double() - exact).abs().max().item()
    assert err <= 0.95 * FILT_EPS, err            # the proven bound leaves slack for the fp32 sums
    # what the table holds: fl16(1024 * n), zero beyond C
    f16 = filt.view(torch.float16)
    assert f16.shape[1] >= c and (f16[:, c:] == 0).all()
    assert torch.equal(f16[:, :c], (un * 1024.0).to(torch.float16))


CASES = [
    # n, e, C, hubs, top_k, thr, kind
    (3000, 60000, 40, ((0, 2999), (3, 900), (9, 300)), 16, 0.0, "normal"),
    (3000, 60000, 40, ((0, 2999), (3, 900)), 16, 0.9, "normal"),
    (2000, 50000, 40, ((1, 1500),), 10, 0.0, "parallel"),
    (2000, 50000, 64, ((1, 1500),), 1, 0.99, "sparse"),
    (1500, 40000, 48, ((2, 1400), (7, 200)), 32, -1.0, "normal"),
    (1500, 40000, 128, ((2, 1400),), 16, 0.0, "sparse"),
    (1200, 30000, 256, ((2, 1100),), 4, 0.0, "tiny"),
    (1000, 30000, 36, ((4, 999),), 16, 0.0, "sparse"),
    (1000, 30000, 40, ((4, 999),), 0, 0.0, "normal"),
]


@pytest.mark.parametrize("n,e,c,hubs,k,thr,kind", CASES)
def test_filter_changes_nothing(cuda, n, e, c, hubs, k, thr, kind):
    from sngnn_amd import _lib, ops
    from sngnn_amd.graph import Graph
    lib = _lib.load()
    ei = random_graph(n, e, seed=n + c, hubs=hubs).to(cuda)
    h = _rows(n, c, 7 * c + k, kind).to(cuda)
    g = Graph(ei, n, True, True)
    res = []
    try:
        for on in (2, 0):                       # always / never (1 = auto is the default)
            lib.sngnn_filter_enable(on)
            out, wsel, _, _, _ = ops.aggregate_forward(g, h, k, thr, save_for_backward=True)
            _, _, _, sel_src, sel_w = ops.aggregate_forward(g, h, k, thr, want_selection=True)
            res.append((out, wsel, sel_src, sel_w))
    finally:
        lib.sngnn_filter_enable(1)
    for a, b, what in zip(res[0], res[1], ("out", "wsel", "sel_src", "sel_w")):
        assert torch.equal(a, b), what
    # and through the entry for callers that hold the unit rows (+ filter rows)
    un, nrm, filt = ops.normalize_rows_filter(h)
    out_p, src_p, w_p = ops.aggregate_forward_normalized(g, un, nrm, k, thr, want_selection=True, filt=filt)
    out_n, src_n, w_n = ops.aggregate_forward_normalized(g, un, nrm, k, thr, want_selection=True)
    for a, b in ((out_p, res[0][0]), (out_n, res[0][0]), (src_p, res[0][2]), (src_n, res[0][2]),
                 (w_p, res[0][3]), (w_n, res[0][3])):
        assert torch.equal(a, b)


@pytest.mark.parametrize("c,k,thr", [(40, 16, 0.0), (40, 4, 0.5), (7, 3, 0.0), (48, None, 0.0), (64, 16, 0.9)])
def test_row_filtered_passes_equal_one_pass(cuda, c, k, thr):
    """sngnn_agg_forward_rows: two launches with complementary row flags write exactly what one
    unrestricted forward writes - out, the saved selection and the inverse norms (one rank's
    interior / boundary passes, sngnn_amd/dist.py:halo_aggregate)."""
    from sngnn_amd import ops
    from sngnn_amd.graph import Graph
    n = 2500
    ei = random_graph(n, 50000, seed=c, hubs=((0, 2499), (3, 700), (9, 300), (11, 140))).to(cuda)
    h = _rows(n, c, 5 + c, "normal").to(cuda)
    g = Graph(ei, n, True, True)
    out, wsel, inv, _, _ = ops.aggregate_forward(g, h, k, thr, save_for_backward=True)
    gen = torch.Generator().manual_seed(2)
    flag = (torch.rand(n, generator=gen) < 0.4).to(torch.uint8).to(cuda)
    flag[0] = 1                                   # the biggest hub on one side, the next on the other
    flag[3] = 0
    un, nrm, filt = ops.normalize_rows_filter(h)
    if filt is not None and not ops.filter_wanted(g, c, k, thr):
        filt = None
    if k is None:                                 # nothing selected: sngnn_agg_forward scores on the fly from h
        un, nrm, filt = h, None, None
    out2 = torch.full_like(out, float("nan"))
    wsel2 = torch.full_like(wsel, float("nan"))
    inv2 = torch.full_like(inv, float("nan"))
    for want in (0, 1):
        ops.aggregate_forward_rows(g, un, nrm, filt, k, thr, flag, want, out2, wsel2, inv2)
        if want == 0:                             # the other side's rows are untouched so far
            assert torch.isnan(out2[flag.bool()]).all() and not torch.isnan(out2[~flag.bool()]).any()
    assert torch.equal(out2, out) and torch.equal(wsel2, wsel) and torch.equal(inv2, inv)


OTF_CASES = [
    # n, e, C, hubs, top_k, thr, kind
    (3000, 60000, 40, ((0, 2999), (3, 900), (9, 300)), 16, 0.0, "normal"),
    (3000, 60000, 40, ((0, 2999), (3, 900)), 1, 0.99, "normal"),
    (2000, 50000, 40, ((1, 1500),), 10, 0.0, "parallel"),       # every cosine within 1e-3 of 1: doubt everywhere
    (2000, 50000, 8, ((1, 1500),), 4, 0.0, "sparse"),           # many duplicate / zero rows: exact ties
    (1500, 40000, 48, ((2, 1400), (7, 200)), 40, -1.0, "normal"),   # top_k > 32: scores through HBM scratch
    (1200, 30000, 7, ((2, 1100),), 3, 0.5, "tiny"),
    (1000, 30000, 1, ((4, 999),), 5, -0.5, "normal"),           # C == 1: every cosine is exactly +-1
    (1000, 30000, 40, ((4, 999),), None, 0.0, "normal"),        # SNConv: on the fly is the default
    (900, 20000, 64, ((4, 800),), 300, 0.2, "sparse"),          # streaming split rows
]


@pytest.mark.parametrize("n,e,c,hubs,k,thr,kind", OTF_CASES)
def test_scoring_on_the_fly_selects_what_the_table_selects(cuda, n, e, c, hubs, k, thr, kind):
    """sngnn_tuning_set(2, mode): scoring straight from h (fast cosine; exact normalise-then-dot
    wherever a decision is within the error bound) against the normalisation pass + unit-row
    table: saved selections (wsel: which edges, exact-tie order included) identical, outputs
    equal to rounding (the kept edges' weights are fast values)."""
    from sngnn_amd import _lib, ops
    from sngnn_amd.graph import Graph
    lib = _lib.load()
    ei = random_graph(n, e, seed=n + c, hubs=hubs).to(cuda)
    h = _rows(n, c, 11 * c + (k or 0), kind).to(cuda)
    h[11] = h[12] * 2.0                        # power-of-two multiple: same unit row
    g = Graph(ei, n, True, k is not None)
    res = {}
    try:
        for mode in (1, 2):                    # table always / on the fly always
            lib.sngnn_tuning_set(2, mode)
            out, wsel, inv, _, _ = ops.aggregate_forward(g, h, k, thr, save_for_backward=True)
            sel = ops.aggregate_forward(g, h, k, thr, want_selection=True)[3] if k else None
            res[mode] = (out, wsel, inv, sel)
    finally:
        lib.sngnn_tuning_set(2, 0)
    kept1, kept2 = res[1][1] > -3.0, res[2][1] > -3.0
    assert torch.equal(kept1, kept2)
    assert torch.equal(res[1][2], res[2][2])                         # inverse norms: same IEEE formula
    if k:
        assert torch.equal(res[1][3], res[2][3])                     # selected sources in rank order
    w1, w2 = res[1][1][kept1], res[2][1][kept2]
    assert (w1 - w2).abs().max() <= 2e-5 if w1.numel() else True     # weights: fast vs exact
    assert (res[1][0] - res[2][0]).abs().max() <= 2e-5 * max(1.0, float(res[1][0].abs().max()))


def test_layer_level_filter_hint(cuda):
    """conv._FilterHint: a layer whose rows give the filter nothing to prune (nearly parallel rows behind a
    threshold they all pass) tells its next forwards not to use it (sngnn_epilogue_t.no_filter); a layer
    whose threshold prunes keeps it.  Either way the same bits."""
    import sngnn_amd
    from sngnn_amd import ops
    from tests.helpers import random_graph
    n, f, c = 3000, 24, 40
    ei = random_graph(n, 30000, seed=5, hubs=((0, 2500), (1, 300))).to(cuda)
    gen = torch.Generator().manual_seed(0)
    x_par = (torch.rand(n, f, generator=gen) * 0.05 + 1.0).to(cuda)         # every cosine ~ 1
    x_rnd = torch.randn(n, f, generator=gen).to(cuda)
    for x, expect in ((x_par, True), (x_rnd, False)):
        torch.manual_seed(1)
        conv = sngnn_amd.SNConv_plus(f, c, n, top_k=16, thr=0.9).to(cuda)
        with torch.no_grad():
            conv.lin.weight.abs_()                                              # keep parallel rows parallel
            g = sngnn_amd.graph.GLOBAL_CACHE.get(ei, n, True, True)
            assert ops.filter_wanted(g, c, 16, 0.9)
            out1 = conv(x, ei)                  # first forward: filter by the knobs; the probe is ENQUEUED
            assert conv._filt_hint.no_filter is False      # (the forward does not wait for its verdict)
            torch.cuda.synchronize()
            out2 = conv(x, ei)                  # second: takes the finished probe's verdict, runs by the hint
            assert conv._filt_hint.no_filter is expect
        assert torch.equal(out1, out2)
        xg = x.clone().requires_grad_(True)
        o = conv(xg, ei)
        o.square().sum().backward()
        assert torch.equal(o.detach(), out1) and torch.isfinite(xg.grad).all()


def test_filter_by_the_graphs_prunable_share_at_thr_zero(cuda):
    """Round 5's rule (agg_fwd.hip:use_filter): with NO pruning threshold the filter is still used for the rows
    above the small class when top_k itself prunes most of the graph's edges (sum of (in-degree - top_k) over the
    rows of >= 32 in-edges is >= 0.62 E': products' degree law, dense uniform graphs; not arxiv's).  Same bits as
    with the filter off - outputs, selections, the kept bits a training forward writes and the gradient - and
    rows shorter than 32 in-edges skip it."""
    from sngnn_amd import _lib, ops, synth
    from sngnn_amd.graph import Graph
    lib = _lib.load()
    k, thr, c = 16, 0.0, 48
    rng = np.random.default_rng(3)
    dense = torch.from_numpy(synth.make_edges(rng, 12000, 600000, 0, uniform=True)).to(cuda)        # in-degree 50
    law = torch.from_numpy(synth.make_edges(rng, 30000, 30000 * 50, 4000)).to(cuda)                  # products' law
    sparse = torch.from_numpy(synth.make_edges(rng, 30000, 30000 * 7, 4000)).to(cuda)                # arxiv's
    short = torch.from_numpy(synth.make_edges(rng, 12000, 12000 * 24, 0, uniform=True)).to(cuda)    # in-degree 24
    for ei, n, want in ((dense, 12000, True), (law, 30000, True), (sparse, 30000, False), (short, 12000, False)):
        g = Graph(ei, n, True, True)
        assert ops.filter_wanted(g, c, k, thr) is want, (n, ei.size(1))
        assert ops.filter_wanted(g, c, k, 0.9)                 # (a pruning threshold: as before)
        assert not ops.filter_wanted(g, 32, k, thr)            # (no filter rows for one-line unit rows)
        if not want:
            continue
        h = _rows(n, c, 7, "normal").to(cuda)
        gout = torch.randn(n, c, generator=torch.Generator().manual_seed(2)).to(cuda)
        res = {}
        for mode in (0, 1):
            lib.sngnn_filter_enable(mode)
            try:
                out, wsel, inv, sel_src, sel_w = ops.aggregate_forward(g, h, k, thr, save_for_backward=True,
                                                                       want_selection=True)
                hg = h.clone().requires_grad_(True)
                o2 = ops.aggregate(hg, g, k, thr)                # the training forward (kept bits where supported)
                o2.backward(gout)
                res[mode] = (out, wsel, inv, sel_src, sel_w, o2.detach(), hg.grad)
            finally:
                lib.sngnn_filter_enable(1)
        for a, b in zip(res[0], res[1]):
            assert torch.equal(a, b)
