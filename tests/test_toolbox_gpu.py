"""GPU parity of the Sim-GFA toolbox kernels against the oracle's restatement of
SimGFAToolbox/dense.py (small variants are the semantic spec)."""
import numpy as np
import pytest
import torch

from oracle import sngnn_oracle as O
from tests.helpers import random_graph

pytestmark = pytest.mark.gpu


def bow(n, f, seed, density=0.05):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(n, f, generator=g) < density).float()
    x[3] = x[4]
    x[7] = 0.0                     # an all-zero row normalises to 0
    return x


@pytest.mark.parametrize("n,f", [(50, 9), (300, 257), (515, 1000), (129, 16)])
def test_cosine_dense_mfma(cuda, n, f):
    from sngnn_amd import toolbox as T
    x = torch.randn(n, f, generator=torch.Generator().manual_seed(n))
    x[2] = 0.0
    S = T.cosine_similarity_dense_small(x.to(cuda)).cpu()
    ref = O.cosine_similarity_dense_small(x)
    assert S.shape == ref.shape
    assert (S - ref).abs().max() < 2e-6
    # asymmetric data check of the tile/epilogue indexing: S[i, j] really is rows i, j
    i, j = 1, n - 2
    want = torch.dot(x[i], x[j]) / (x[i].norm() * x[j].norm())
    assert abs(S[i, j] - want) < 2e-6 and abs(S[j, i] - want) < 2e-6


def test_node_and_linked_similarity(cuda):
    from sngnn_amd import toolbox as T
    n, f = 200, 120
    x = bow(n, f, 1)
    ei = random_graph(n, 1500, 2)
    sim, mean = T.node_similarity_dense_small(x.to(cuda))
    rsim, rmean = O.node_similarity_dense_small(x)
    assert sim.shape == rsim.shape and (sim.cpu() - rsim).abs().max() < 2e-6
    assert abs(float(mean) - float(rmean)) < 1e-6
    lsim, lmean = T.linked_node_similarity_dense_small(x.to(cuda), ei.to(cuda))
    rl, rlm = O.linked_node_similarity_dense_small(x, ei)
    assert lsim.shape == rl.shape and (lsim.cpu() - rl).abs().max() < 2e-6
    assert abs(float(lmean) - float(rlm)) < 1e-6
    # the "large" mean: documented precedence quirk and the corrected value
    _, q = T.node_similarity_dense_large_parted(x.to(cuda))
    _, qr = O.node_similarity_dense_large_parted(x)
    assert abs(float(q) - float(qr)) <= 1e-4 * abs(float(qr))
    _, c = T.node_similarity_dense_large_parted(x.to(cuda), corrected=True)
    assert abs(float(c) - float(rmean)) < 1e-5


def test_neighborhood_similarity(cuda):
    from sngnn_amd import toolbox as T
    n, f = 150, 64
    x = bow(n, f, 3, 0.1)
    ei = random_graph(n, 900, 4)
    ei = ei[:, ei[0] < n - 5]                  # the last nodes have no out-edge
    w, wm = T.neighborhood_similarity_dense_small(x.to(cuda), ei.to(cuda))
    rw, rwm = O.neighborhood_similarity_dense_small(x, ei)
    assert w.shape == rw.shape and (w.cpu() - rw).abs().max() < 2e-6
    assert abs(float(wm) - float(rwm)) < 1e-6
    per_node, mean_all = T.neighborhood_similarity_dense_large(x.to(cuda), ei.to(cuda))
    assert per_node.shape == (n, 1)
    full = torch.zeros(n)
    full[: rw.numel()] = rw
    assert (per_node.cpu().flatten() - full).abs().max() < 2e-6
    assert abs(float(mean_all) - float(full.sum() / n)) < 1e-6


@pytest.mark.parametrize("n,f,c", [(120, 33, 3), (700, 300, 5), (257, 1000, 7)])
def test_class_similarity_without_forming_S(cuda, n, f, c):
    from sngnn_amd import toolbox as T
    x = torch.randn(n, f, generator=torch.Generator().manual_seed(c))
    y = torch.randint(0, c, (n,), generator=torch.Generator().manual_seed(n))
    y[:c] = torch.arange(c)                        # every class present
    mat, mean = T.class_similarity_dense_small(x.to(cuda), y.to(cuda))
    rmat, rmean = O.class_similarity_dense_small(x, y)
    assert mat.shape == rmat.shape
    # block means are O(1/sqrt(F |A||B|)) small: compare on the scale of the largest entry
    tol = 2e-5 * float(rmat.abs().max())
    assert (mat.cpu() - rmat).abs().max() < tol
    assert (T.class_similarity_dense_large(x.to(cuda), y.to(cuda)).cpu() - rmat).abs().max() < tol


@pytest.mark.parametrize("n,f,c", [(40000, 64, 7), (169343, 128, 40)])
def test_class_similarity_at_large_n_against_the_closed_form(cuda, n, f, c):
    """At sizes where S cannot be formed (the second case: config 4's node count, feature width and class
    count - 28.7 G cosines summed by the tile kernel's epilogue): the block mean over a class pair is
    <sum of A's unit rows, sum of B's unit rows> / (|A| |B|), which float64 evaluates in O(N F)."""
    from sngnn_amd import toolbox as T
    x = torch.randn(n, f, generator=torch.Generator().manual_seed(c))
    y = torch.randint(0, c, (n,), generator=torch.Generator().manual_seed(n))
    y[:c] = torch.arange(c)
    mat = T.class_similarity_dense_large(x.to(cuda), y.to(cuda)).cpu().double()
    xn = torch.nn.functional.normalize(x, dim=1).double()          # (the reference normalises in fp32)
    sums = torch.zeros(c, f, dtype=torch.float64).index_add_(0, y, xn)
    cnt = torch.bincount(y, minlength=c).double()
    want = (sums @ sums.t()) / (cnt[:, None] * cnt[None, :])
    assert mat.shape == want.shape
    # a block mean is a sum of |A||B| cosines of either sign divided by their number: judged on the scale of
    # the diagonal blocks (which hold the |A| ones of the self pairs) and of fp32 sums of that many terms
    assert float((mat - want).abs().max()) <= 2e-5 * float(want.abs().max()) + 1e-7


def test_graph_statistics_at_config4_size(cuda):
    """The "large" statistics (dense.py:9-101) on config 4's graph and feature shape, where the reference
    loops over 1 000-row blocks / over every node in Python: node similarity against the closed form
    |sum of unit rows|^2 in float64; linked-node and neighbourhood similarity against a float64 per-edge
    cosine and its group means."""
    from sngnn_amd import synth
    from sngnn_amd import toolbox as T
    d = synth.make_dataset("arxiv")
    x, ei = d.x, d.edge_index
    n = x.size(0)
    xn = torch.nn.functional.normalize(x, dim=1).double()
    # node similarity: the reference's value (its precedence slip included) and the corrected mean
    total = float(xn.sum(0).square().sum())
    _, quirk = T.node_similarity_dense_large_parted(x.to(cuda))
    _, mean = T.node_similarity_dense_large_parted(x.to(cuda), corrected=True)
    assert abs(float(quirk) - (total - n) / (n - 1) * n) <= 2e-5 * abs((total - n) / (n - 1) * n) + 1e-3
    assert abs(float(mean) - (total - n) / (n * (n - 1.0))) <= 2e-5 * abs(total / (n * (n - 1.0))) + 1e-9
    # per-edge cosines, listed by (src, dst) order; means per source node
    order = torch.argsort(ei[0] * n + ei[1], stable=True)
    es = ei[:, order]
    cos = (xn[es[0]] * xn[es[1]]).sum(1)
    sim, m = T.linked_node_similarity_dense_large(x.to(cuda), ei.to(cuda))
    assert float((sim.cpu().double().flatten() - cos).abs().max()) <= 2e-6
    assert abs(float(m) - float(cos.mean())) <= 2e-6
    per_node, nm = T.neighborhood_similarity_dense_large(x.to(cuda), ei.to(cuda))
    cnt = torch.bincount(es[0], minlength=n).double()
    want = torch.zeros(n, dtype=torch.float64).index_add_(0, es[0], cos) / cnt.clamp_min(1)
    assert float((per_node.cpu().double().flatten() - want).abs().max()) <= 5e-6
    assert abs(float(nm) - float(want.sum() / n)) <= 2e-6


def test_toolbox_argument_errors(cuda):
    from sngnn_amd import toolbox as T
    x = torch.randn(10, 4)
    with pytest.raises(ValueError, match="GPU"):
        T.cosine_similarity_dense_small(x)
    bad = torch.tensor([[0, 1], [2, 99]])
    with pytest.raises(ValueError, match="outside"):
        T.edge_cosine(x.to(cuda), bad.to(cuda))


def test_sparse_variants_against_scipy_sklearn(cuda):
    """SimGFAToolbox/sparse.py semantics: column-normalised M^T M on the adjacency
    (toolbox-example.py:28-29), checked against scipy/sklearn, the libraries the
    reference itself calls (sparse.py:13-14)."""
    import scipy.sparse as sp
    import sklearn.preprocessing as pp
    from sngnn_amd import toolbox as T
    n = 300
    ei = random_graph(n, 2500, 8)
    adj = sp.csc_matrix((np.ones(ei.size(1)), (ei[0].numpy(), ei[1].numpy())), shape=(n, n))
    coln = pp.normalize(adj.tocsc(), axis=0)
    ref = torch.from_numpy((coln.T * coln).toarray()).float()
    sim = T.cosine_similarity_sparse(adj, device=cuda)
    assert sim.shape == (n, n) and (sim.cpu() - ref).abs().max() < 2e-6
    allsim, mean = T.node_similarity_sparse(adj, device=cuda)
    assert allsim.shape == (n * n, 1) and abs(float(mean) - float(ref.mean())) < 1e-6
    lv, lm = T.linked_node_similarity_sparse(adj, ei, device=cuda)
    want = ref[ei[0], ei[1]]
    assert (lv.cpu().flatten() - want).abs().max() < 2e-6 and abs(float(lm) - float(want.mean())) < 1e-6
    pn, pm = T.neighborhood_similarity_sparse(adj, ei, device=cuda)
    full = torch.zeros(n)
    cnt = torch.bincount(ei[0], minlength=n)
    full.index_add_(0, ei[0], want)
    full = torch.where(cnt > 0, full / cnt.clamp(min=1), torch.zeros(n))
    assert (pn.cpu().flatten() - full).abs().max() < 2e-6 and abs(float(pm) - float(full.sum() / n)) < 1e-6
    y = torch.randint(0, 4, (n,), generator=torch.Generator().manual_seed(1))
    y[:4] = torch.arange(4)
    cm = T.class_similarity_sparse(adj, y, device=cuda).cpu()
    for a in range(4):
        for b in range(4):
            blk = ref[y == a][:, y == b]
            assert abs(float(cm[a, b]) - float(blk.mean())) < 1e-6


def test_signed_edge_attention_ggcn(cuda):
    """models.py:1512-1519 with core torch as the checker (F.cosine_similarity is what
    the reference calls there)."""
    import torch.nn.functional as F
    from sngnn_amd import toolbox as T
    from tests.helpers import random_graph
    n, c = 700, 64
    wh = torch.randn(n, c, generator=torch.Generator().manual_seed(8))
    idx = random_graph(n, 9000, seed=2)
    sim = F.cosine_similarity(wh[idx[0]], wh[idx[1]])
    pos, neg = T.signed_edge_attention(idx.to(cuda), wh.to(cuda))
    np.testing.assert_allclose(pos.cpu().numpy(), F.relu(sim).numpy(), atol=2e-6)
    np.testing.assert_allclose(neg.cpu().numpy(), (-F.relu(-sim)).numpy(), atol=2e-6)
    assert (pos >= 0).all() and (neg <= 0).all() and ((pos == 0) | (neg == 0)).all()


@pytest.mark.parametrize("n,f,k,excl", [(300, 16, 5, True), (1000, 64, 16, True), (777, 33, 32, False),
                                        (130, 7, 10, True), (5, 8, 8, True), (1, 4, 3, True),
                                        (2500, 128, 16, True), (900, 96, 7, True), (640, 64, 31, False),
                                        (513, 32, 1, True), (2277, 200, 10, True), (1030, 51, 32, True)])
@pytest.mark.parametrize("route", [0, 1, 2])
def test_knn_graph_matches_dense_topk(cuda, n, f, k, excl, route):
    """The kNN builder against the materialised similarity (fp64): every returned neighbour's
    cosine is right, the list is ordered, and nothing clearly better was left out (pairs closer
    than 2e-6 may swap).  route (sngnn_tuning_set(6, .)): 0 = by shape (default), 1 = the fused
    MFMA scan with its per-row top-k epilogue, 2 = dense cosine kernel + row selection (what
    small graphs with wide features take by default)."""
    from sngnn_amd import _lib
    from sngnn_amd import toolbox as T
    gen = torch.Generator().manual_seed(n + k)
    x = torch.randn(n, f, generator=gen)
    if n > 12:
        x[7] = x[3]                       # exact duplicates: tie broken by node id
        x[11] = 0.0                       # a zero row: cosine 0 with everything
    try:
        _lib.load().sngnn_tuning_set(6, route)
        idx, sim = T.knn_graph(x.to(cuda), k, exclude_self=excl)
    finally:
        _lib.load().sngnn_tuning_set(6, 0)
    idx, sim = idx.cpu(), sim.cpu()
    xn = torch.nn.functional.normalize(x.double(), dim=1)
    S = xn @ xn.t()
    if excl:
        S.fill_diagonal_(-float("inf"))
    m = min(k, n - (1 if excl else 0))
    assert (idx[:, m:] == -1).all() and (sim[:, m:] == 0).all()
    if m == 0:
        return
    got, gs = idx[:, :m], sim[:, :m]
    assert (got >= 0).all() and (got < n).all()
    assert all(len(set(r.tolist())) == m for r in got)                    # no duplicates
    if excl:
        assert (got != torch.arange(n).unsqueeze(1)).all()
    true = S.gather(1, got)
    assert (gs.double() - true).abs().max() <= 2e-6
    assert (gs[:, 1:] <= gs[:, :-1] + 1e-7).all()                          # rank order
    kth = torch.topk(S, m, dim=1).values[:, -1]
    assert (true.min(dim=1).values >= kth - 2e-6).all()                    # nothing better missed
    if n > 12 and excl and k >= 2:
        # node 3 and node 7 are identical: each is the other's nearest neighbour
        assert got[3, 0] == 7 and got[7, 0] == 3


@pytest.mark.parametrize("n,f,k", [(40000, 64, 16), (169343, 128, 16)])
def test_knn_graph_at_large_n_on_sampled_rows(cuda, n, f, k):
    """Above KNN_DENSE_MAX_N the builder is the fused MFMA scan with its per-row top-k epilogue (N^2 scores
    never stored; the second case is BASELINE config 4's node count and feature width: 7.3 TFLOP of
    cosines).  300 sampled rows (the planted duplicate and zero rows among them) against a float64 scan of
    all N candidates: every returned cosine right, rank order, nothing better than the k-th left out."""
    from sngnn_amd import toolbox as T
    gen = torch.Generator().manual_seed(n)
    x = torch.randn(n, f, generator=gen)
    x[7] = x[3]
    x[11] = 0.0
    idx, sim = T.knn_graph(x.to(cuda), k, exclude_self=True)
    idx, sim = idx.cpu(), sim.cpu()
    assert (idx >= 0).all() and (idx < n).all()
    rows = torch.cat([torch.tensor([3, 7, 11, 0, n - 1]), torch.randint(0, n, (295,), generator=gen)])
    xn = torch.nn.functional.normalize(x.double(), dim=1)
    S = xn[rows] @ xn.t()                                   # [300, N] in float64
    S[torch.arange(rows.numel()), rows] = -float("inf")     # exclude_self
    got, gs = idx[rows], sim[rows]
    assert (got != rows.unsqueeze(1)).all()
    assert all(len(set(r.tolist())) == k for r in got)
    true = S.gather(1, got)
    assert (gs.double() - true).abs().max() <= 2e-6
    assert (gs[:, 1:] <= gs[:, :-1] + 1e-7).all()
    kth = torch.topk(S, k, dim=1).values[:, -1]
    assert (true.min(dim=1).values >= kth - 2e-6).all()
    assert got[0, 0] == 7 and got[1, 0] == 3                # the duplicates find each other first


def test_knn_edge_index_feeds_the_conv_layer(cuda):
    import sngnn_amd
    from sngnn_amd import toolbox as T
    x = torch.randn(400, 24, generator=torch.Generator().manual_seed(1)).to(cuda)
    ei = T.knn_edge_index(x, 8)
    assert ei.shape == (2, 400 * 8) and ei.dtype == torch.int64
    assert torch.equal(ei[1], torch.arange(400, device=cuda).repeat_interleave(8))
    conv = sngnn_amd.SNConv_plus(24, 16, 400, top_k=4, thr=0.0).to(cuda)
    out = conv(x, ei)
    assert out.shape == (400, 16) and torch.isfinite(out).all()
    with pytest.raises(ValueError):
        T.knn_graph(x, 33)


def test_linked_node_similarity_dense_large(cuda):
    """SimGFAToolbox/dense.py:33-62: the linked-pair similarities listed by source node
    (edges sorted by (src, dst), PyG sort_edge_index) equal dense.py:152-155's values in that
    order; the mean is the mean over the linked pairs."""
    from sngnn_amd import toolbox as T
    n, f = 400, 37
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, f, generator=gen)
    x[3] = 0.0
    ei = random_graph(n, 3000, 12)
    ei = ei[:, torch.randperm(ei.size(1), generator=gen)]           # unsorted input
    vals, mean = T.linked_node_similarity_dense_large(x.to(cuda), ei.to(cuda))
    key = ei[0] * n + ei[1]
    order = torch.argsort(key, stable=True)
    ref_all, ref_mean = O.linked_node_similarity_dense_small(x, ei[:, order])
    assert vals.shape == (ei.size(1), 1)
    assert (vals.cpu() - ref_all).abs().max() < 2e-6
    assert abs(float(mean) - float(ref_mean)) < 1e-6


def test_sparse_statistics_never_form_the_product(cuda):
    """A [300 000 x 300 000] adjacency (9e10 entries: nothing dense fits): the linked /
    neighbourhood / class statistics of SimGFAToolbox/sparse.py:45-152 stay sparse on the GPU.
    Checked against scipy on sampled pairs and against the same code's own small-N results
    being exact (previous test)."""
    import scipy.sparse as sp
    import sklearn.preprocessing as pp
    from sngnn_amd import toolbox as T
    n = 300_000
    rng = np.random.default_rng(4)
    src = rng.integers(0, n, size=2_000_000)
    dst = (src + rng.integers(-50, 50, size=src.size)) % n          # neighbours share neighbours
    adj = sp.csc_matrix((np.ones(src.size), (src, dst)), shape=(n, n))
    ei = torch.from_numpy(np.stack([src[:50_000], dst[:50_000]]))
    lv, lm = T.linked_node_similarity_sparse(adj, ei, device=cuda)
    coln = pp.normalize(adj, axis=0)
    pick = rng.integers(0, 50_000, size=300)
    for p in pick:
        a, b = int(ei[0, p]), int(ei[1, p])
        want = float(coln[:, a].multiply(coln[:, b]).sum())
        assert abs(float(lv[p]) - want) < 2e-6, (a, b)
    assert 0.0 <= float(lm) <= 1.0
    pn, pm = T.neighborhood_similarity_sparse(adj, ei, device=cuda)
    assert pn.shape == (n, 1) and abs(float(pm) - float(pn.sum()) / n) < 1e-6
    y = torch.from_numpy(rng.integers(0, 3, size=n))
    cm = T.class_similarity_sparse(adj, y, device=cuda).cpu()
    # <m_A, m_B> against scipy: m_A = coln @ onehot_A
    oh = sp.csc_matrix((np.ones(n), (np.arange(n), y.numpy())), shape=(n, 3))
    m = (coln @ oh).toarray()                                        # [rows, 3]
    want = torch.from_numpy(m.T @ m) / (torch.bincount(y, minlength=3).double()[:, None] *
                                        torch.bincount(y, minlength=3).double()[None, :])
    assert (cm.double() - want).abs().max() <= 1e-5 * want.abs().max()
    # fixed-order sums (no float atomics): a second run gives the same BITS - column norms and class sums
    sp1, sp2 = T._SparseCols(adj, cuda), T._SparseCols(adj, cuda)
    assert torch.equal(sp1.vals, sp2.vals)
    assert torch.equal(cm, T.class_similarity_sparse(adj, y, device=cuda).cpu())
    with pytest.raises(ValueError, match="does not fit"):
        T.cosine_similarity_sparse(adj, device=cuda)


# ---------------------------------------------------------------------------
# a13-a15 against fixtures made by the reference's OWN SimGFAToolbox/dense.py and sparse.py
# (tests/golden/pin_reference.py:pin_toolbox; inputs + the reference's outputs)
# ---------------------------------------------------------------------------
import os

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
COS_TOL = 2e-6          # fp32 cosine: |s| <= 1, a few ulps of accumulated rounding in the dot products


def _fx(name, cuda):
    z = np.load(os.path.join(GOLDEN, name))
    dev = {k: torch.from_numpy(z[k]).to(cuda) for k in ("x", "edge_index", "y") if k in z}
    return z, dev


def _close(got, want, tol=COS_TOL):
    got = torch.as_tensor(got).detach().cpu().double().numpy()
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    assert np.abs(got - want).max() <= tol, np.abs(got - want).max()


def test_dense_small_variants_against_reference_fixture(cuda):
    """dense.py:138-179 run by the reference's own file -> toolbox_dense_small.npz."""
    from sngnn_amd import toolbox as T
    z, d = _fx("toolbox_dense_small.npz", cuda)
    x, ei, y = d["x"], d["edge_index"], d["y"]
    _close(T.cosine_similarity_dense_small(x), z["cosine"])
    sim, mean = T.node_similarity_dense_small(x)
    _close(sim, z["node_sim"]), _close(mean, z["node_mean"], 1e-6)
    sim, mean = T.linked_node_similarity_dense_small(x, ei)
    _close(sim, z["linked_sim"]), _close(mean, z["linked_mean"], 1e-6)
    w, mean = T.neighborhood_similarity_dense_small(x, ei)
    _close(w, z["nbr_weight"]), _close(mean, z["nbr_mean"], 1e-6)
    mat, mean = T.class_similarity_dense_small(x, y)
    _close(mat, z["class_mat"], 1e-6), _close(mean, z["class_mean"], 1e-6)


@pytest.mark.parametrize("name", ["toolbox_dense_small.npz", "toolbox_dense_parted.npz"])
def test_dense_large_variants_against_reference_fixture(cuda, name):
    """dense.py:9-130 (Python row / block loops in the reference; 1 100 rows = two of its
    1000-row blocks) -> the same statistics from the O(NF) / per-edge kernels."""
    from sngnn_amd import toolbox as T
    z, d = _fx(name, cuda)
    x, ei, y = d["x"], d["edge_index"], d["y"]
    _, q = T.node_similarity_dense_large_parted(x)
    assert abs(float(q) - float(z["parted_mean"])) <= 1e-5 * abs(float(z["parted_mean"]))
    sim, mean = T.linked_node_similarity_dense_large(x, ei)
    _close(sim, z["linked_large_sim"]), _close(mean, z["linked_large_mean"], 1e-6)
    per_node, mean_all = T.neighborhood_similarity_dense_large(x, ei)
    _close(per_node, z["nbr_large_sim"]), _close(mean_all, z["nbr_large_mean"], 1e-6)
    _close(T.class_similarity_dense_large(x, y), z["class_large_mat"], 1e-6)
    _, c = T.node_similarity_dense_large_parted(x, corrected=True)
    _close(c, z["node_mean"], 1e-6)


def test_sparse_variants_against_reference_fixture(cuda):
    """sparse.py:8-152 run by the reference's own file on real scipy + scikit-learn, on the
    adjacency built as SimGFAToolbox/utils.py:5-11 builds it (toolbox-example.py:28-29)."""
    from sngnn_amd import toolbox as T
    z, d = _fx("toolbox_sparse_adj.npz", cuda)
    ei, y = d["edge_index"], d["y"]
    n = int(z["n"][0])
    adj = T.edge_index_to_sparse_csc_tensor(torch.empty(n, 1), ei)
    _close(T.cosine_similarity_sparse(adj, device=cuda), z["cosine"])
    allsim, mean = T.node_similarity_sparse(adj, device=cuda)
    _close(allsim, z["node_sim"]), _close(mean, z["node_mean"], 1e-6)
    lv, lm = T.linked_node_similarity_sparse(adj, ei, device=cuda)
    _close(lv, z["linked_sim"]), _close(lm, z["linked_mean"], 1e-6)
    pn, pm = T.neighborhood_similarity_sparse(adj, ei, device=cuda)
    _close(pn, z["nbr_sim"]), _close(pm, z["nbr_mean"], 1e-6)
    _close(T.class_similarity_sparse(adj, y, device=cuda), z["class_mat"], 1e-6)


@pytest.mark.parametrize("n,f", [(40, 4), (130, 16), (300, 28), (1500, 200), (700, 2325 % 200 + 7)])
def test_cosine_dense_narrow_inputs_and_short_last_split(cuda, n, f):
    """Shapes whose K range is narrower than one 32-float K-step somewhere: ld < 32, or a last
    contraction split shorter than 32 (N <= 1664 with F = 200: k_begin 192 of ld 200).  The
    prefetch of an out-of-range K-step must stay inside the row (it ran up to 112
    bytes past the end of x for the clamped last row)."""
    from sngnn_amd import toolbox as T
    x = torch.randn(n, f, generator=torch.Generator().manual_seed(n + f))
    x[1] = 0.0
    S = T.cosine_similarity_dense_small(x.to(cuda)).cpu()
    ref = O.cosine_similarity_dense_small(x)
    assert (S - ref).abs().max() < 2e-6


def test_class_sums_are_deterministic_and_atomics_free(cuda):
    """The class statistics add in a fixed order: two runs give the same bits; classes that are
    tiny, that span many 256-row segments, that start exactly on a segment boundary, and an
    empty class."""
    from sngnn_amd import toolbox as T
    n, f = 5000, 70
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, f, generator=g)
    y = torch.empty(n, dtype=torch.int64)
    sizes = [256, 3, 1, 2000, 252, 1024, 0, 1464]            # class 6 is empty
    assert sum(sizes) == n
    y[torch.randperm(n, generator=g)] = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes))
    a, da = T.class_block_sums(x.to(cuda), y.to(cuda), len(sizes))
    b, db = T.class_block_sums(x.to(cuda), y.to(cuda), len(sizes))
    assert torch.equal(a, b) and torch.equal(da, db)
    xn = torch.nn.functional.normalize(x.double(), dim=1)
    m = torch.zeros(len(sizes), f, dtype=torch.float64).index_add_(0, y, xn)
    want = m @ m.t()
    assert (a.cpu() - want).abs().max() <= 1e-6 * want.abs().max()
    assert abs(float(da) - n) < 1e-3


@pytest.mark.parametrize("n,f", [(700, 128), (1500, 1433), (2277, 200), (300, 36)])
def test_cosine_dense_on_the_bf16_cores_rounds_like_fp32(cuda, n, f):
    """sngnn_tuning_set(5, mode): the dense cosine's products on the bf16 matrix cores after an exact
    three-way split of both panels (default) or with fp32 MFMAs - against float64 the split form is
    no worse than the fp32 form, and both are far inside the fp32 cosine tolerance."""
    from sngnn_amd import _lib, toolbox as T
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + f)
    x = torch.randn(n, f, generator=g)
    x[:, ::5] *= 2.0 ** -7                                    # a dynamic range inside the rows
    x[3] = x[4] * 3.0                                         # cosine exactly 1
    x[9] = 0.0
    xn = torch.nn.functional.normalize(x.double(), dim=1)
    ref = xn @ xn.t()
    out = {}
    try:
        for mode in (0, 1):
            lib.sngnn_tuning_set(5, mode)
            out[mode] = T.cosine_similarity_dense_small(x.to(cuda)).cpu().double()
    finally:
        lib.sngnn_tuning_set(5, 0)
    err = {m: float((out[m] - ref).abs().max()) for m in out}
    assert err[0] <= COS_TOL and err[1] <= COS_TOL, err
    assert err[0] <= 1.5 * err[1] + 2.0 ** -24, err
    assert float((out[0] - out[0].t()).abs().max()) <= 2.0 ** -22          # (mirror tiles: same bits; diagonal tiles: to rounding)
    assert (out[0][9] == 0).all()


def test_segment_mean_is_the_cpu_scatter_mean_bit_for_bit(cuda):
    """``sngnn_segment_mean`` (dense.py:163's ``scatter_mean(sim, edge_index[0], dim=0)``): every
    group's values added in ENTRY order in fp32, divided by max(count, 1) - equal to the oracle's
    serial scatter bit for bit on an UNSORTED index with hubs, empty groups and trailing empty
    groups, and the same bits on every run (no atomics)."""
    from sngnn_amd import toolbox as T
    g = torch.Generator().manual_seed(21)
    e, m = 200_000, 5000
    index = torch.randint(0, m - 40, (e,), generator=g)          # the last 40 groups are empty
    index[index == 17] = 18                                      # an empty group in the middle
    index[torch.randperm(e, generator=g)[:30_000]] = 99          # a 30 000-entry hub
    val = torch.randn(e, generator=g)
    want = O.scatter_mean(val, index, m)
    cnt_want = torch.bincount(index, minlength=m)
    got, cnt = T.segment_mean(val.to(cuda), index.to(cuda), m)
    assert torch.equal(got.cpu(), want)
    assert torch.equal(cnt.cpu().long(), cnt_want)
    again, _ = T.segment_mean(val.to(cuda), index.to(cuda), m)
    assert torch.equal(got, again)
    assert float(got[17]) == 0.0 and bool((got[m - 40:] == 0).all())
    # degenerate shapes
    z, c = T.segment_mean(torch.empty(0, device=cuda), torch.empty(0, dtype=torch.long, device=cuda), 7)
    assert bool((z == 0).all()) and bool((c == 0).all())
    with pytest.raises(ValueError):
        T.segment_mean(val[:10].to(cuda), torch.full((10,), m, dtype=torch.long, device=cuda), m)


def test_neighborhood_similarity_mean_matches_the_reference_order_exactly(cuda):
    """dense.py:158-164 end to end: given the SAME per-edge cosines, the kernel's mean by source is
    the oracle's ``scatter_mean`` bit for bit (the cosines themselves differ in the last ulp between
    the CPU's and the GPU's dot products, hence the tolerance in test_neighborhood_similarity)."""
    from sngnn_amd import toolbox as T
    n, f = 400, 48
    x = bow(n, f, 9, 0.15)
    ei = random_graph(n, 6000, 11, hubs=((3, 350),))
    ei = ei.flip(0)                              # the hub as a SOURCE; edges now unsorted by source
    sim = T.edge_cosine(x.to(cuda), ei.to(cuda))
    length = int(ei[0].max()) + 1
    got, _ = T.segment_mean(sim, ei[0].to(cuda), length)
    assert torch.equal(got.cpu(), O.scatter_mean_1d(sim.cpu(), ei[0]))
