"""CPU: the oracle against its pins - the hand-derived KATs,
the committed golden vectors, the literal-loop forms, and the independent C
restatement."""
import glob
import json
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from oracle import c_oracle as CO
from oracle import sngnn_oracle as O
from tests.helpers import random_graph

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_kat_appendix_b_both_oracles():
    kat = json.load(open(os.path.join(GOLDEN, "kat_appendix_b.json")))
    h = torch.tensor(kat["h"], dtype=torch.float32)
    ei = torch.tensor(kat["edge_index"], dtype=torch.int64)
    s = O.edge_cosine(torch.nn.functional.normalize(h, dim=-1), ei)
    np.testing.assert_allclose(s.numpy(), kat["edge_cosine"], atol=1e-7)
    for case in kat["cases"]:
        for use_loop in (False, True):
            r = O.aggregate_reference(h, ei, add_loops=True, remove_loops=case["remove_loops"],
                                      top_k=case["top_k"], thr=case["thr"], use_loop=use_loop)
            np.testing.assert_allclose(r["out"].numpy(), case["out"], atol=1e-6)
            if case["top_k"] is not None:
                assert r["sel_src"].tolist() == case["sel_src"]
        c = CO.aggregate(h.numpy(), ei.numpy(), add_loops=True, remove_loops=case["remove_loops"],
                         top_k=case["top_k"], thr=case["thr"])
        np.testing.assert_allclose(c["out"], case["out"], atol=1e-6)
        if case["top_k"] is not None:
            assert c["sel_src"].tolist() == case["sel_src"]


def test_tie_break_first_occurrence_and_self_loop_last():
    """Appendix B tie case: identical neighbours -> lower edge position first; with
    loops kept, a neighbour identical to the target beats the appended self-loop."""
    h = torch.tensor([[1., 2.], [3., 1.], [3., 1.], [1., 2.]])
    ei = torch.tensor([[1, 2, 3], [0, 0, 0]])          # 1->0, 2->0 tie; 3 identical to 0
    r = O.aggregate_reference(h, ei, add_loops=True, remove_loops=False, top_k=2, thr=-1.0)
    assert r["sel_src"][0].tolist() == [3, 0]           # cos 1.0: node 3 (pos 2) before loop (pos 3)
    r = O.aggregate_reference(h, ei, add_loops=True, remove_loops=True, top_k=2, thr=-1.0)
    assert r["sel_src"][0].tolist() == [3, 1]           # then the tie 1 vs 2 -> node 1


def test_threshold_compare_is_fp32():
    # models.py:152,257 compare an fp32 tensor with the Python float thr
    t = torch.tensor([0.9], dtype=torch.float32)
    assert bool((t >= 0.9).item()) and not (float(t.item()) >= 0.9 + 1e-8)
    s = torch.tensor([0.9, 0.5], dtype=torch.float32)
    w, _ = O.topk_threshold_weights(s, torch.tensor([0, 0]), 2, 0.9)
    assert w.tolist() == [pytest.approx(0.9), 0.0]


def test_mean_divides_by_full_in_degree_and_isolated_rows_are_zero():
    h = torch.tensor([[1., 0.], [1., 0.], [0., 1.], [5., 5.]])
    ei = torch.tensor([[1, 2], [0, 0]])
    r = O.aggregate_reference(h, ei, add_loops=True, remove_loops=True, top_k=1, thr=0.5)
    assert r["out"][0].tolist() == [0.5, 0.0]           # one kept edge, divided by 2
    assert r["out"][3].tolist() == [0.0, 0.0]


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "agg_*.npz"))))
def test_golden_vectors_reproduce(path):
    z = np.load(path)
    add, rem, k = (int(v) for v in z["params"])
    k = None if k < 0 else k
    thr = float(z["thr"][0])
    h = torch.from_numpy(z["h"]).requires_grad_(True)
    ei = torch.from_numpy(z["edge_index"])
    r = O.aggregate_reference(h, ei, add_loops=bool(add), remove_loops=bool(rem), top_k=k, thr=thr)
    assert torch.equal(r["ei"], torch.from_numpy(z["ei_prime"]))
    np.testing.assert_array_equal(r["out"].detach().numpy(), z["out"])
    (r["out"] * torch.from_numpy(z["gout"])).sum().backward()
    np.testing.assert_allclose(h.grad.numpy(), z["grad_h"], rtol=1e-6, atol=1e-7)
    c = CO.aggregate(z["h"], z["edge_index"], add_loops=bool(add), remove_loops=bool(rem),
                     top_k=k, thr=thr)
    np.testing.assert_array_equal(c["ei"], z["ei_prime"])
    np.testing.assert_allclose(c["out"], z["out"], rtol=1e-5, atol=2e-6)
    if k is not None:
        np.testing.assert_array_equal(r["sel_src"].numpy(), z["sel_src"])
        np.testing.assert_array_equal(c["sel_src"], z["sel_src"])


@settings(max_examples=25, deadline=None)
@given(n=st.integers(2, 40), e=st.integers(1, 200), c=st.integers(1, 9), k=st.integers(0, 6),
       thr=st.sampled_from([-1.5, -0.2, 0.0, 0.3, 0.9]), rem=st.booleans(),
       seed=st.integers(0, 10_000))
def test_vectorised_scatter_max_equals_literal_loop(n, e, c, k, thr, rem, seed):
    ei = random_graph(n, e, seed)
    h = torch.randn(n, c, generator=torch.Generator().manual_seed(seed))
    h[0] = h[1]
    a = O.aggregate_reference(h, ei, add_loops=True, remove_loops=rem, top_k=k, thr=thr)
    b = O.aggregate_reference(h, ei, add_loops=True, remove_loops=rem, top_k=k, thr=thr,
                              use_loop=True)
    assert torch.equal(a["out"], b["out"]) and torch.equal(a["sel_src"], b["sel_src"])
    # properties of the operator itself
    deg = torch.bincount(a["ei"][1], minlength=n)
    assert torch.equal((a["sel_src"] >= 0).sum(1) <= torch.minimum(deg, torch.tensor(k)),
                       torch.ones(n, dtype=torch.bool))
    kept = a["weight"] != 0
    assert bool((a["s"][kept] >= torch.tensor(thr, dtype=torch.float32)).all())
    assert bool((a["out"][deg == 0] == 0).all())


def test_c_oracle_agrees_with_torch_oracle_on_selection_and_sums():
    ei = random_graph(500, 6000, 3, hubs=((0, 400),))
    h = torch.randn(500, 12, generator=torch.Generator().manual_seed(0))
    h[3] = h[4]
    for k, thr, rem in ((None, 0.0, False), (1, 0.0, True), (8, 0.2, True), (16, -1.5, False)):
        a = O.aggregate_reference(h, ei, add_loops=True, remove_loops=rem, top_k=k, thr=thr)
        b = CO.aggregate(h.numpy(), ei.numpy(), add_loops=True, remove_loops=rem, top_k=k, thr=thr)
        np.testing.assert_allclose(b["out"], a["out"].numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(b["s"], a["s"].numpy(), atol=1e-6)
        if k is not None:
            np.testing.assert_array_equal(b["sel_src"], a["sel_src"].numpy())


def test_adjacency_branch_and_row_min_quirk():
    n = 30
    ei = random_graph(n, 150, 1)
    ei = ei[:, ei[0] >= 2]                               # row.min() == 2 (models.py:125)
    eip = O.sn_edge_list(ei, n, True, True)
    W = torch.randn(4, n, generator=torch.Generator().manual_seed(1))
    b = torch.randn(4, generator=torch.Generator().manual_seed(2))
    out = O.adj_linear_reference(W, b, eip, n)
    want = b.repeat(n, 1)
    m = int(eip[0].min())
    for s_, d_ in eip.t().tolist():
        want[s_ - m] += W[:, d_]
    np.testing.assert_allclose(out.numpy(), want.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(CO.adj_linear(W.numpy(), b.numpy(), eip.numpy(), n), out.numpy(),
                               rtol=1e-5, atol=1e-5)


def test_toolbox_small_variants():
    x = torch.randn(50, 9, generator=torch.Generator().manual_seed(4))
    y = torch.randint(0, 3, (50,), generator=torch.Generator().manual_seed(5))
    ei = random_graph(50, 300, 6)
    S = O.cosine_similarity_dense_small(x)
    np.testing.assert_allclose(S.numpy(), CO.cosine_dense(x.numpy()), atol=1e-6)
    sim, mean = O.node_similarity_dense_small(x)
    assert sim.numel() == 50 * 49
    np.testing.assert_allclose(float(mean), float((S.sum() - S.diag().sum()) / (50 * 49)), rtol=1e-5)
    lsim, lmean = O.linked_node_similarity_dense_small(x, ei)
    np.testing.assert_allclose(lsim.flatten().numpy(), S[ei[0], ei[1]].numpy())
    w, wm = O.neighborhood_similarity_dense_small(x, ei)
    assert w.numel() == int(ei[0].max()) + 1
    cm, _ = O.class_similarity_dense_small(x, y)
    np.testing.assert_allclose(cm.numpy(), O.class_similarity_dense_large(x, y).numpy(), atol=1e-6)


def test_attention_restatement_known_answers():
    """models.py:393-405.  Hand-derived: two orthogonal unit rows and their sum."""
    h = torch.tensor([[1.0, 0.0], [0.0, 1.0], [1.0, 1.0]])
    ei = torch.tensor([[0, 1, 2, 2], [2, 2, 2, 0]])          # (2,2) is an ORIGINAL loop
    res = O.attention_reference(h, ei)
    # loop removed first, then one loop per node appended at the end
    assert res["ei"].tolist() == [[0, 1, 2, 0, 1, 2], [2, 2, 0, 0, 1, 2]]
    r = 2 ** -0.5
    np.testing.assert_allclose(res["s"].numpy(), [r, r, r, 1, 1, 1], rtol=1e-6)
    e, o = np.exp(r), np.exp(1.0)
    alpha = [e / (2 * e + o), e / (2 * e + o), e / (e + o), o / (e + o), 1.0, o / (2 * e + o)]
    np.testing.assert_allclose(res["alpha"].numpy(), alpha, rtol=1e-6)
    out2 = (alpha[0] * h[0] + alpha[1] * h[1] + alpha[5] * h[2]).numpy()
    np.testing.assert_allclose(res["out"][2].numpy(), out2, rtol=1e-6)
    np.testing.assert_allclose(res["out"][1].numpy(), [0.0, 1.0])
    # the segment softmax equals torch.softmax applied group by group
    gen = torch.Generator().manual_seed(0)
    s = torch.rand(200, generator=gen) * 2 - 1
    idx = torch.randint(0, 17, (200,), generator=gen)
    got = O.segment_softmax(s, idx, 17)
    for i in range(17):
        m = idx == i
        if m.any():
            np.testing.assert_allclose(got[m].numpy(), torch.softmax(s[m], 0).numpy(), rtol=2e-6)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden",
                                                                "attn_*.npz"))))
def test_attention_golden_vectors_reproduce(path):
    z = np.load(path)
    res = O.attention_reference(torch.from_numpy(z["h"]), torch.from_numpy(z["edge_index"]))
    assert np.array_equal(res["ei"].numpy(), z["ei_prime"])
    np.testing.assert_allclose(res["out"].numpy(), z["out"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(res["alpha"].numpy(), z["alpha"], rtol=1e-6)


# ---------------------------------------------------------------------------
# Sim-GFA toolbox: the oracle's restatement against fixtures that hold the outputs of the
# reference's OWN SimGFAToolbox/dense.py and sparse.py (tests/golden/pin_reference.py)
# ---------------------------------------------------------------------------
def _z(name):
    return np.load(os.path.join(GOLDEN, name))


def _t(a):
    return torch.from_numpy(np.asarray(a))


def test_oracle_toolbox_dense_small_fixture_bit_exact():
    z = _z("toolbox_dense_small.npz")
    x, ei, y = _t(z["x"]), _t(z["edge_index"]), _t(z["y"])
    assert torch.equal(O.cosine_similarity_dense_small(x), _t(z["cosine"]))
    for got, key in zip(O.node_similarity_dense_small(x), ("node_sim", "node_mean")):
        assert torch.equal(got, _t(z[key])), key
    for got, key in zip(O.linked_node_similarity_dense_small(x, ei), ("linked_sim", "linked_mean")):
        assert torch.equal(got, _t(z[key])), key
    for got, key in zip(O.neighborhood_similarity_dense_small(x, ei), ("nbr_weight", "nbr_mean")):
        assert torch.equal(got, _t(z[key])), key
    for got, key in zip(O.class_similarity_dense_small(x, y), ("class_mat", "class_mean")):
        assert torch.equal(got, _t(z[key])), key


@pytest.mark.parametrize("name", ["toolbox_dense_small.npz", "toolbox_dense_parted.npz"])
def test_oracle_toolbox_dense_large_fixtures(name):
    """The "large" variants sum in the reference's own order (row-by-row products, 1000-row
    blocks); the restatement agrees to 1e-6 on a cosine and 1e-6 relative on the big sum."""
    z = _z(name)
    x, ei, y = _t(z["x"]), _t(z["edge_index"]), _t(z["y"])
    want = float(z["parted_mean"])
    assert abs(float(O.node_similarity_dense_large_parted(x)[1]) - want) <= 1e-6 * abs(want)
    for got, key in zip(O.linked_node_similarity_dense_large(x, ei), ("linked_large_sim", "linked_large_mean")):
        np.testing.assert_allclose(got.numpy(), z[key], atol=1e-6)
    for got, key in zip(O.neighborhood_similarity_dense_large(x, ei), ("nbr_large_sim", "nbr_large_mean")):
        np.testing.assert_allclose(torch.as_tensor(got).numpy(), z[key], atol=1e-6)
    np.testing.assert_allclose(O.class_similarity_dense_large(x, y).numpy(), z["class_large_mat"], atol=1e-6)
    # the quirk of dense.py:28 is in the fixture: (sum - N) / (N - 1) * N, not a mean
    # (sum = off-diagonal sum + trace; the trace is N minus the all-zero rows)
    n = x.size(0)
    zero_rows = int((x.abs().sum(1) == 0).sum())
    assert abs(want - (float(z["node_mean"]) * n * (n - 1) - zero_rows) / (n - 1) * n) <= 2e-4 * abs(want)


def test_oracle_toolbox_sparse_fixture():
    """sparse.py's arithmetic is scikit-learn + scipy (real, float64): 1e-7."""
    import scipy.sparse as sp
    z = _z("toolbox_sparse_adj.npz")
    ei, y, n = _t(z["edge_index"]), _t(z["y"]), int(z["n"][0])
    adj = sp.csc_matrix((np.full(ei.size(1), 1), (ei[0].numpy(), ei[1].numpy())), shape=(n, n))
    np.testing.assert_allclose(O.cosine_similarity_sparse(adj), z["cosine"], atol=1e-7)
    for got, key in zip(O.node_similarity_sparse(adj), ("node_sim", "node_mean")):
        np.testing.assert_allclose(got.numpy(), z[key], atol=1e-7)
    for got, key in zip(O.linked_node_similarity_sparse(adj, ei), ("linked_sim", "linked_mean")):
        np.testing.assert_allclose(got.numpy(), z[key], atol=1e-7)
    for got, key in zip(O.neighborhood_similarity_sparse(adj, ei), ("nbr_sim", "nbr_mean")):
        np.testing.assert_allclose(torch.as_tensor(got).numpy(), z[key], atol=1e-7)
    np.testing.assert_allclose(O.class_similarity_sparse(adj, y).numpy(), z["class_mat"], atol=1e-7)


def test_ggcn_layer_oracle_against_the_reference_made_fixtures():
    """tests/golden/ggcn_sp_*.npz hold inputs and the outputs of the REFERENCE's GGCNlayer_SP
    (models.py:1453-1553, run verbatim by pin_reference.py: all core torch): the oracle's
    restatement gives the same forward and gradients to rounding (bit-equal forward where the
    fixture was made; torch's threaded CPU sparse products are host- and run-dependent in the last bits)."""
    import glob
    paths = sorted(glob.glob(os.path.join(GOLDEN, "ggcn_sp_*.npz")))
    assert len(paths) >= 3
    for path in paths:
        z = np.load(path)
        n, f, c = (int(v) for v in z["n"])
        kw = dict(zip(("use_degree", "use_sign", "use_decay"), (bool(v) for v in z["flags"])))
        adj = torch.sparse_coo_tensor(torch.from_numpy(z["adj_indices"]), torch.from_numpy(z["adj_values"]), (n, n)).coalesce()
        assert np.array_equal(O.ggcn_degree_precompute(adj)._values().numpy(), z["degree_values"])
        dp = torch.sparse_coo_tensor(adj._indices(), torch.from_numpy(z["degree_values"]), (n, n)).coalesce()
        layer = O.GGCNlayer_SP(f, c, "cpu", **kw)
        layer.load_state_dict({k[6:].replace("fcn_", "fcn."): torch.from_numpy(z[k]) for k in z.files
                               if k.startswith("param_")})
        h = torch.from_numpy(z["h"]).requires_grad_(True)
        out = layer(h, adj, dp)
        # (bit for bit on the machine that made the fixture - pin_reference.py asserts that; another
        # host CPU blocks torch's GEMM and sparse products differently)
        np.testing.assert_allclose(out.detach().numpy(), z["out"], rtol=0, atol=2e-6 * np.abs(z["out"]).max())
        (out * torch.from_numpy(z["gout"])).sum().backward()
        np.testing.assert_allclose(h.grad.numpy(), z["grad_h"], rtol=0, atol=2e-6 * np.abs(z["grad_h"]).max())
        for k, p in layer.named_parameters():
            want = z["grad_" + k.replace(".", "_")]
            np.testing.assert_allclose(p.grad.numpy(), want, rtol=0, atol=2e-6 * max(np.abs(want).max(), 1.0))
