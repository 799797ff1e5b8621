"""GPU parity of the fused aggregation forward, through the C ABI, against the oracle."""
import json
import os

import numpy as np
import pytest
import torch

from tests.helpers import (assert_close, check_selection, oracle_aggregate, random_graph)

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def run_gpu(cuda, h, ei, add_loops, remove_loops, top_k, thr, train=True, want_sel=True):
    from sngnn_amd.graph import Graph
    from sngnn_amd.ops import aggregate_forward
    g = Graph(ei.to(cuda), h.size(0), add_loops, remove_loops)
    out, wsel, inv, sel_src, sel_w = aggregate_forward(
        g, h.to(cuda), top_k, thr, save_for_backward=train,
        want_selection=want_sel and top_k is not None)
    torch.cuda.synchronize()
    return g, out, wsel, inv, sel_src, sel_w


def test_kat_appendix_b(cuda):
    kat = json.load(open(os.path.join(GOLDEN, "kat_appendix_b.json")))
    h = torch.tensor(kat["h"], dtype=torch.float32)
    ei = torch.tensor(kat["edge_index"], dtype=torch.int64)
    for case in kat["cases"]:
        k = case["top_k"]
        g, out, wsel, inv, sel_src, sel_w = run_gpu(cuda, h, ei, True, case["remove_loops"], k,
                                                    case["thr"])
        assert_close(out, torch.tensor(case["out"]), what=case["name"], rtol=1e-6, atol=1e-6)
        if k is not None:
            assert sel_src.cpu().tolist() == case["sel_src"], case["name"]


CASES = [
    # n, e, C, hubs, add, remove, top_k, thr
    (64, 300, 5, (), True, True, 1, 0.0),
    (64, 300, 7, (), True, False, None, 0.0),
    (200, 1500, 40, (), True, True, 16, 0.0),
    (200, 1500, 40, (), True, True, 3, 0.3),
    (200, 1500, 47, ((3, 90), (7, 150)), True, True, 16, 0.0),
    (300, 2500, 32, ((1, 140), (5, 260)), True, False, 10, 0.1),
    (500, 4000, 64, ((0, 499), (9, 300)), True, True, 2, -0.2),
    (500, 4000, 6, ((0, 499),), True, False, None, 0.0),
    (400, 3000, 128, ((2, 350),), True, True, 8, 0.0),
    (300, 2000, 200, ((2, 250),), True, True, 4, 0.05),
    (300, 2000, 129, (), True, False, 20, -1.5),
    (300, 2000, 300, ((4, 200),), True, True, 5, 0.0),
    (2000, 30000, 40, ((0, 1999), (1, 1500), (2, 700)), True, True, 16, 0.0),
    (2000, 30000, 40, ((0, 1999), (1, 1500), (2, 700)), True, True, 200, 0.0),
    (2000, 30000, 8, ((0, 1999),), True, True, 0, 0.0),
]


@pytest.mark.parametrize("n,e,C,hubs,add,rem,k,thr", CASES)
def test_forward_matches_oracle(cuda, n, e, C, hubs, add, rem, k, thr):
    ei = random_graph(n, e, seed=n + e + C, hubs=hubs)
    gen = torch.Generator().manual_seed(C * 7 + n)
    h = torch.randn(n, C, generator=gen)
    h[5] = h[6]                      # exact duplicate rows -> exact ties
    h[7] = 2.0 * h[6]
    h[11] = 0.0                      # zero row: normalises to 0 (eps clamp)
    ref = oracle_aggregate(h, ei, add, rem, k, thr)
    g, out, wsel, inv, sel_src, sel_w = run_gpu(cuda, h, ei, add, rem, k, thr)
    assert g.num_edges == ref["ei"].size(1)
    assert_close(out, ref["out"])
    if k is not None and k > 0:
        near = check_selection(ref, sel_src, sel_w, k, thr, strict=False, h=h)
        assert near <= max(1, n // 100), f"{near} rows needed the near-tie rule"
    # inverse norms
    want_inv = 1.0 / h.norm(dim=1).clamp_min(1e-12)
    assert_close(inv, want_inv, what="inv_norm", rtol=1e-6, atol=0)
    # the same launch without the optional outputs gives the same bits
    _, out2, *_ = run_gpu(cuda, h, ei, add, rem, k, thr, train=False, want_sel=False)
    assert torch.equal(out2.cpu(), out.cpu())


def test_saved_weights_match_oracle(cuda):
    n, C, k, thr = 400, 40, 6, 0.1
    ei = random_graph(n, 5000, seed=3, hubs=((0, 399), (1, 200)))
    h = torch.randn(n, C, generator=torch.Generator().manual_seed(1))
    ref = oracle_aggregate(h, ei, True, True, k, thr)
    g, out, wsel, inv, sel_src, sel_w = run_gpu(cuda, h, ei, True, True, k, thr)
    # map CSR edges back to positions of the E' edge list
    eid = torch.from_numpy(g.array("eid").astype(np.int64))
    w_list = torch.full((g.num_edges,), float("nan"))
    w_list[eid] = wsel.cpu()
    sel = w_list > -3.0
    ref_sel = torch.zeros_like(sel)
    ref_sel[ref["sel_pos"][ref["sel_pos"] >= 0]] = True
    assert torch.equal(sel, ref_sel)
    assert_close(w_list[sel], ref["s"][sel], what="wsel", rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("k,thr,C", [(16, 0.0, 40), (5, 0.1, 7), (32, -0.5, 64)])
def test_many_moderate_split_rows(cuda, k, thr, C):
    """Products-like degree profile: thousands of rows just above the wave-row limit plus a
    few very large hubs, so the split-row finalize runs BOTH its forms (one wave per row for
    rows whose candidates fit one selection, the workgroup tournament for the hubs);
    compared with the C oracle (selection exact up to the near-tie rule of tests/helpers.py:
    with ~10^6 random cosines a pair one ulp apart does occur)."""
    from oracle import c_oracle as CO
    n = 6000
    rng = np.random.default_rng(11)
    hubs = [(int(v), int(d)) for v, d in zip(range(0, 2300), rng.integers(129, 700, size=2300))]
    hubs += [(2300, 5000), (2301, 3000), (2302, 1100)]
    ei = random_graph(n, 20000, seed=5, hubs=hubs)
    h = torch.randn(n, C, generator=torch.Generator().manual_seed(3))
    ref = CO.aggregate(h.numpy(), ei.numpy(), add_loops=True, remove_loops=True, top_k=k, thr=thr)
    g, out, wsel, inv, sel_src, sel_w = run_gpu(cuda, h, ei, True, True, k, thr)
    assert g.num_edges == ref["ei"].shape[1]
    assert_close(out, torch.from_numpy(ref["out"]))
    res = dict(sel_src=torch.from_numpy(ref["sel_src"]), ei=torch.from_numpy(ref["ei"]),
               s=torch.from_numpy(ref["s"]))
    assert check_selection(res, sel_src, sel_w, k, thr, strict=False, h=h) <= 3
    # training-mode bookkeeping of the same rows: kept cosines land on the kept edges
    eid = torch.from_numpy(g.array("eid").astype(np.int64))
    w_list = torch.empty(g.num_edges)
    w_list[eid] = wsel.cpu()
    kept = torch.from_numpy(ref["weight"] != 0) if "weight" in ref else None
    if kept is not None:
        assert torch.equal(w_list > -3.0, kept | ((w_list > -3.0) & (torch.from_numpy(ref["s"]) == 0)))
