"""GPU parity of the cosine-attention mode (AGNNConv / AGNN, models.py:336-405) with
the oracle's restatement, through the C ABI (sngnn_attn_forward / _backward)."""
import numpy as np
import pytest
import torch

from oracle import sngnn_oracle as O
from tests.helpers import assert_close, random_graph
from tests.test_agg_backward_gpu import assert_grad_close

pytestmark = pytest.mark.gpu

# n, e, C, hubs  -  C covers the 4/2/1-float lane layouts and G = 8..64, R up to 4
CASES = [
    (64, 300, 5, ()),
    (64, 300, 6, ()),
    (200, 1500, 40, ()),
    (200, 1500, 47, ((3, 90), (7, 150))),
    (500, 4000, 64, ((0, 499), (9, 300))),
    (500, 4000, 2, ((0, 499),)),
    (400, 3000, 128, ((2, 350),)),
    (300, 2000, 129, ()),
    (300, 2000, 300, ((4, 200),)),
    (1500, 20000, 40, ((0, 1499), (1, 900), (2, 129), (3, 128), (4, 17), (5, 16))),
    (300, 2000, 512, ((4, 200),)),
    # a 12 000-edge row: more than one trip of the split-row finalizes' batched rewrites (8 x 1024
    # alpha values / 4 x 1024 records a trip), and > 1024 tasks' partials summed across the workgroup
    (20000, 60000, 8, ((0, 12000), (1, 5000), (2, 4097))),
    (140000, 200000, 3, ((7, 135000),)),
]


def _inputs(n, e, C, hubs):
    ei = random_graph(n, e, seed=n + e + C, hubs=hubs)
    # original self-loops (must be replaced, not duplicated) and a hub SOURCE
    loops = torch.arange(0, n, 3)
    ei = torch.cat([ei, torch.stack([loops, loops]),
                    torch.stack([torch.full((n // 2,), 3), torch.arange(n // 2) * 2 + 1])], 1)
    ei = torch.unique(ei, dim=1)
    gen = torch.Generator().manual_seed(C + n)
    h = torch.randn(n, C, generator=gen)
    h[5] = h[6]
    gout = torch.randn(n, C, generator=gen)
    return ei, h, gout


@pytest.mark.parametrize("n,e,C,hubs", CASES)
def test_forward_backward_match_oracle(cuda, n, e, C, hubs):
    from sngnn_amd.graph import Graph, LOOPS_REPLACE
    from sngnn_amd import ops
    ei, h, gout = _inputs(n, e, C, hubs)
    h_ref = h.clone().requires_grad_(True)
    ref = O.attention_reference(h_ref, ei)
    (ref["out"] * gout).sum().backward()

    g = Graph(ei.to(cuda), n, True, LOOPS_REPLACE)
    assert g.num_edges == ref["ei"].size(1)
    out, alpha = ops.attention_forward(g, h.to(cuda))
    torch.cuda.synchronize()
    assert_close(out, ref["out"])
    # alpha comes back in CSR order; eid maps it to the reference's edge list
    eid = torch.from_numpy(g.array("eid").astype(np.int64))
    a_list = torch.empty(g.num_edges)
    a_list[eid] = alpha.cpu()
    assert_close(a_list, ref["alpha"], what="alpha", rtol=1e-5, atol=1e-7)
    # without alpha (inference) the output is the same bits
    out2, none = ops.attention_forward(g, h.to(cuda), save_for_backward=False)
    assert none is None and torch.equal(out2, out)

    h_gpu = h.to(cuda).requires_grad_(True)
    o = ops.attention(h_gpu, g)
    (o * gout.to(cuda)).sum().backward()
    torch.cuda.synchronize()
    assert torch.equal(o, out)
    assert_grad_close(h_gpu.grad, h_ref.grad, "grad_h")
    h2 = h.to(cuda).requires_grad_(True)
    (ops.attention(h2, g) * gout.to(cuda)).sum().backward()
    assert torch.equal(h2.grad, h_gpu.grad), "backward must be deterministic"


def test_edge_list_replaces_original_loops(cuda):
    from sngnn_amd.graph import Graph, LOOPS_REPLACE
    n = 50
    ei, _, _ = _inputs(n, 300, 4, ())
    want = O.agnn_edge_list(ei, n)
    g = Graph(ei.to(cuda), n, True, LOOPS_REPLACE)
    rowptr, col, eid = g.array("rowptr"), g.array("col"), g.array("eid")
    dst = np.repeat(np.arange(n), np.diff(rowptr))
    got = np.empty((2, g.num_edges), np.int64)
    got[0, eid] = col
    got[1, eid] = dst
    assert np.array_equal(got, want.numpy())
    # exactly one loop per node, and it is the LAST in-edge of its row (appended)
    assert ((got[0] == got[1]).sum()) == n
    assert np.array_equal(col[rowptr[1:] - 1], np.arange(n))


def test_rows_sum_to_one_and_constant_features(cuda):
    """Linearity property at a size the oracle is not needed for: alpha sums to 1 per
    target, so constant features are reproduced exactly up to rounding."""
    from sngnn_amd.graph import Graph, LOOPS_REPLACE
    from sngnn_amd import ops
    n, C = 20000, 40
    ei = random_graph(n, 200000, seed=3, hubs=((0, 15000), (1, 2000)))
    g = Graph(ei.to(cuda), n, True, LOOPS_REPLACE)
    h = torch.randn(n, C, generator=torch.Generator().manual_seed(0)).to(cuda)
    out, alpha = ops.attention_forward(g, h)
    rowptr = torch.from_numpy(g.array("rowptr").astype(np.int64)).to(cuda)
    sums = torch.zeros(n, device=cuda).index_add_(
        0, torch.repeat_interleave(torch.arange(n, device=cuda), rowptr.diff()), alpha)
    assert (sums - 1).abs().max().item() < 2e-5
    assert alpha.min().item() > 0
    # every output row is a convex combination of source rows
    assert (out.abs().max(dim=1).values <= h.abs().max() + 1e-5).all()
    const = torch.ones(n, C, device=cuda) * 0.75
    oc, _ = ops.attention_forward(g, const)
    assert (oc - 0.75).abs().max().item() < 2e-5


@pytest.mark.parametrize("bn", [False, True])
def test_agnn_model_matches_oracle(cuda, bn):
    import sngnn_amd
    from sngnn_amd.synth import Data
    n, f, hid, classes = 300, 24, 40, 5
    ei, _, _ = _inputs(n, 2500, 8, ((2, 200),))
    x = torch.randn(n, f, generator=torch.Generator().manual_seed(4))
    y = torch.randint(0, classes, (n,), generator=torch.Generator().manual_seed(5))
    torch.manual_seed(11)
    ref = O.AGNN(f, hid, classes, 3, bn)
    torch.manual_seed(11)
    mod = sngnn_amd.AGNN(f, hid, classes, 3, bn)
    assert list(ref.state_dict()) == list(mod.state_dict())
    for a, b in zip(ref.state_dict().values(), mod.state_dict().values()):
        assert torch.equal(a, b)
    mod = mod.to(cuda)
    ref.eval(), mod.eval()
    want = ref(Data(x=x, edge_index=ei))
    got = mod(Data(x=x.to(cuda), edge_index=ei.to(cuda)))
    assert_close(got, want, what="log-probs", rtol=2e-5, atol=2e-5)
    # one training-mode backward (dropout off so both sides see the same mask)
    ref.train(), mod.train()
    ref.dropout.p = mod.dropout.p = 0.0
    torch.nn.functional.nll_loss(ref(Data(x=x, edge_index=ei)), y).backward()
    torch.nn.functional.nll_loss(mod(Data(x=x.to(cuda), edge_index=ei.to(cuda))), y.to(cuda)).backward()
    for (name, p), q in zip(ref.named_parameters(), mod.parameters()):
        assert_grad_close(q.grad, p.grad, name, rel=1e-4)


def test_partition_equals_whole_graph(cuda):
    """Two node-range partitions reproduce the single-graph result (the multi-GPU path
    without the collectives: h is already 'all-gathered', partial grads are summed)."""
    from sngnn_amd.graph import Graph, LOOPS_REPLACE
    from sngnn_amd import ops
    n, C = 400, 40
    ei, h, gout = _inputs(n, 3000, C, ((7, 300),))
    eid, hd, gd = ei.to(cuda), h.to(cuda), gout.to(cuda)
    whole = Graph(eid, n, True, LOOPS_REPLACE)
    hw = hd.clone().requires_grad_(True)
    ow = ops.attention(hw, whole)
    (ow * gd).sum().backward()
    cut = 170
    outs, grad = [], torch.zeros_like(hd)
    for r0, r1 in ((0, cut), (cut, n)):
        part = Graph(eid, n, True, LOOPS_REPLACE, row_range=(r0, r1))
        hp = hd.clone().requires_grad_(True)
        op = ops.attention(hp, part)
        (op * gd[r0:r1]).sum().backward()
        outs.append(op.detach())
        grad += hp.grad
    assert torch.equal(torch.cat(outs), ow.detach())
    assert_grad_close(grad, hw.grad, "summed partial grads", rel=2e-6)


def test_bad_arguments(cuda):
    from sngnn_amd.graph import Graph, LOOPS_REPLACE
    from sngnn_amd import ops
    ei = random_graph(30, 100, seed=1).to(cuda)
    g = Graph(ei, 30, True, LOOPS_REPLACE)
    with pytest.raises(ValueError):
        ops.attention_forward(g, torch.zeros(29, 8, device=cuda))
    with pytest.raises(ValueError):
        ops.attention_forward(g, torch.zeros(30, 513, device=cuda))
    with pytest.raises(ValueError):
        Graph(ei, 30, True, 3)


def test_agnn_graphed_epoch_matches_eager(cuda):
    """The attention model runs under the HIP-graph epoch like the SNGNN family."""
    import sngnn_amd
    from sngnn_amd import synth
    from sngnn_amd import train as T
    data = synth.make_dataset("cora", scale=0.5).to(cuda)
    n, f = data.x.shape
    runs = []
    for graphed in (False, True):
        torch.manual_seed(11)
        model = sngnn_amd.AGNN(f, 16, 7, 1).to(cuda)
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4, capturable=True)
        fn = T.train_graphed if graphed else T.train
        runs.append(fn(model, data, opt, epochs=5, patience=100))
    for a, b in zip(runs[0]["history"], runs[1]["history"]):
        for k in ("train_loss", "val_loss", "test_loss", "train_acc", "val_acc", "test_acc"):
            assert abs(a[k] - b[k]) <= 1e-4 * max(1.0, abs(a[k])), (a["epoch"], k, a[k], b[k])


def test_attention_at_full_arxiv_size_against_oracle_autograd(cuda):
    """The cosine-attention mode at BASELINE config 4's size (169 343 nodes, 1.16 M edges, the 13 k-edge
    hub: split rows of more than a hundred tasks, the task-wide finalizes), C = 40: forward, alpha and
    the gradient against the oracle's restatement of AGNNConv (models.py:377-405) and its autograd."""
    from sngnn_amd import ops, synth
    from sngnn_amd.graph import Graph, LOOPS_REPLACE
    d = synth.make_dataset("arxiv", with_features=False)
    n = d.x.size(0)
    gen = torch.Generator().manual_seed(11)
    h = torch.randn(n, 40, generator=gen)
    gout = torch.randn(n, 40, generator=gen)
    h_ref = h.clone().requires_grad_(True)
    ref = O.attention_reference(h_ref, d.edge_index)
    (ref["out"] * gout).sum().backward()
    g = Graph(d.edge_index.to(cuda), n, True, LOOPS_REPLACE)
    assert g.num_edges == ref["ei"].size(1) and g.max_in_degree > 10000
    out, alpha = ops.attention_forward(g, h.to(cuda))
    assert_close(out, ref["out"])
    eid = torch.from_numpy(g.array("eid").astype(np.int64))
    a_list = torch.empty(g.num_edges)
    a_list[eid] = alpha.cpu()
    assert_close(a_list, ref["alpha"], what="alpha", rtol=1e-5, atol=1e-7)
    # every row's coefficients sum to one (the hub's 13 k of them too)
    sums = torch.zeros(n).index_add_(0, ref["ei"][1], a_list)
    assert float((sums - 1).abs().max()) <= 2e-5
    hg = h.to(cuda).requires_grad_(True)
    (ops.attention(hg, g) * gout.to(cuda)).sum().backward()
    assert_grad_close(hg.grad, h_ref.grad, "grad_h at arxiv size")
