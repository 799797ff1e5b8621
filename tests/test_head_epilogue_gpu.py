"""The classification head inside the LAST layer's own launches (sngnn_epilogue_t.head_*, ops.HeadEpilogue):
log_softmax + nll_loss on a mask + accuracy count (models.py:86,211,303; train.py:81-84, 98-102, 112-116) -
split rows in their finalize, the others read back by extra workgroups of that launch - against the
stand-alone head on the stored logits (the same per-row instruction sequence: equal gradient BITS, metrics
equal up to the order of the row sum) and against torch's own log_softmax / nll_loss."""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import random_graph

pytestmark = pytest.mark.gpu

# n, e, C, top_k, thr, hubs: C covers the 8- and 16-lane rows; hubs: wave rows, split rows of both
# finalize kinds (<= 8 tasks: one wave; more: the workgroup tournament), top_k None: no selection
CASES = [
    (300, 2000, 8, 4, 0.0, ()),
    (300, 2000, 40, 16, 0.0, ((3, 90), (7, 200))),
    (3000, 30000, 40, 16, 0.0, ((0, 2900), (1, 1500), (2, 600), (5, 129), (6, 128), (9, 17))),
    (3000, 30000, 32, 2, 0.3, ((0, 2900), (1, 700))),
    (3000, 30000, 64, None, 0.0, ((0, 2900), (1, 700), (4, 2200))),
    (2000, 12000, 12, 1, 0.0, ((0, 1999),)),
    (500, 3000, 4, 3, -1.0, ()),
    (40000, 300000, 40, 16, 0.0, ((0, 30000), (1, 9000), (2, 1100))),
    # products-like: thousands of moderate split rows - the finalize is two launches (tournament rows,
    # wave rows) and the head role a launch of its own in front of them
    (6000, 1300000, 40, 16, 0.0, ((0, 5000), (1, 2500))),
    (6000, 1300000, 8, None, 0.0, ((0, 5000),)),
]


def _setup(cuda, n, e, c, hubs, seed=0):
    from sngnn_amd.graph import Graph
    gen = torch.Generator().manual_seed(seed + n + c)
    ei = random_graph(n, e, seed=seed + n + e, hubs=hubs)
    h = torch.randn(n, c, generator=gen).to(cuda)
    y = torch.randint(0, c, (n,), generator=gen).to(cuda)
    r = torch.rand(n, generator=gen)
    m_train = (r < 0.6).to(torch.uint8).to(cuda)
    sets = (((r >= 0.6) & (r < 0.8)).to(torch.uint8) + 2 * (r >= 0.8).to(torch.uint8)).to(cuda)
    g = Graph(ei.to(cuda), n, True, True)
    return g, h, y, m_train, sets


@pytest.mark.parametrize("n,e,c,top_k,thr,hubs", CASES)
@pytest.mark.parametrize("with_bias", [False, True])
def test_head_epilogue_equals_the_head_on_the_stored_logits(cuda, n, e, c, top_k, thr, hubs, with_bias):
    from sngnn_amd import ops
    g, h, y, m_train, sets = _setup(cuda, n, e, c, hubs)
    assert ops.head_supported(g, c, top_k)
    bias = (torch.randn(c, generator=torch.Generator().manual_seed(5)) * 0.3).to(cuda) if with_bias else None
    logits = ops.aggregate_forward(g, h, top_k, thr)[0]
    if bias is not None:
        logits = logits + bias
    n_tr = int(m_train.sum())
    n_a, n_b = int((sets & 1).ne(0).sum()), int((sets & 2).ne(0).sum())

    # training form: metrics of one split + the gradient as the layer's output
    (loss_ref, corr_ref), grad_ref = ops.head_nll_with_grad(logits, y, m_train, n_tr)
    met = torch.zeros(2, device=cuda)
    head = ops.HeadEpilogue(y, m_train, met, n_tr, grad=True)
    out = ops.aggregate(h, g, top_k, thr, None, None, bias, head)
    torch.cuda.synchronize()
    assert torch.equal(out, grad_ref), "the gradient rows must be the stand-alone head's, bit for bit"
    assert float(met[1]) == float(corr_ref)
    assert abs(float(met[0]) - float(loss_ref)) <= 2e-6 * max(1.0, abs(float(loss_ref)))
    # against torch itself
    lp = F.log_softmax(logits.double(), dim=1)
    mb = m_train.bool()
    assert abs(float(met[0]) - float(F.nll_loss(lp[mb], y[mb]))) <= 1e-5
    assert float(met[1]) == float((lp[mb].argmax(1) == y[mb]).sum())

    # evaluation form: two splits off one forward; the output holds the logits
    met4_ref = ops.head_nll2(logits, y, sets, n_a, n_b)
    met4 = torch.zeros(4, device=cuda)
    stored = ops.aggregate(h, g, top_k, thr, None, None, bias, ops.HeadEpilogue(y, sets, met4, n_a, n_b))
    assert torch.equal(stored, logits)
    assert float(met4[1]) == float(met4_ref[1]) and float(met4[3]) == float(met4_ref[3])
    for q in (0, 2):
        assert abs(float(met4[q]) - float(met4_ref[q])) <= 2e-6 * max(1.0, abs(float(met4_ref[q])))
    # one split, evaluation (the reference's separate validation / test passes)
    met2 = torch.zeros(2, device=cuda)
    ops.aggregate(h, g, top_k, thr, None, None, bias, ops.HeadEpilogue(y, (sets & 1).contiguous(), met2, n_a))
    assert float(met2[1]) == float(met4[1]) and abs(float(met2[0]) - float(met4[0])) <= 2e-6 * max(1.0, abs(float(met4[0])))
    # deterministic
    met4c = torch.zeros(4, device=cuda)
    ops.aggregate(h, g, top_k, thr, None, None, bias, ops.HeadEpilogue(y, sets, met4c, n_a, n_b))
    assert torch.equal(met4c, met4)


def test_head_epilogue_backward_is_the_backward_of_the_loss(cuda):
    """h.grad (and the conv bias's) through G.backward(G) == through loss.backward() on the op sequence."""
    from sngnn_amd import ops
    n, e, c, top_k, thr = 3000, 30000, 40, 16, 0.0
    g, h, y, m_train, _ = _setup(cuda, n, e, c, ((0, 2900), (1, 700), (5, 129)))
    n_tr = int(m_train.sum())
    bias = (torch.randn(c, generator=torch.Generator().manual_seed(9)) * 0.1).to(cuda)
    h1, b1 = h.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    logits = ops.aggregate(h1, g, top_k, thr) + b1
    loss, _ = ops.head_nll(logits, y, m_train, n_tr)
    loss.backward()
    h2, b2 = h.clone().requires_grad_(True), bias.clone().requires_grad_(True)
    met = torch.zeros(2, device=cuda)
    out = ops.aggregate(h2, g, top_k, thr, None, None, b2, ops.HeadEpilogue(y, m_train, met, n_tr, grad=True))
    out.backward(out.detach())
    assert torch.equal(h2.grad, h1.grad)
    assert torch.allclose(b2.grad, b1.grad, rtol=1e-5, atol=1e-7)
    assert abs(float(met[0]) - float(loss.detach())) <= 2e-6 * max(1.0, abs(float(loss.detach())))


def test_head_epilogue_argument_checks(cuda):
    from sngnn_amd import ops
    g, h, y, m_train, sets = _setup(cuda, 200, 1000, 6, ())           # C % 4 != 0
    assert not ops.head_supported(g, 6, 2)
    g, h, y, m_train, sets = _setup(cuda, 200, 1000, 72, ())          # C > 64
    assert not ops.head_supported(g, 72, 2)
    with pytest.raises(ValueError):
        ops.HeadEpilogue(y, sets, torch.zeros(4, device=cuda), 1, 1, grad=True)
    with pytest.raises(ValueError):
        ops.HeadEpilogue(y, sets, torch.zeros(2, device=cuda), 1, 1)
    with pytest.raises(Exception):
        ops.aggregate(h, g, 2, 0.0, None, None, None, ops.HeadEpilogue(y, m_train, torch.zeros(2, device=cuda), 5))


@pytest.mark.parametrize("kind,args", [
    ("SNGNN_Plus", (24, 16, 8, None, 1, 4, 0.0, 1, 0.5)),
    ("SNGNN_Plus", (24, 16, 8, None, 2, 2, 0.0, 1, 0.0)),
    ("SNGNN", (24, 16, 8, 1)),
    ("SNGNN_Plus_Plus", (24, 16, 8, None, 1, 4, 0.0, 0.3, 1, 0.5)),     # blend last: the head in the blend's pass
    ("SNGNN_Plus_Plus", (24, 16, 8, None, 2, 4, 0.0, 0.3, 1, 0.0)),
])
def test_graphed_epoch_with_the_head_in_the_launches_equals_without(cuda, kind, args, monkeypatch):
    import sngnn_amd
    from sngnn_amd import synth
    from sngnn_amd.train import GraphedEpoch
    n = 1500
    ei = random_graph(n, 12000, seed=4, hubs=((0, 1400), (1, 300)))
    gen = torch.Generator().manual_seed(2)
    r = torch.rand(n, generator=gen)
    data = synth.Data(x=torch.randn(n, 24, generator=gen), edge_index=ei, y=torch.randint(0, 8, (n,), generator=gen),
                      train_mask=r < 0.6, val_mask=(r >= 0.6) & (r < 0.8), test_mask=r >= 0.8).to(cuda)
    args = tuple(n if a is None else a for a in args)
    runs = {}
    for fuse in ("1", "0"):
        monkeypatch.setenv("SNGNN_FUSE_HEAD", fuse)
        torch.manual_seed(0)
        model = getattr(sngnn_amd, kind)(*args).to(cuda)
        if hasattr(model, "dropout"):
            model.dropout.p = 0.0
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
        ge = GraphedEpoch(model, data, opt, warmup=1)
        runs[fuse] = [ge.run() for _ in range(4)]
    for a, b in zip(runs["1"], runs["0"]):
        for key in a:
            assert abs(a[key] - b[key]) <= 2e-5 * max(1.0, abs(b[key])), (key, a[key], b[key])


@pytest.mark.parametrize("n,c", [(300, 8), (5000, 40), (5000, 64), (777, 4)])
def test_blend_with_the_head_behind_it_equals_blend_then_head(cuda, n, c):
    """ops.blend_head (sngnn_head_nll_blend): SNGNN++'s last blend (models.py:134) and the classification head in
    one pass, against ops.blend followed by the stand-alone head: the same gradient BITS (the blend is formed with
    the blend kernel's rounding, the per-row arithmetic is head_row.h's), metrics up to the order of the row sum;
    its backward is the blend's backward on that gradient; the two-split evaluation form likewise."""
    from sngnn_amd import ops
    gen = torch.Generator().manual_seed(n + c)
    o0 = torch.randn(n, c, generator=gen).to(cuda).requires_grad_(True)
    o1 = torch.randn(n, c, generator=gen).to(cuda).requires_grad_(True)
    beta = torch.tensor([0.3], device=cuda, requires_grad=True)
    y = torch.randint(0, c, (n,), generator=gen).to(cuda)
    r = torch.rand(n, generator=gen)
    m_train = (r < 0.6).to(torch.uint8).to(cuda)
    sets = (((r >= 0.6) & (r < 0.8)).to(torch.uint8) + 2 * (r >= 0.8).to(torch.uint8)).to(cuda)
    n_tr, n_a, n_b = int(m_train.sum()), int((sets & 1).ne(0).sum()), int((sets & 2).ne(0).sum())
    # reference: blend, then the head on the stored logits
    z = ops.blend(o0, o1, beta)
    (loss_ref, corr_ref), grad_ref = ops.head_nll_with_grad(z, y, m_train, n_tr)
    z.backward(grad_ref)
    want = [t.grad.clone() for t in (o0, o1, beta)]
    for t in (o0, o1, beta):
        t.grad = None
    # fused, training form
    met = torch.zeros(2, device=cuda)
    head = ops.HeadEpilogue(y, m_train, met, n_tr, grad=True)
    g = ops.blend_head(o0, o1, beta, head)
    assert head.applied and torch.equal(g.detach(), grad_ref)
    assert abs(float(met[0]) - float(loss_ref)) <= 2e-6 * max(1.0, abs(float(loss_ref))) and float(met[1]) == float(corr_ref)
    g.backward(g.detach())
    for got, w in zip((o0.grad, o1.grad, beta.grad), want):
        assert torch.equal(got, w)
    # fused, evaluation form: both splits' metrics, the logits as the result
    met4 = torch.zeros(4, device=cuda)
    head2 = ops.HeadEpilogue(y, sets, met4, n_a, n_b)
    with torch.no_grad():
        logits = ops.blend_head(o0, o1, beta, head2)
        ref4 = ops.head_nll2(z.detach(), y, sets, n_a, n_b)
    assert head2.applied and torch.equal(logits, z.detach())
    assert float((met4 - ref4).abs().max()) <= 2e-6 * max(1.0, float(ref4.abs().max()))
    # shapes the fused form does not take: the caller is told
    head3 = ops.HeadEpilogue(y, m_train, met, n_tr, grad=True)
    assert ops.blend_head(o0[:, :c - 1], o1[:, :c - 1], beta, head3) is None and not head3.applied
