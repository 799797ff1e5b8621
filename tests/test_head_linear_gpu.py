"""GPU: the fused classification head and the Linear weight-gradient kernel against
plain PyTorch fp32 (floating-point kernels: a torch reference is the checker here)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,c", [(1000, 5), (5000, 40), (777, 47), (300, 130)])
def test_head_matches_log_softmax_nll_and_accuracy(cuda, n, c):
    from sngnn_amd import ops
    g = torch.Generator().manual_seed(n)
    z = (torch.randn(n, c, generator=g) * 3).to(cuda).requires_grad_(True)
    y = torch.randint(0, c, (n,), generator=g).to(cuda)
    mask = (torch.rand(n, generator=g) < 0.6).to(cuda)
    z_ref = z.detach().clone().requires_grad_(True)
    logp = F.log_softmax(z_ref, dim=1)
    loss_ref = F.nll_loss(logp[mask], y[mask])
    loss_ref.backward()
    corr_ref = int((logp[mask].max(dim=1)[1] == y[mask]).sum())
    loss, corr = ops.head_nll(z, y, mask.to(torch.uint8), int(mask.sum()))
    (loss * 1.0).backward()
    assert abs(float(loss) - float(loss_ref)) <= 2e-6 * max(1.0, abs(float(loss_ref)))
    assert int(corr) == corr_ref
    assert (z.grad - z_ref.grad).abs().max() <= 1e-6 * max(1e-3, float(z_ref.grad.abs().max())) + 1e-9
    assert bool((z.grad[~mask] == 0).all())
    # no-grad evaluation path gives the same numbers
    with torch.no_grad():
        l2, c2 = ops.head_nll(z.detach(), y, mask.to(torch.uint8), int(mask.sum()))
    assert float(l2) == float(loss) and int(c2) == corr_ref


@pytest.mark.parametrize("n,c", [(5000, 40), (777, 7), (70001, 47), (300, 64), (1, 3)])
def test_head_two_splits_equal_two_calls(cuda, n, c):
    """sngnn_head_nll2 (validation + test metrics off one forward, train.py:92-117) gives exactly
    what two sngnn_head_nll calls give - overlapping, disjoint and empty splits."""
    from sngnn_amd import ops
    g = torch.Generator().manual_seed(n + c)
    z = (torch.randn(n, c, generator=g) * 3).to(cuda)
    y = torch.randint(0, c, (n,), generator=g).to(cuda)
    r = torch.rand(n, generator=g)
    for ma, mb in (((r < 0.3), (r > 0.6)), ((r < 0.7), (r > 0.4)), ((r < 2), (r < 0))):
        ma, mb = ma.to(cuda).to(torch.uint8), mb.to(cuda).to(torch.uint8)
        la, ca = ops.head_nll(z, y, ma, int(ma.sum()))
        lb, cb = ops.head_nll(z, y, mb, int(mb.sum()))
        out = ops.head_nll2(z, y, ma | (mb << 1), int(ma.sum()), int(mb.sum()))
        assert out.tolist() == [float(la), float(ca), float(lb), float(cb)]
    with pytest.raises(ValueError):
        ops.head_nll2(torch.zeros(4, 65, device=cuda), y[:4], ma[:4], 1, 1)


@pytest.mark.parametrize("n,f,c", [(3000, 128, 40), (2277, 2325, 5), (1000, 33, 47), (513, 200, 70),
                                   (9000, 128, 40), (5001, 77, 64), (4500, 1433, 7), (4100, 100, 33),
                                   # row-tile MFMA path: F in {16, 32, 64, 128}, every tile count
                                   (5000, 16, 5), (4097, 32, 40), (6000, 64, 64), (4500, 128, 17),
                                   (70001, 128, 40), (4111, 32, 33), (8200, 64, 1)])
def test_linear_wgrad_matches_autograd(cuda, n, f, c):
    from sngnn_amd import ops
    g = torch.Generator().manual_seed(f)
    x = torch.randn(n, f, generator=g).to(cuda)
    lin = torch.nn.Linear(f, c).to(cuda)
    gout = torch.randn(n, c, generator=g).to(cuda)
    (lin(x) * gout).sum().backward()
    gw_ref, gb_ref = lin.weight.grad.clone(), lin.bias.grad.clone()
    lin.zero_grad()
    out = ops.linear(x, lin)
    ref_out = lin(x)
    assert (out - ref_out).abs().max() <= 2e-6 * max(1.0, float(ref_out.abs().max()))
    (out * gout).sum().backward()
    tol_w = 2e-5 * float(gw_ref.abs().max())
    assert (lin.weight.grad - gw_ref).abs().max() <= tol_w
    assert (lin.bias.grad - gb_ref).abs().max() <= 2e-5 * float(gb_ref.abs().max())
    # x that needs a gradient (hidden layers)
    xr = x.clone().requires_grad_(True)
    lin.zero_grad()
    (ops.linear(xr, lin) * gout).sum().backward()
    xr2 = x.clone().requires_grad_(True)
    (lin(xr2) * gout).sum().backward()
    assert (xr.grad - xr2.grad).abs().max() <= 1e-4 * float(xr2.grad.abs().max())


@pytest.mark.parametrize("n,c", [(1000, 40), (777, 5), (4096, 47), (1, 1)])
def test_blend_matches_the_reference_expression(cuda, n, c):
    """models.py:134 and its autograd: forward and both tensor gradients bit-exact, the
    scalar gradient a differently ordered sum."""
    from sngnn_amd import ops
    g = torch.Generator().manual_seed(n + c)
    o0 = torch.randn(n, c, generator=g).to(cuda).requires_grad_(True)
    o1 = torch.randn(n, c, generator=g).to(cuda).requires_grad_(True)
    gout = torch.randn(n, c, generator=g).to(cuda)
    beta = torch.tensor([0.3], device=cuda, requires_grad=True)
    ref = beta * o0 + (1 - beta) * o1
    ref.backward(gout)
    want = (ref.detach().clone(), o0.grad.clone(), o1.grad.clone(), beta.grad.clone())
    o0.grad = o1.grad = beta.grad = None
    out = ops.blend(o0, o1, beta)
    out.backward(gout)
    assert torch.equal(out, want[0])
    assert torch.equal(o0.grad, want[1]) and torch.equal(o1.grad, want[2])
    assert abs(float(beta.grad) - float(want[3])) <= 2e-5 * max(1.0, abs(float(want[3])))
    # anything the fused path does not take falls back to the expression
    assert torch.equal(ops.blend(o0.detach()[:, : max(c - 1, 1)], o1.detach()[:, : max(c - 1, 1)], beta.detach()),
                       (beta * o0[:, : max(c - 1, 1)] + (1 - beta) * o1[:, : max(c - 1, 1)]).detach())


@pytest.mark.parametrize("n,f,c", [(5000, 128, 40), (4097, 128, 32), (6000, 32, 40), (4500, 64, 64), (8191, 16, 8),
                                   (5003, 128, 48), (4200, 64, 36), (4100, 128, 4)])
def test_lin_normalising_epilogue_is_bitwise_the_normalisation_pass(cuda, n, f, c):
    """sngnn_linear_forward_normalized (models.py:237-238 in one launch): h equals
    sngnn_linear_forward's bit for bit, and the unit rows / norms / filter rows equal what
    sngnn_normalize_rows_filter computes from that h - bit for bit, incl. duplicate rows, a zero
    row and a row with one non-zero channel."""
    from sngnn_amd import _lib, ops
    lib = _lib.load()
    assert lib.sngnn_linear_normalized_supported(n, f, c)
    g = torch.Generator().manual_seed(n + c)
    x = torch.randn(n, f, generator=g)
    x[5] = x[6]
    x[9] = 0.0
    w = torch.randn(c, f, generator=g) / f ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    b0 = torch.zeros(c)
    for bias in (b, b0):
        xd, wd, bd = x.to(cuda), w.to(cuda), bias.to(cuda)
        st = torch.cuda.current_stream().cuda_stream
        h_ref = torch.empty(n, c, device=cuda)
        _lib.check(lib.sngnn_linear_forward(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), n, f, c, h_ref.data_ptr(), st), "lin")
        h = torch.empty(n, c, device=cuda)
        un = torch.full((n, c), float("nan"), device=cuda)
        nrm = torch.full((n,), float("nan"), device=cuda)
        fb = ops.filter_row_bytes(c)
        filt = torch.full((n, fb), 0xAB, dtype=torch.uint8, device=cuda) if fb == 128 else None
        _lib.check(lib.sngnn_linear_forward_normalized(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), n, f, c, h.data_ptr(),
                                                       un.data_ptr(), nrm.data_ptr(), _lib.ptr(filt), st), "lin+norm")
        assert torch.equal(h, h_ref)
        un2, nrm2, filt2 = ops.normalize_rows_filter(h_ref)
        assert torch.equal(un, un2) and torch.equal(nrm, nrm2)
        if filt is not None:
            assert torch.equal(filt, filt2)
    # with a zero bias row 9 of h is zero: unit row 0, norm clamped at eps
    assert (un[9] == 0).all() and float(nrm[9]) == pytest.approx(1e-12)


def test_model_path_uses_the_epilogue_and_matches_the_pass(cuda):
    """SNConv_plus on a graph big enough for the hand-written lin: the forward that takes the
    unit rows from lin's epilogue equals the one that runs the normalisation pass, bit for bit,
    and so do the gradients."""
    import sngnn_amd
    from sngnn_amd import ops
    from tests.helpers import random_graph
    n, f, c = 5000, 64, 40
    ei = random_graph(n, 60000, seed=3, hubs=((0, 4000), (5, 900))).to(cuda)
    x = torch.randn(n, f, generator=torch.Generator().manual_seed(1)).to(cuda)
    gout = torch.randn(n, c, generator=torch.Generator().manual_seed(2)).to(cuda)
    torch.manual_seed(0)
    conv = sngnn_amd.SNConv_plus(f, c, n, top_k=8, thr=0.1).to(cuda)
    seen = {}
    orig, orig_e = ops._forward_prepared, ops._forward_epilogue
    ops._forward_prepared = lambda *a, **k: (seen.__setitem__("prepared", True), orig(*a, **k))[1]
    # (a training call that saves the kept bits goes through _forward_epilogue: it takes the prepared
    # entry point exactly when ``unit`` - its third argument - carries unit rows)
    ops._forward_epilogue = lambda *a, **k: (seen.__setitem__("prepared", a[2] is not None and a[2].n is not None),
                                             orig_e(*a, **k))[1]
    try:
        out = conv(x, ei)
        (out * gout).sum().backward()
    finally:
        ops._forward_prepared, ops._forward_epilogue = orig, orig_e
    assert seen.get("prepared"), "the conv did not take the unit rows from lin's epilogue"
    g1 = [p.grad.clone() for p in conv.parameters()]
    conv.zero_grad()
    h = ops.linear(x, conv.lin)
    out2 = ops.aggregate(h, sngnn_amd.graph.GLOBAL_CACHE.get(ei, n, True, True), 8, 0.1)
    (out2 * gout).sum().backward()
    assert torch.equal(out, out2)
    for a, p in zip(g1, conv.parameters()):
        assert torch.equal(a, p.grad)


@pytest.mark.parametrize("n,f,c,scale", [(6000, 128, 40, 1.0), (5000, 64, 64, 1e-3), (4100, 32, 17, 1e4),
                                         (4500, 128, 48, 1.0), (4097, 64, 5, 1.0)])
def test_lin_on_the_bf16_matrix_cores_rounds_like_fp32(cuda, n, f, c, scale):
    """sngnn_tuning_set(5, mode): k_linear_rows multiplies on the bf16 matrix cores after an EXACT
    three-way split of both operands (eight partial products, fp32 accumulation; default) or with
    fp32 MFMAs.  Against float64: both forms are fp32 dot products - the split form's error is no
    larger than the fp32 form's (to 1.5x + one ulp of the result's scale), and far below what a
    bf16 or even a tf32-like product would give; cancelling rows, a wide dynamic range inside a row,
    a row with an infinity or a NaN comes out NaN (documented deviation: fp32 gives +-inf for inf)."""
    from sngnn_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + f + c)
    x = torch.randn(n, f, generator=g) * scale
    x[:, ::3] *= 2.0 ** -9                                   # a wide dynamic range inside every row
    w = torch.randn(c, f, generator=g) / f ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    x[7] = 0.0
    x[11, 1::2] = -x[11, 0::2]                               # products that cancel: set w so that they do
    w[:, 1::2] = w[:, 0::2]
    ref = (x.double() @ w.double().t() + b.double())
    xd, wd, bd = x.to(cuda), w.to(cuda), b.to(cuda)
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    try:
        for mode in (0, 1):
            lib.sngnn_tuning_set(5, mode)
            h = torch.empty(n, c, device=cuda)
            _lib.check(lib.sngnn_linear_forward(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), n, f, c, h.data_ptr(), st), "lin")
            out[mode] = h.cpu().double()
    finally:
        lib.sngnn_tuning_set(5, 0)
    # the bound of an fp32 dot product: (f + 2) roundings of 2^-24 on the sum of |x w|
    mag = x.double().abs() @ w.double().abs().t() + b.double().abs()
    err = {m: ((out[m] - ref).abs() / mag.clamp_min(1e-300)).max().item() for m in out}
    assert err[1] <= (f + 2) * 2.0 ** -24 and err[0] <= (f + 2) * 2.0 ** -24, err
    assert err[0] <= 1.5 * err[1] + 2.0 ** -24, err
    assert torch.equal(out[0][7], b.double()) and (out[0][11] - b.double()).abs().max() <= 1e-6 * scale
    # non-finite inputs: the affected rows come out non-finite (NaN: the split of an infinity
    # holds inf - inf), every other row is untouched
    x2 = x[:4096].clone()
    x2[3, 5] = float("inf")
    x2[4, 6] = float("-inf")
    x2[5, 0] = float("nan")
    h2 = torch.empty(4096, c, device=cuda)
    xx = x2.to(cuda)
    _lib.check(lib.sngnn_linear_forward(xx.data_ptr(), wd.data_ptr(), bd.data_ptr(), 4096, f, c, h2.data_ptr(), st), "lin")
    got = h2.cpu()
    assert not torch.isfinite(got[3:6]).any()
    keep = torch.ones(4096, dtype=torch.bool)
    keep[3:6] = False
    assert torch.equal(got[keep].double(), out[0][:4096][keep])
