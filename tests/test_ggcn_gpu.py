"""GPU: GGCN's sparse layer (models.py:1453-1553) on the signed-attention kernels against the
committed fixtures - made by running the REFERENCE class itself (tests/golden/pin_reference.py:
pin_ggcn; all core torch, no third-party stand-in on its path) - and against the oracle's
restatement on graphs with split rows, wave rows and isolated nodes."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import sngnn_oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _layer_pair(f, c, kw, state, dev):
    from sngnn_amd.ggcn import GGCNlayer_SP
    ours, ref = GGCNlayer_SP(f, c, dev, **kw), O.GGCNlayer_SP(f, c, "cpu", **kw)
    assert list(ours.state_dict()) == list(ref.state_dict())
    ref.load_state_dict(state)
    ours.load_state_dict(state)
    return ours.to(dev), ref


def _compare(ours, ref, adj, dp, h, gout, dev, want=None):
    hg = h.to(dev).requires_grad_(True)
    out = ours(hg, adj.to(dev), None if dp is None else dp.to(dev))
    (out * gout.to(dev)).sum().backward()
    hr = h.clone().requires_grad_(True)
    out_r = ref(hr, adj, dp)
    (out_r * gout).sum().backward()
    if want is not None:                                   # the reference's own outputs
        out_r_np, grad_h_np, grads = want
        # (bit for bit on the machine that made the fixture; another host CPU blocks torch's GEMM and
        # sparse products differently, hence rounding here)
        np.testing.assert_allclose(out_r.detach().numpy(), out_r_np, rtol=0, atol=2e-6 * np.abs(out_r_np).max())
        np.testing.assert_allclose(hr.grad.numpy(), grad_h_np, rtol=0, atol=2e-6 * np.abs(grad_h_np).max())
    scale = float(out_r.detach().abs().max())
    assert float((out.detach().cpu() - out_r.detach()).abs().max()) <= 1e-5 * scale + 1e-6
    gs = float(hr.grad.abs().max())
    assert float((hg.grad.cpu() - hr.grad).abs().max()) <= 2e-5 * gs
    for (k, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        sc = max(float(q.grad.abs().max()), 1e-6)
        err = float((p.grad.cpu() - q.grad).abs().max())
        # scalars (deg_coeff, coeff, scale) are signed sums over every edge: priced against the
        # size of what is summed, not of what is left
        tol = 2e-5 * sc if q.grad.numel() > 3 else 2e-5 * sc + 1e-5
        assert err <= tol, (k, err, sc)
        if want is not None:
            np.testing.assert_allclose(q.grad.numpy(), want[2]["grad_" + k.replace(".", "_")], rtol=0,
                                       atol=2e-6 * max(np.abs(q.grad.numpy()).max(), 1.0))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "ggcn_sp_*.npz"))))
def test_layer_against_the_reference_made_fixtures(cuda, path):
    z = np.load(path)
    n, f, c = (int(v) for v in z["n"])
    kw = dict(zip(("use_degree", "use_sign", "use_decay"), (bool(v) for v in z["flags"])))
    adj = torch.sparse_coo_tensor(torch.from_numpy(z["adj_indices"]), torch.from_numpy(z["adj_values"]), (n, n)).coalesce()
    dp = torch.sparse_coo_tensor(adj._indices(), torch.from_numpy(z["degree_values"]), (n, n)).coalesce()
    state = {k[6:].replace("fcn_", "fcn."): torch.from_numpy(z[k]) for k in z.files if k.startswith("param_")}
    ours, ref = _layer_pair(f, c, kw, state, cuda)
    grads = {k: z[k] for k in z.files if k.startswith("grad_") and k != "grad_h"}
    _compare(ours, ref, adj, dp, torch.from_numpy(z["h"]), torch.from_numpy(z["gout"]), cuda,
             want=(z["out"], z["grad_h"], grads))


def _case(n, e, f, seed, hubs=()):
    from tests.helpers import random_graph
    ei = random_graph(n - 2, e, seed=seed, hubs=hubs)
    a = torch.zeros(n, n)
    a[ei[1], ei[0]] = 1.0
    a = ((a + a.t()) > 0).float()
    a.fill_diagonal_(1.0)
    d = a.sum(1)
    adj = (a / torch.sqrt(d[:, None] * d[None, :])).to_sparse().coalesce()
    gen = torch.Generator().manual_seed(seed)
    h = torch.randn(n, f, generator=gen)
    h[5] = h[6]
    return adj, h, gen


@pytest.mark.parametrize("c,kw", [(7, dict()), (32, dict(use_decay=False)), (48, dict(use_degree=False)), (130, dict()),
                                  (5, dict(use_sign=False))])
def test_layer_against_the_oracle_with_hubs_and_isolated_nodes(cuda, c, kw):
    from sngnn_amd.ggcn import precompute_degree_s
    n, f = 3000, 20
    adj, h, gen = _case(n, 12000, f, seed=c, hubs=((11, 2500), (12, 700), (13, 129), (14, 128), (15, 17)))
    dp = O.ggcn_degree_precompute(adj)
    assert torch.equal(precompute_degree_s(adj.to(cuda))._values().cpu(), dp._values())
    torch.manual_seed(c)
    ref = O.GGCNlayer_SP(f, c, "cpu", **kw)
    with torch.no_grad():
        if kw.get("use_sign", True):
            ref.coeff.copy_(torch.tensor([0.5, -0.3, 0.2]))
        if kw.get("use_degree", True):
            ref.deg_coeff.copy_(torch.tensor([0.4, -0.1]))
        ref.fcn.weight[0].zero_()          # a zero channel
    # (an exactly zero ROW of Wh is not compared: F.cosine_similarity clamps the norm at 1e-8 where the
    # kernels clamp at F.normalize's 1e-12, so such a row's 1 / eps-sized gradient differs by that
    # ratio - and between torch versions: 1.10 clamps the PRODUCT of the two norms)
    ours, ref = _layer_pair(f, c, kw, ref.state_dict(), cuda)
    gout = torch.randn(n, c, generator=gen)
    _compare(ours, ref, adj, dp, h, gout, cuda)


def test_signed_propagate_is_deterministic_and_caches_its_structure(cuda):
    from sngnn_amd.ggcn import GGCNlayer_SP, precompute_degree_s
    adj, h, gen = _case(800, 5000, 16, seed=3, hubs=((2, 500),))
    adj = adj.to(cuda)
    dp = precompute_degree_s(adj)
    torch.manual_seed(0)
    layer = GGCNlayer_SP(16, 24, cuda).to(cuda)
    outs = []
    for _ in range(3):
        hg = h.to(cuda).requires_grad_(True)
        out = layer(hg, adj, dp)
        out.square().sum().backward()
        outs.append((out.detach().clone(), hg.grad.clone()))
        st = layer._structure
    assert all(torch.equal(outs[0][0], o) and torch.equal(outs[0][1], g) for o, g in outs[1:])
    assert layer._structure is st
    with pytest.raises(ValueError):
        layer(h, adj, dp)                                  # CPU features: no CPU path


def test_layer_at_arxiv_size_against_the_oracle(cuda):
    """GGCNlayer_SP on the symmetrised, self-looped, symmetrically normalised adjacency of BASELINE config
    4's graph (169 343 nodes, 2.4 M off-diagonal entries; the hub's row has > 13 000 of them: split rows
    of the signed-attention kernels), C = 40: output, grad_h and all five parameter gradients."""
    from sngnn_amd import synth
    from sngnn_amd.ggcn import precompute_degree_s
    d = synth.make_dataset("arxiv", with_features=False)
    n = d.x.size(0)
    ei = d.edge_index
    ei = ei[:, ei[0] != ei[1]]
    both = torch.cat([ei, ei.flip(0), torch.arange(n).repeat(2, 1)], dim=1)
    a = torch.sparse_coo_tensor(both, torch.ones(both.size(1)), (n, n)).coalesce()
    idx = a._indices()
    deg = torch.zeros(n).index_add_(0, idx[0], torch.ones(idx.size(1)))          # duplicates merged: 0/1 pattern
    val = 1.0 / torch.sqrt(deg[idx[0]] * deg[idx[1]])
    adj = torch.sparse_coo_tensor(idx, val, (n, n)).coalesce()
    dp = O.ggcn_degree_precompute(adj)
    assert torch.equal(precompute_degree_s(adj.to(cuda))._values().cpu(), dp._values())
    f, c = 24, 40
    gen = torch.Generator().manual_seed(21)
    h = torch.randn(n, f, generator=gen)
    gout = torch.randn(n, c, generator=gen)
    torch.manual_seed(21)
    ref = O.GGCNlayer_SP(f, c, "cpu")
    with torch.no_grad():
        ref.coeff.copy_(torch.tensor([0.5, -0.3, 0.2]))
        ref.deg_coeff.copy_(torch.tensor([0.4, -0.1]))
    ours, ref = _layer_pair(f, c, dict(), ref.state_dict(), cuda)
    _compare(ours, ref, adj, dp, h, gout, cuda)
