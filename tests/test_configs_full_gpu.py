"""BASELINE.json's configs 1-3 at their FULL workload sizes: the drop-in
modules against the oracle's restatement of the reference classes - log-probs, every
parameter gradient, and the per-row selections of every selecting layer (differing rows are
counted: they must be ulp-level near ties, tests/helpers.model_selection_report).

config 1  Cora       SNGNN 1 layer                       N 2 708  E 10 556  F 1 433  C 7
config 2  Chameleon  SNGNN_Plus top_k 10 thr 0.9 hidden 32, 1 layer (README.md:63) and 2
                                                          N 2 277  E 36 101  F 2 325  C 5 / 32 -> 5
config 3  Actor      SNGNN_Plus_Plus top_k 10 thr 0.9 init_beta 0.0
                                                          N 7 600  E 30 019  F 932    C 5 / 32 -> 5
config 4  arxiv      SNGNN_Plus top_k 16 thr 0.0, 1 layer (C 40) and 2 layers (32 -> 40)
                                                          N 169 343  E 1 166 243  F 128
(config 4's operator alone, forward and backward against the C / torch oracles: tests/test_golden_gpu.py;
config 5: tests/test_products_gpu.py)."""
import pytest
import torch
import torch.nn.functional as F

from sngnn_amd import synth
from tests.helpers import assert_close, model_selection_report
from tests.test_models_gpu import build_pair

pytestmark = pytest.mark.gpu


def _check(cuda, kind, args, data, label, train_modes=(False,), lin_is_blas=None):
    from tests import helpers
    ours, ref = build_pair(kind, args)
    ours = ours.to(cuda)
    d = data.to(cuda)
    worst_out = 0.0
    for train in train_modes:
        ours.train(train), ref.train(train)
        out_ref = ref(data)
        out = ours(d)
        assert out.shape == out_ref.shape
        # the north star's own gate (1e-5 rtol, + the 2e-6 absolute floor of tests/helpers.py) at MODEL level too -
        # although the two sides' ``lin`` ran on different devices and a near-tie flip moves whole rows: measured at
        # most 0.54x of it over configs 1-4 at full size (round 5; the gate was 1e-4 / 2e-5 before)
        assert_close(out, out_ref, what=f"{label} log-probs", rtol=1e-5, atol=2e-6)
        worst_out = max(worst_out, float(((out.detach().cpu() - out_ref.detach()).abs() /
                                          (2e-6 + 1e-5 * out_ref.detach().abs())).max()))
    ours.zero_grad(), ref.zero_grad()
    F.nll_loss(ref(data)[data.train_mask], data.y[data.train_mask]).backward()
    F.nll_loss(ours(d)[d.train_mask], d.y[d.train_mask]).backward()
    worst_g = 0.0
    for (name, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, name
        gq = q.grad.to_dense() if q.grad.is_sparse else q.grad
        scale = gq.abs().max().clamp_min(1e-12)
        err = (p.grad.cpu() - gq).abs().max()
        # (2e-5 of the gradient's max-norm - measured 2.5e-7 .. 8.2e-6 - plus an absolute floor for the parameters
        # whose whole gradient is ~1e-4: there a second layer's near-tie flips, counted above, are the error)
        assert err <= 2e-5 * scale + 2e-7, f"{label}.{name}: err {err.item():.3e} scale {scale.item():.3e}"
        worst_g = max(worst_g, float(err / scale))
    helpers.REPORT_LINES.append(f"{label}: log-probs at {worst_out:.2f}x the operator gate (1e-5 rtol + 2e-6), "
                                f"worst parameter gradient {worst_g:.2e} of its max-norm (gates: 1e-5 + 2e-6; 2e-5 + 2e-7 abs)")
    return model_selection_report(ours, ref, data, d, label)


def test_config1_cora_sngnn_full_size(cuda):
    data = synth.make_dataset("cora")
    n, f = data.x.shape
    assert (n, f) == (2708, 1433) and abs(data.edge_index.size(1) - 10556) <= 11
    ours, ref = build_pair("SNGNN", (f, 32, 7, 1))
    ours, d = ours.to(cuda), data.to(cuda)
    ours.eval(), ref.eval()                      # SNGNN's dropout is hard-wired to 0.5 (models.py:283)
    assert_close(ours(d), ref(data), what="cora SNGNN log-probs", rtol=1e-5, atol=2e-6)
    ours.dropout.p = ref.dropout.p = 0.0
    ours.train(), ref.train()
    F.nll_loss(ref(data)[data.train_mask], data.y[data.train_mask]).backward()
    F.nll_loss(ours(d)[d.train_mask], d.y[d.train_mask]).backward()
    for (name, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        scale = q.grad.abs().max().clamp_min(1e-12)
        assert (p.grad.cpu() - q.grad).abs().max() <= 2e-5 * scale + 2e-7, name
    # two layers: the hidden conv at C = 32, dropout off
    ours, ref = build_pair("SNGNN", (f, 32, 7, 2))
    ours = ours.to(cuda)
    ours.eval(), ref.eval()
    assert_close(ours(d), ref(data), what="cora SNGNN 2-layer log-probs", rtol=1e-5, atol=2e-6)


@pytest.mark.parametrize("layers", [1, 2])
def test_config2_chameleon_sngnn_plus_full_size(cuda, layers):
    """F = 2 325 -> the rocBLAS ``lin`` branch; max in-degree 732 (split rows, candidate
    finalize); top_k 10, thr 0.9, loops removed (README.md:63)."""
    data = synth.make_dataset("chameleon")
    n, f = data.x.shape
    assert (n, f) == (2277, 2325) and abs(data.edge_index.size(1) - 36101) <= 36
    deg = torch.bincount(data.edge_index[1], minlength=n)
    assert int(deg.max()) >= 600      # 732 drawn, duplicates coalesced: split-row class (> 128)
    differ, rows = _check(cuda, "SNGNN_Plus", (f, 32, 5, n, layers, 10, 0.9, 1, 0.0), data,
                          f"chameleon SNGNN_Plus {layers}-layer", train_modes=(True, False))
    assert rows == n * layers and differ <= max(2, rows // 500)
    # the same model with a threshold that keeps edges (thr 0.9 on 1 %-dense bag-of-words rows
    # keeps little besides duplicates): selection and weighted mean at this size
    differ, rows = _check(cuda, "SNGNN_Plus", (f, 32, 5, n, layers, 10, 0.0, 1, 0.0), data,
                          f"chameleon SNGNN_Plus {layers}-layer thr=0")
    # (2 layers: the second conv sees 5-channel rows that are nearly parallel - every cosine
    # within 1e-4 of 1 - so rows whose 10th and 11th candidates are one ulp apart are common:
    # ~30 of 2 277 there, each checked to BE an ulp-level tie and counted in the summary)
    assert differ <= (max(2, rows // 500) if layers == 1 else rows // 50)


@pytest.mark.parametrize("layers", [1, 2])
def test_config3_actor_sngnn_plus_plus_full_size(cuda, layers):
    data = synth.make_dataset("actor")
    n, f = data.x.shape
    assert (n, f) == (7600, 932)
    differ, rows = _check(cuda, "SNGNN_Plus_Plus", (f, 32, 5, n, layers, 10, 0.9, 0.0, 1, 0.0), data,
                          f"actor SNGNN_Plus_Plus {layers}-layer", train_modes=(True, False))
    assert differ <= max(2, rows // 500)
    # the same model where the selection and BOTH branches matter: thr 0.0 keeps up to 10 edges a row
    # (at thr 0.9 the bag-of-words rows keep ~400 of 29 940 edges) and init_beta 0.3 blends the
    # adjacency branch in, so every parameter of both branches gets a gradient
    differ, rows = _check(cuda, "SNGNN_Plus_Plus", (f, 32, 5, n, layers, 10, 0.0, 0.3, 1, 0.0), data,
                          f"actor SNGNN_Plus_Plus {layers}-layer thr=0 beta=0.3", train_modes=(True, False))
    assert differ <= (max(2, rows // 500) if layers == 1 else rows // 50)


@pytest.mark.parametrize("layers", [1, 2])
def test_config4_arxiv_sngnn_plus_full_size(cuda, layers):
    """The MODEL at config 4's size: 128 -> 40 (README.md:63's shape of a run) and 128 -> 32 -> 40, top_k 16,
    thr 0.0, loops removed; dropout 0 so that the two sides see the same activations.  Log-probs, every
    parameter gradient, and the per-row selections of every layer (near ties counted)."""
    data = synth.make_dataset("arxiv")
    n, f = data.x.shape
    assert (n, f) == (169343, 128) and int(torch.bincount(data.edge_index[1], minlength=n).max()) > 10000
    differ, rows = _check(cuda, "SNGNN_Plus", (f, 32, 40, n, layers, 16, 0.0, 1, 0.0), data,
                          f"arxiv SNGNN_Plus {layers}-layer", train_modes=(True,))
    # (1 layer: no differing row.  2 layers: the second conv sees rows that are nearly parallel - every cosine
    # within 1.3e-5 of 1 - so its 16th and 17th candidates are often a few ulps apart: 119 of its 169 343
    # rows, each checked to BE such a tie - largest gap 3.5 ulps - and counted in the summary)
    assert rows == n * layers and differ <= (rows // 5000 if layers == 1 else rows // 1000)
