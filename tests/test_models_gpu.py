"""Model-level parity: the drop-in nn.Modules against the oracle's restatement of
the reference classes (same seed -> same parameters -> same outputs and grads)."""
import pytest
import torch
import torch.nn.functional as F

from oracle import sngnn_oracle as O
from sngnn_amd.synth import Data
from tests.helpers import assert_close, random_graph

pytestmark = pytest.mark.gpu


def build_pair(kind, args, seed=1234):
    import sngnn_amd
    torch.manual_seed(seed)
    ours = getattr(sngnn_amd, kind)(*args)
    torch.manual_seed(seed)
    ref = getattr(O, kind)(*args)
    for (k1, v1), (k2, v2) in zip(ours.state_dict().items(), ref.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    return ours, ref


MODELS = [
    ("SNGNN", lambda F_, N: (F_, 16, 7, 1)),
    ("SNGNN", lambda F_, N: (F_, 16, 7, 2, True)),
    ("SNGNN_Plus", lambda F_, N: (F_, 32, 5, N, 1, 10, 0.9, 1, 0.0)),
    ("SNGNN_Plus", lambda F_, N: (F_, 32, 5, N, 2, 3, 0.0, 0, 0.0, True)),
    ("SNGNN_Plus_Plus", lambda F_, N: (F_, 32, 5, N, 1, 10, 0.9, 0.0, 1, 0.0)),
    ("SNGNN_Plus_Plus", lambda F_, N: (F_, 32, 5, N, 2, 4, 0.1, 0.3, 1, 0.0, True)),
    ("SNGNN_Plus_Plus", lambda F_, N: (F_, 24, 6, N, 3, 2, 0.0, 0.5, 0, 0.0)),
]


@pytest.mark.parametrize("kind,mk", MODELS)
def test_model_forward_backward(cuda, kind, mk):
    n, f = 300, 40
    ei = random_graph(n, 2500, seed=11, hubs=((0, 299), (4, 160)))
    gen = torch.Generator().manual_seed(5)
    x = (torch.rand(n, f, generator=gen) < 0.1).float()       # bag-of-words like
    x[10] = x[11]
    y = torch.randint(0, 5, (n,), generator=gen)
    mask = torch.rand(n, generator=gen) < 0.6
    ours, ref = build_pair(kind, mk(f, n))
    ours = ours.to(cuda)
    # the adjacency table keeps its gather-friendly layout through .to()
    for m in ours.modules():
        if hasattr(m, "w") and hasattr(m.w, "weight"):
            assert m.w.weight.t().is_contiguous()
    # SNGNN's dropout is hard-wired to 0.5 (models.py:283): compare in eval mode only
    modes = (False,) if kind == "SNGNN" else (True, False)
    for train in modes:
        ours.train(train)
        ref.train(train)
        out_ref = ref(Data(x=x, edge_index=ei))
        out = ours(Data(x=x.to(cuda), edge_index=ei.to(cuda)))
        assert_close(out, out_ref, what=f"{kind} log-probs", rtol=1e-4, atol=2e-5)
    if kind == "SNGNN":
        ours.dropout.p = ref.dropout.p = 0.0
    ours.train()
    ref.train()
    ours.zero_grad()
    ref.zero_grad()
    F.nll_loss(ref(Data(x=x, edge_index=ei))[mask], y[mask]).backward()
    d = Data(x=x.to(cuda), edge_index=ei.to(cuda))
    F.nll_loss(ours(d)[mask.to(cuda)], y.to(cuda)[mask.to(cuda)]).backward()
    for (name, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        assert p.grad is not None, name
        gq = q.grad.to_dense() if q.grad.is_sparse else q.grad
        scale = gq.abs().max().clamp_min(1e-12)
        err = (p.grad.cpu() - gq).abs().max()
        assert err <= 2e-4 * scale + 1e-7, f"{kind}.{name}: err {err.item():.3e} scale {scale.item():.3e}"


def test_state_dict_round_trip(cuda):
    ours, ref = build_pair("SNGNN_Plus_Plus", (12, 8, 3, 50, 2, 4, 0.1, 0.3, 1, 0.5, True))
    ours2, _ = build_pair("SNGNN_Plus_Plus", (12, 8, 3, 50, 2, 4, 0.1, 0.3, 1, 0.5, True), seed=7)
    ours2.load_state_dict(ref.state_dict())          # a reference checkpoint loads as is
    for (k, v), (_, w) in zip(ours2.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(v.cpu(), w), k
    ours2 = ours2.to(cuda)
    assert ours2.lins[0].w.weight.shape == (8, 50)
    assert ours2.lins[0].w.weight.t().is_contiguous()


@pytest.mark.parametrize("kind", ["SNGNN_Plus", "SNGNN_Plus_Plus"])
def test_graphed_epoch_matches_eager_trajectory(cuda, kind):
    """The HIP-graph epoch is the same arithmetic as the reference-style eager loop
    (SNGNN++: fused blend, fused Adam over the column-major w.weight)."""
    import sngnn_amd
    from sngnn_amd import train as T
    from sngnn_amd import synth
    data = synth.make_dataset("cora", scale=0.5).to(cuda)
    n, f = data.x.shape
    runs = []
    for graphed in (False, True):
        torch.manual_seed(11)
        if kind == "SNGNN_Plus":
            model = sngnn_amd.SNGNN_Plus(f, 16, 7, n, 2, 3, 0.1, 1, 0.0).to(cuda)
        else:
            model = sngnn_amd.SNGNN_Plus_Plus(f, 16, 7, n, 2, 3, 0.1, 0.4, 1, 0.0).to(cuda)
        # same optimizer flavour on both sides: torch's capturable Adam keeps the step
        # count and bias corrections on the device (fp32), which alone moves the
        # trajectory by ~1e-3 relative to the default host-side (fp64) bookkeeping
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4, capturable=True)
        fn = T.train_graphed if graphed else T.train
        runs.append(fn(model, data, opt, epochs=6, patience=100))
    for a, b in zip(runs[0]["history"], runs[1]["history"]):
        for k in ("train_loss", "val_loss", "test_loss", "train_acc", "val_acc", "test_acc"):
            assert abs(a[k] - b[k]) <= 1e-4 * max(1.0, abs(a[k])), (a["epoch"], k, a[k], b[k])


def test_graphed_epoch_survives_graph_cache_eviction(cuda):
    """A captured epoch holds raw pointers into its graphs' structure arrays and workspaces;
    building more graphs than the cache holds (16, FIFO) between replays must neither free
    them nor change the replayed metrics."""
    import sngnn_amd
    from sngnn_amd import train as T
    from sngnn_amd import synth
    from sngnn_amd.graph import GLOBAL_CACHE
    from tests.helpers import random_graph
    data = synth.make_dataset("cora", scale=0.5).to(cuda)
    n, f = data.x.shape

    def run(evict):
        torch.manual_seed(5)
        model = sngnn_amd.SNGNN_Plus(f, 16, 7, n, 2, 3, 0.1, 1, 0.0).to(cuda)
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4, capturable=True)
        ge = T.GraphedEpoch(model, data, opt, warmup=0)
        recs = []
        for ep in range(4):
            if evict:
                keep = []
                for q in range(20):                 # more than the cache holds: everything older is evicted
                    ei = random_graph(300 + q, 2000, seed=100 * ep + q).to(cuda)
                    keep.append(GLOBAL_CACHE.get(ei, 300 + q, True, bool(q % 2)))
                    scratch = torch.full((200_000,), float(q), device=cuda)     # recycle freed memory
                del keep, scratch
            recs.append(ge.run())
        return recs

    a, b = run(False), run(True)
    for ra, rb in zip(a, b):
        for k in ra:
            assert abs(ra[k] - rb[k]) <= 1e-6 * max(1.0, abs(ra[k])), (k, ra[k], rb[k])


def test_graphed_epoch_one_eval_forward_equals_two(cuda):
    """validate_step and test_step (train.py:92-117) run the same eval-mode forward; the graphed
    epoch runs it once by default.  Every metric must be bit-identical to the two-pass replay."""
    import sngnn_amd
    from sngnn_amd import train as T
    from sngnn_amd import synth
    data = synth.make_dataset("cora", scale=0.5).to(cuda)
    n, f = data.x.shape

    def run(share):
        torch.manual_seed(11)
        model = sngnn_amd.SNGNN_Plus(f, 16, 7, n, 2, 3, 0.1, 1, 0.5, bn=True).to(cuda)
        opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4, capturable=True)
        ge = T.GraphedEpoch(model, data, opt, warmup=0, share_eval_forward=share)
        return [ge.run() for _ in range(5)]

    a, b = run(True), run(False)
    assert a == b


@pytest.mark.parametrize("kind,args,name", [
    ("SNGNN_Plus", lambda f, n: (f, 16, 7, n, 2, 3, 0.1, 1, 0.0), "plus_2layer"),
    ("SNGNN_Plus_Plus", lambda f, n: (f, 16, 7, n, 1, 4, 0.2, 0.3, 1, 0.0), "plusplus_1layer"),
    ("SNGNN", lambda f, n: (f, 16, 7, 1), "sngnn_1layer"),
    ("AGNN", lambda f, n: (f, 16, 7, 1), "agnn_1layer"),
])
def test_training_trajectory_matches_committed_fixture(cuda, kind, args, name):
    """SURVEY.md 8 harness row: 5 epochs of the reference loop (train.py:73-160) on the
    committed fixture (tests/golden/traj_*.npz, made on the CPU by the oracle's models):
    same initial parameters, loss / accuracy trajectory and final parameters."""
    import os
    import numpy as np
    import sngnn_amd
    from sngnn_amd import synth
    from sngnn_amd import train as T
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", f"traj_{name}.npz"))
    data = synth.make_dataset("cora", seed=7, scale=0.25)
    torch.manual_seed(1234)
    model = getattr(sngnn_amd, kind)(*args(data.x.size(1), data.x.size(0)))
    for k, v in model.state_dict().items():
        assert np.array_equal(v.numpy(), z["init." + k]), k
    model = model.to(cuda)
    d = data.to(cuda)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    res = T.train(model, d, opt, epochs=5, patience=100)
    got = np.array([[h["train_loss"], h["val_loss"], h["val_acc"], h["test_loss"], h["test_acc"]]
                    for h in res["history"]])
    want = z["traj"]
    assert np.allclose(got[:, [0, 1, 3]], want[:, [0, 1, 3]], rtol=1e-4, atol=1e-5), (got, want)
    assert np.allclose(got[:, [2, 4]], want[:, [2, 4]], atol=2.0 / d.x.size(0) + 1e-9)
    for k, v in model.state_dict().items():
        w = z["final." + k]
        # Adam turns a last-bit difference of a near-zero gradient into a visible step;
        # 5e-5 is 0.1 % of the 0.05 a parameter can move in 5 steps at lr 0.01
        assert np.abs(v.cpu().numpy() - w).max() <= 5e-5, k


def test_odd_channel_count_takes_the_padded_path_and_matches_oracle(cuda):
    """C = 47 (ogbn-products' class count): h is produced with one zero channel so rows
    are 16-byte aligned; outputs, gradients and the state_dict are unaffected."""
    n, f, c = 400, 30, 47
    ei = random_graph(n, 5000, seed=3, hubs=((0, 399), (7, 150)))
    gen = torch.Generator().manual_seed(2)
    x = torch.randn(n, f, generator=gen)
    y = torch.randint(0, c, (n,), generator=gen)
    ours, ref = build_pair("SNGNN_Plus", (f, 47, c, n, 2, 8, 0.0, 1, 0.0))
    ours = ours.to(cuda)
    assert ours.lins[0].lin.weight.shape == (47, f)
    out_ref = ref(Data(x=x, edge_index=ei))
    out = ours(Data(x=x.to(cuda), edge_index=ei.to(cuda)))
    assert out.shape == (n, c)
    assert_close(out, out_ref, what="log-probs", rtol=1e-4, atol=2e-5)
    F.nll_loss(out_ref, y).backward()
    F.nll_loss(out, y.to(cuda)).backward()
    for (name, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
        scale = q.grad.abs().max().clamp_min(1e-12)
        assert (p.grad.cpu() - q.grad).abs().max() <= 2e-4 * scale + 1e-7, name


def test_actor_topology_sngnn_plus_plus(cuda):
    """BASELINE config 3 on the REAL Actor graph (tests/golden/actor_topology.npz: 30 019
    edges, 93 self-loops, max in-degree 1296, 29 % isolated targets) with synthetic
    bag-of-words features: SNGNN_Plus_Plus top_k=10 thr=0.9 init_beta=0.0, forward and
    gradients against the oracle."""
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "actor_topology.npz"))
    ei = torch.from_numpy(z["edge_index"].astype(np.int64))
    n, f, c = 7600, 64, 5
    gen = torch.Generator().manual_seed(8)
    x = (torch.rand(n, f, generator=gen) < 0.08).float()
    x[100] = x[101]
    y = torch.from_numpy(z["y"].astype(np.int64))
    mask = torch.from_numpy(z["train_mask0"])
    for args in ((f, 32, c, n, 1, 10, 0.9, 0.0, 1, 0.0), (f, 32, c, n, 2, 10, 0.5, 0.3, 0, 0.0)):
        ours, ref = build_pair("SNGNN_Plus_Plus", args)
        ours = ours.to(cuda)
        out_ref = ref(Data(x=x, edge_index=ei))
        out = ours(Data(x=x.to(cuda), edge_index=ei.to(cuda)))
        assert_close(out, out_ref, what="log-probs", rtol=1e-4, atol=2e-5)
        F.nll_loss(out_ref[mask], y[mask]).backward()
        F.nll_loss(out[mask.to(cuda)], y.to(cuda)[mask.to(cuda)]).backward()
        for (name, p), (_, q) in zip(ours.named_parameters(), ref.named_parameters()):
            gq = q.grad.to_dense() if q.grad.is_sparse else q.grad
            scale = gq.abs().max().clamp_min(1e-12)
            assert (p.grad.cpu() - gq).abs().max() <= 2e-4 * scale + 1e-7, name


def test_real_actor_published_hyperparameters(cuda):
    """BASELINE config 3 end to end on the REAL Actor data the reference bundles (features,
    labels, geom-gcn split 0; fixtures tests/golden/actor_{topology,features}.npz) with the
    published sweep's hyper-parameters (train_script_SNGNN_plus_plus.sh:5-44: lr 0.1, weight
    decay 5e-4, dropout 0, 1 layer, top_k 1, thr 0.99, self-loops kept, init_beta 0.3, seed
    1234): 8 epochs of the trainer's loop against the trajectory the reference's own
    SNGNN_Plus_Plus class produced on the CPU (tests/golden/pin_reference.py)."""
    import os
    import numpy as np
    import sngnn_amd
    from sngnn_amd import train as T
    gdir = os.path.join(os.path.dirname(__file__), "golden")
    topo, feat = np.load(os.path.join(gdir, "actor_topology.npz")), np.load(os.path.join(gdir, "actor_features.npz"))
    z = np.load(os.path.join(gdir, "traj_actor_real_plusplus.npz"))
    n, f = (int(v) for v in feat["shape"])
    x = torch.zeros(n, f)
    x[torch.from_numpy(feat["row"].astype(np.int64)), torch.from_numpy(feat["col"].astype(np.int64))] = \
        torch.from_numpy(feat["val"])
    assert (n, f) == (7600, 932)
    data = Data(x=x, edge_index=torch.from_numpy(topo["edge_index"].astype(np.int64)),
                y=torch.from_numpy(topo["y"].astype(np.int64)),
                train_mask=torch.from_numpy(topo["train_mask0"]).bool(),
                val_mask=torch.from_numpy(topo["val_mask0"]).bool(),
                test_mask=torch.from_numpy(topo["test_mask0"]).bool())
    thr, beta0 = (float(v) for v in z["thr_beta"])
    torch.manual_seed(1234)
    model = sngnn_amd.SNGNN_Plus_Plus(f, 64, 5, n, 1, 1, thr, beta0, 0, 0.0)
    for k, v in model.state_dict().items():
        if "init." + k in z.files:
            assert np.array_equal(v.numpy(), z["init." + k]), k
    model = model.to(cuda)
    d = data.to(cuda)
    opt = torch.optim.Adam(model.parameters(), lr=0.1, weight_decay=5e-4)
    res = T.train(model, d, opt, epochs=8, patience=300)
    got = np.array([[h["train_loss"], h["val_loss"], h["val_acc"], h["test_loss"], h["test_acc"]]
                    for h in res["history"]])
    want = z["traj"]
    assert np.allclose(got[:, [0, 1, 3]], want[:, [0, 1, 3]], rtol=1e-4, atol=1e-5), (got, want)
    # accuracies: a handful of the 1 520 / 2 432 masked rows may sit on a decision boundary
    assert np.abs(got[:, [2, 4]] - want[:, [2, 4]]).max() <= 3.0 / 1520, (got[:, [2, 4]], want[:, [2, 4]])
    # final parameters (lr 0.1: Adam moves a parameter by up to 0.8 in 8 steps; 1e-3 of that)
    for k, v in model.state_dict().items():
        w = z["final." + k]
        assert np.abs(v.cpu().numpy() - w).max() <= 1e-3, k


@pytest.mark.parametrize("kind,args", [
    ("SNGNN_Plus", dict(hidden=64, layers=2, top_k=1, thr=0.0, rem=1, classes=40)),      # the training scripts' shape
    ("SNGNN_Plus", dict(hidden=32, layers=3, top_k=16, thr=0.2, rem=0, classes=7)),      # 32 -> 32: row-tile gx, mask left to the producer
    ("SNGNN", dict(hidden=24, layers=2, classes=5)),                                     # conv bias in the epilogue, no selection
    ("SNGNN_Plus_Plus", dict(hidden=32, layers=2, top_k=4, thr=0.0, rem=1, classes=6)),  # the blend takes the epilogue
    ("SNGNN_Plus_Plus", dict(hidden=20, layers=3, top_k=16, thr=0.1, rem=0, classes=9)),
])
def test_fused_hidden_epilogue_equals_the_op_sequence(cuda, kind, args):
    """models.py:204-209's relu_ + dropout between two conv layers as the aggregation's store
    epilogue / the next lin's input-gradient epilogue (ops.HiddenEpilogue) against the plain op
    sequence (models.FUSE_HIDDEN = False): same log-probs and same parameter gradients BIT FOR BIT
    without dropout (evaluation and training with p = 0); with dropout the fused forward is the op
    sequence under the mask it drew (checked through the gradients of a replay with that mask)."""
    import sngnn_amd
    from sngnn_amd import models as M
    from sngnn_amd import synth
    n, f = 6000, 48
    d = synth.make_dataset("actor", with_features=False)
    ei = d.edge_index[:, (d.edge_index[0] < n) & (d.edge_index[1] < n)].to(cuda)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, f, generator=gen).to(cuda)
    y = torch.randint(0, args["classes"], (n,), generator=gen).to(cuda)
    data = sngnn_amd.Data(x=x, edge_index=ei, y=y)

    def make():
        torch.manual_seed(11)
        if kind == "SNGNN":
            m = sngnn_amd.SNGNN(f, args["hidden"], args["classes"], args["layers"])
        elif kind == "SNGNN_Plus":
            m = sngnn_amd.SNGNN_Plus(f, args["hidden"], args["classes"], n, args["layers"], args["top_k"], args["thr"],
                                     args["rem"], 0.0)
        else:
            m = sngnn_amd.SNGNN_Plus_Plus(f, args["hidden"], args["classes"], n, args["layers"], args["top_k"],
                                          args["thr"], 0.3, args["rem"], 0.0)
        m = m.to(cuda)
        with torch.no_grad():
            for c in m.lins:
                if getattr(c, "bias", None) is not None:
                    c.bias.uniform_(-0.1, 0.1)          # a conv bias that matters
        return m

    def run(fuse, train, p):
        M.FUSE_HIDDEN = fuse
        try:
            m = make()
            m.dropout.p = p
            m.train(train)
            out = m(data)
            grads = None
            if train:
                torch.nn.functional.nll_loss(out, y).backward()
                grads = {k: v.grad.clone() for k, v in m.named_parameters()}
            return out.detach(), grads
        finally:
            M.FUSE_HIDDEN = True

    for train in (False, True):
        a, ga = run(True, train, 0.0)
        b, gb = run(False, train, 0.0)
        assert torch.equal(a, b), (kind, train)
        if train:
            for k in ga:
                assert torch.equal(ga[k], gb[k]), (kind, k, float((ga[k] - gb[k]).abs().max()))
    # with dropout: finite, and about half of the hidden activations dropped
    out, grads = run(True, True, 0.5)
    assert bool(torch.isfinite(out).all()) and all(bool(torch.isfinite(g).all()) for g in grads.values())


def test_hidden_epilogue_operator_with_dropout_matches_autograd_of_its_own_mask(cuda):
    """ops.aggregate(..., epilogue, bias) in training mode with p = 0.4: the stored rows are
    where(keep, relu(mean + bias) / (1 - p), 0) for SOME keep mask with about 60 % ones, and the
    gradients of h and bias are autograd's through that very expression (mask read back from the
    output)."""
    from sngnn_amd import ops
    from sngnn_amd.graph import Graph
    from tests.helpers import random_graph
    n, c, k, thr, p = 5000, 32, 4, 0.0, 0.4
    ei = random_graph(n, 40000, seed=9, hubs=((3, 700), (4, 200))).to(cuda)
    g = Graph(ei, n, True, True)
    gen = torch.Generator().manual_seed(2)
    h = torch.randn(n, c, generator=gen).to(cuda).requires_grad_(True)
    bias = (torch.randn(c, generator=gen) * 0.2).to(cuda).requires_grad_(True)
    gout = torch.randn(n, c, generator=gen).to(cuda)
    epi = ops.HiddenEpilogue(True, p, True)
    epi.applied = True
    x1 = ops.aggregate(h, g, k, thr, None, epi, bias)
    x1.backward(gout)
    gh, gb = h.grad.clone(), bias.grad.clone()
    h.grad = bias.grad = None
    y = ops.aggregate(h, g, k, thr) + bias                     # the plain path, differentiable
    pos = y.detach() > 0
    kept = x1.detach() != 0
    assert bool((kept <= pos).all())                            # nothing kept that relu removed
    frac = float(kept.sum()) / float(pos.sum())
    assert 0.57 < frac < 0.63, frac
    scale = 1.0 / (1.0 - p)
    want = torch.where(kept, torch.relu(y) * scale, torch.zeros_like(y))
    assert torch.equal(x1.detach(), want.detach())
    want.backward(gout)
    assert float((gh - h.grad).abs().max()) <= 1e-6 * float(h.grad.abs().max())
    assert float((gb - bias.grad).abs().max()) <= 1e-5 * float(bias.grad.abs().max())


@pytest.mark.parametrize("kind", ["SNGNN_Plus", "SNGNN"])
def test_graphed_epoch_with_fused_dropout_draws_a_new_mask_every_replay(cuda, kind):
    """train_graphed builds GraphedEpoch(warmup=0): the first TRAINING forward is the captured one.  The
    fused hidden epilogue's dropout seeds must then already exist on the device (created outside the
    capture, models._Stack.prepare_capture) and advance inside it: every replay drops other elements and
    keeps about 1 - p of the activations relu left."""
    import sngnn_amd
    from sngnn_amd import train as T
    from sngnn_amd import synth
    data = synth.make_dataset("cora", scale=0.5).to(cuda)
    n, f = data.x.shape
    torch.manual_seed(7)
    if kind == "SNGNN_Plus":
        model, p = sngnn_amd.SNGNN_Plus(f, 32, 7, n, 2, 3, 0.0, 1, 0.4).to(cuda), 0.4
    else:
        model, p = sngnn_amd.SNGNN(f, 32, 7, 2).to(cuda), 0.5           # (models.py:283: fixed at 0.5)
    assert getattr(model, "_drop_seed", None) is None
    hidden = torch.zeros(n, 32, device=cuda)

    def grab(mod, args, out):            # captured with the epoch: a copy of the TRAINING forward's activations
        if mod.training:
            hidden.copy_(out)
    model.lins[0].register_forward_hook(grab)
    opt = torch.optim.Adam(model.parameters(), lr=0.0, weight_decay=0.0, capturable=True)   # lr 0: same weights
    ge = T.GraphedEpoch(model, data, opt, warmup=0)
    assert model._drop_seed is not None and model._drop_seed.is_cuda
    seeds, masks = [], []
    for _ in range(3):
        ge.run()
        seeds.append(int(model._drop_seed[0]))
        masks.append((hidden != 0).clone())
    assert seeds[1] == seeds[0] + 1 and seeds[2] == seeds[1] + 1, seeds        # advanced INSIDE the graph
    assert not torch.equal(masks[0], masks[1]) and not torch.equal(masks[1], masks[2])
    # the same weights every replay (lr 0): an element relu kept is non-zero in a replay unless dropped
    alive = masks[0] | masks[1] | masks[2]
    for m in masks:
        frac = float(m.sum()) / float(alive.sum())
        assert abs(frac - (1.0 - p) / (1.0 - p ** 3)) < 0.03, frac
    # a fresh model without the preparation refuses to create its seeds inside a capture
    torch.manual_seed(7)
    m2 = sngnn_amd.SNGNN_Plus(f, 32, 7, n, 2, 3, 0.0, 1, 0.4).to(cuda).train()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.no_grad():
        m2.eval()
        m2(data)
        m2.train()
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="dropout seeds"):
        with torch.cuda.graph(g, stream=side):
            m2(data)
    torch.cuda.synchronize()


def test_forward_and_backward_enqueue_only(cuda):
    """no host synchronisation inside forward / backward (the reference has one per
    top-k round, models.py:257).  After warm-up (graph build, workspaces, lazily created seeds) the three
    models run forward + loss + backward under torch's sync debug mode "error" - every blocking call torch
    makes (``.item()``, ``nonzero``, blocking copies) raises - including the forward in which the fp16
    filter's data probe (conv._FilterHint) is due, and the one that takes its verdict."""
    import sngnn_amd
    from sngnn_amd import synth
    d = synth.make_dataset("actor", with_features=False)
    n, f, c = 7600, 48, 8
    ei = d.edge_index.to(cuda)
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, f, generator=gen).to(cuda)
    y = torch.randint(0, c, (n,), generator=gen).to(cuda)
    idx = torch.nonzero(torch.rand(n, generator=gen) < 0.6).flatten().to(cuda)      # (index form of the mask:
    yi = y[idx]                                                                     #  boolean indexing is the harness' sync)
    data = sngnn_amd.Data(x=x, edge_index=ei, y=y)
    torch.manual_seed(1)
    models = [sngnn_amd.SNGNN(f, 40, c, 2).to(cuda),
              sngnn_amd.SNGNN_Plus(f, 40, c, n, 2, 16, 0.9, 1, 0.5).to(cuda),          # thr 0.9, C 40: the filter + its probe
              sngnn_amd.SNGNN_Plus_Plus(f, 40, c, n, 2, 16, 0.3, 0.3, 1, 0.5).to(cuda),
              sngnn_amd.SNGNN_Plus(f, 32, c, n, 3, 2, 0.0, 0, 0.2, True).to(cuda)]     # batch norm: the unfused sequence

    def step(m):
        out = m(data)
        loss = F.nll_loss(out.index_select(0, idx), yi)
        m.zero_grad(set_to_none=True)
        loss.backward()
        return loss

    for m in models:
        m.train()
        for _ in range(3):
            step(m)
    torch.cuda.synchronize()
    hints = [l._filt_hint for m in models for l in m.lins if getattr(l, "_filt_hint", None) is not None]
    assert hints
    torch.cuda.set_sync_debug_mode("error")
    try:
        for m in models:
            for l in m.lins:
                h = getattr(l, "_filt_hint", None)
                if h is not None:
                    h._calls = h.EVERY                # the next forward's probe is due
            losses = [step(m) for _ in range(3)]
            m.eval()
            with torch.no_grad():
                m(data)
            m.train()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert all(bool(torch.isfinite(l)) for l in losses)
    # the probes were enqueued under the debug mode and their verdicts arrive without a wait
    assert any(h._flag is not None for h in hints)


def test_fused_hidden_epilogue_refuses_a_second_consumer(cuda):
    """ops.HiddenEpilogue's backward hand-over (the next lin's store pre-masks the gradient) is only right
    when the activated tensor has ONE consumer; a second one (here: the loss also reads the hidden
    activations) must raise, not return a silently wrong gradient - and the unfused sequence handles it."""
    import sngnn_amd
    from sngnn_amd import models as M
    from sngnn_amd import synth
    n, f = 6000, 48
    d = synth.make_dataset("actor", with_features=False)
    ei = d.edge_index[:, (d.edge_index[0] < n) & (d.edge_index[1] < n)].to(cuda)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, f, generator=gen).to(cuda)
    y = torch.randint(0, 6, (n,), generator=gen).to(cuda)
    data = sngnn_amd.Data(x=x, edge_index=ei, y=y)

    def loss_with_two_consumers(fuse):
        M.FUSE_HIDDEN = fuse
        try:
            torch.manual_seed(11)
            m = sngnn_amd.SNGNN_Plus(f, 32, 6, n, 2, 4, 0.0, 1, 0.0).to(cuda).train()
            seen = []
            m.lins[0].register_forward_hook(lambda mod, a, out: seen.append(out))
            out = m(data)
            hidden = seen[0]
            if not fuse:                      # (the hook saw the conv's raw output: apply the wrapper's relu)
                hidden = torch.relu(hidden)
            loss = F.nll_loss(out, y) + 0.1 * hidden.sum()
            loss.backward()
            return {k: v.grad.clone() for k, v in m.named_parameters()}
        finally:
            M.FUSE_HIDDEN = True

    with pytest.raises(RuntimeError, match="exactly one consumer"):
        loss_with_two_consumers(True)
    grads = loss_with_two_consumers(False)
    assert all(bool(torch.isfinite(g).all()) for g in grads.values())


@pytest.mark.parametrize("kind", ["SNGNN_Plus", "SNGNN"])
def test_eval_mode_batch_norm_folds_into_the_next_lin(cuda, kind):
    """models.py:207-208 in EVALUATION: BatchNorm1d on its running statistics is a per-channel scale and shift;
    the wrappers fold it into the next conv's ``lin`` (conv.LinFold) and put the conv's bias + relu into the
    aggregation's stores - no elementwise pass between two conv layers.  Equal to the op sequence
    (models.FUSE_HIDDEN = False) to fp32 rounding, with running statistics and affine parameters that matter;
    training mode keeps the op sequence (batch statistics) and is untouched."""
    import sngnn_amd
    from sngnn_amd import models as M
    from sngnn_amd import synth
    n, f = 6000, 48
    d = synth.make_dataset("actor", with_features=False)
    ei = d.edge_index[:, (d.edge_index[0] < n) & (d.edge_index[1] < n)].to(cuda)
    gen = torch.Generator().manual_seed(5)
    x = torch.randn(n, f, generator=gen).to(cuda)
    data = sngnn_amd.Data(x=x, edge_index=ei)
    torch.manual_seed(3)
    if kind == "SNGNN_Plus":
        m = sngnn_amd.SNGNN_Plus(f, 32, 7, n, 3, 4, 0.0, 1, 0.3, True).to(cuda)
    else:
        m = sngnn_amd.SNGNN(f, 24, 7, 3, True).to(cuda)
    with torch.no_grad():
        for bn in m.bns:
            bn.running_mean.uniform_(-0.3, 0.3)
            bn.running_var.uniform_(0.5, 2.0)
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.2, 0.2)
        for c in m.lins:
            c.bias.uniform_(-0.1, 0.1)
    m.eval()
    seen = {}
    from sngnn_amd import conv as CV
    orig = CV.LinFold.apply

    def spy(self, lin):
        seen["folds"] = seen.get("folds", 0) + 1
        return orig(self, lin)
    CV.LinFold.apply = spy
    try:
        with torch.no_grad():
            fused = m(data)
            M.FUSE_HIDDEN = False
            plain = m(data)
    finally:
        M.FUSE_HIDDEN = True
        CV.LinFold.apply = orig
    assert seen.get("folds", 0) == 2                                   # both hidden layers' norms were folded
    assert float((fused - plain).abs().max()) <= 2e-5 * max(1.0, float(plain.abs().max()))
    # the running statistics are untouched by an evaluation forward, and training still normalises by the batch
    m.train()
    rm = m.bns[0].running_mean.clone()
    out = m(data)
    assert not torch.equal(rm, m.bns[0].running_mean) and bool(torch.isfinite(out).all())
