"""CPU: host-side logic - module surface (names, signatures, state_dict keys, init
streams), synthetic generators, trainer loop semantics."""
import inspect

import numpy as np
import pytest
import torch

import sngnn_amd
from oracle import sngnn_oracle as O
from sngnn_amd import synth


def test_constructor_signatures_match_the_reference():
    # models.py:266, :162-163, :36-37 and :306, :215-216, :90-91
    sig = lambda f: list(inspect.signature(f).parameters)[1:]
    assert sig(sngnn_amd.SNGNN.__init__) == ["in_channels", "hidden_channels", "out_channels",
                                             "num_layers", "bn"]
    assert sig(sngnn_amd.SNGNN_Plus.__init__) == [
        "in_channels", "hidden_channels", "out_channels", "num_nodes", "num_layers", "top_k",
        "thr", "is_remove_self_loops", "droput_rate", "bn"]
    assert sig(sngnn_amd.SNGNN_Plus_Plus.__init__) == [
        "in_channels", "hidden_channels", "out_channels", "num_nodes", "num_layers", "top_k",
        "thr", "init_beta", "is_remove_self_loops", "droput_rate", "bn"]
    assert sig(sngnn_amd.SNConv.__init__) == ["in_channels", "out_channels", "aggr", "bias"]
    assert sig(sngnn_amd.SNConv_plus.__init__) == [
        "in_channels", "out_channels", "num_nodes", "top_k", "thr", "is_remove_self_loops",
        "bias", "aggr"]
    assert sig(sngnn_amd.SNConv_plus_plus.__init__) == [
        "in_channels", "out_channels", "num_nodes", "top_k", "thr", "init_beta",
        "is_remove_self_loops", "bias", "aggr"]
    d = inspect.signature(sngnn_amd.SNGNN_Plus.__init__).parameters
    assert (d["top_k"].default, d["thr"].default, d["is_remove_self_loops"].default,
            d["droput_rate"].default) == (2, 0.0, 1, 0.5)


@pytest.mark.parametrize("kind,args", [
    ("SNGNN", (10, 8, 3, 1)), ("SNGNN", (10, 8, 3, 3, True)),
    ("SNGNN_Plus", (10, 8, 3, 40, 2, 4, 0.1, 1, 0.5, True)),
    ("SNGNN_Plus", (10, 8, 3, 40, 1)),
    ("SNGNN_Plus_Plus", (10, 8, 3, 40, 2, 4, 0.1, 0.3, 0, 0.5, True)),
    ("SNGNN_Plus_Plus", (10, 8, 3, 40, 1, 2, 0.0, 0.0)),
])
def test_state_dict_keys_shapes_and_seeded_init_equal_the_restated_reference(kind, args):
    torch.manual_seed(7)
    ours = getattr(sngnn_amd, kind)(*args)
    torch.manual_seed(7)
    ref = getattr(O, kind)(*args)
    a, b = ours.state_dict(), ref.state_dict()
    assert list(a) == list(b)
    for k in a:
        assert a[k].shape == b[k].shape and torch.equal(a[k], b[k]), k
    # bn flag lands in the conv's bias slot (models.py:52-53,177-178)
    has_bias = any(k.endswith("lins.0.bias") for k in a)
    assert has_bias == (kind == "SNGNN" or bool(ours.bn))


def test_adjacency_table_layout():
    m = sngnn_amd.SNConv_plus_plus(6, 4, 30)
    assert m.w.weight.shape == (4, 30) and m.w.weight.t().is_contiguous()
    assert float(m.beta) == 0.5
    m2 = sngnn_amd.SNConv_plus_plus(6, 4, 30, init_beta=0.0)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2.load_state_dict(sd)
    assert torch.equal(m2.w.weight, m.w.weight) and m2.w.weight.t().is_contiguous()


@pytest.mark.parametrize("name", ["cora", "chameleon", "actor"])
def test_synthetic_datasets_have_the_published_shapes(name):
    n, e, f, c, max_deg, kind, dens = synth.SHAPES[name]
    d = synth.make_dataset(name)
    ei = d.edge_index.numpy()
    assert d.x.shape == (n, f) and ei.shape == (2, e)
    key = ei[0].astype(np.int64) * (n + 1) + ei[1]
    assert (np.diff(key) > 0).all(), "sorted by (src, dst) and de-duplicated (coalesce)"
    deg = np.bincount(ei[1], minlength=n)
    assert deg.max() <= max_deg and (deg == 0).mean() >= 0.01
    assert (ei[0] == ei[1]).sum() > 0
    assert int(d.train_mask.sum() + d.val_mask.sum() + d.test_mask.sum()) == n
    d2 = synth.make_dataset(name)
    assert torch.equal(d.edge_index, d2.edge_index) and torch.equal(d.x, d2.x)


def test_partition_edge_generation():
    rng = np.random.default_rng(0)
    ei = synth.make_edges(rng, 100, 600, 50, n_src=400, dst_offset=200)
    assert ei[1].min() >= 200 and ei[1].max() < 300 and ei[0].max() < 400


def test_trainer_loop_early_stopping_semantics():
    """train.py:150-158: strict '<' on the validation loss, patience counter, test
    accuracy taken at the best validation loss."""
    from sngnn_amd import train as T

    class Fake(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))
            self.calls = 0

        def forward(self, data):
            self.calls += 1
            return torch.log_softmax(self.p * 0 + data.x, dim=1)

    n = 12
    data = synth.Data(x=torch.randn(n, 3), edge_index=torch.zeros(2, 0, dtype=torch.long),
                      y=torch.randint(0, 3, (n,)), train_mask=torch.arange(n) < 6,
                      val_mask=(torch.arange(n) >= 6) & (torch.arange(n) < 9),
                      test_mask=torch.arange(n) >= 9)
    model = Fake()
    opt = torch.optim.Adam(model.parameters(), lr=0.1)
    res = T.train(model, data, opt, epochs=50, patience=3)
    # the output never changes -> val loss never improves after epoch 0 -> stop at epoch 3
    assert len(res["history"]) == 4 and model.calls == 12
    assert res["final_test_acc"] == res["history"][0]["test_acc"]


def _write_geom(tmp, index_features):
    import numpy as np
    raw = tmp / "raw"
    raw.mkdir()
    if index_features:
        lines = ["node_id\tfeature(feature_amount:5)\tlabel", "2\t0,4\t1", "0\t1\t0", "1\t2,3,5\t2", "3\t\t1"]
    else:
        lines = ["node_id\tfeature\tlabel", "0\t0.5,1.0,0\t0", "1\t0,0,2.5\t2", "2\t1,1,1\t1", "3\t0,3,0\t1"]
    (raw / "out1_node_feature_label.txt").write_text("\n".join(lines) + "\n")
    (raw / "out1_graph_edges.txt").write_text("node_id\tnode_id\n2\t1\n0\t1\n2\t1\n3\t3\n1\t0\n")
    for i in (1, 0):
        np.savez(raw / f"toy_split_0.6_0.2_{i}.npz", train_mask=np.array([1, 0, i, 0], np.uint8),
                 val_mask=np.array([0, 1, 0, 0], np.uint8), test_mask=np.array([0, 0, 1 - i, 1], np.uint8))
    return str(raw)


@pytest.mark.parametrize("index_features", [False, True])
def test_geom_gcn_parser(tmp_path, index_features):
    """datasets/datasets.py:157-190, 208-250 (dense feature rows) and :263-304 (Actor's
    feature-index lists): coalesced edges, stacked split masks in split order."""
    from sngnn_amd import datasets as DS
    d = DS.load_geom_gcn(_write_geom(tmp_path, index_features), "toy")
    assert d.edge_index.tolist() == [[0, 1, 2, 3], [1, 0, 1, 3]]          # sorted, de-duplicated
    assert d.train_mask.shape == (2, 4) and d.train_mask[1].tolist() == [True, False, True, False]
    if index_features:
        assert d.x.shape == (4, 6) and d.x[1].tolist() == [0, 0, 1, 1, 0, 1] and d.x[3].sum() == 0
        assert d.y.tolist() == [0, 2, 1, 1]                               # placed by node id
    else:
        assert d.x.shape == (4, 3) and d.x[1].tolist() == [0, 0, 2.5] and d.y.tolist() == [0, 2, 1, 1]
    s = DS.select_split(d, 0)
    assert s.train_mask.tolist() == [True, False, False, False] and s.x is d.x


def test_actor_fixture_statistics():
    """tests/golden/actor_topology.npz was parsed from the raw files the reference
    bundles (datasets/data/Actor/raw); its statistics are the ones SURVEY.md row 13
    measured independently."""
    import os
    import numpy as np
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "actor_topology.npz"))
    n, e, loops, max_deg, zero_deg, f, classes = (int(v) for v in z["stats"])
    assert (n, e, loops, max_deg, zero_deg, f, classes) == (7600, 30019, 93, 1296, 2236, 932, 5)
    ei = z["edge_index"].astype(np.int64)
    assert ei.shape == (2, e) and (np.diff(ei[0] * n + ei[1]) > 0).all()
    assert (int(z["train_mask0"].sum()), int(z["val_mask0"].sum()), int(z["test_mask0"].sum())) == \
        (3648, 2432, 1520)
    raw = "/root/reference/datasets/data/Actor/raw"
    if os.path.isdir(raw):                                  # where the reference is mounted: re-parse
        from sngnn_amd import datasets as DS
        d = DS.load_geom_gcn(raw, "film")
        assert np.array_equal(d.edge_index.numpy(), ei) and d.x.shape == (n, f)
        assert np.array_equal(d.y.numpy(), z["y"].astype(np.int64))


def test_ogb_raw_layout_loader(tmp_path):
    """largescale_datasets.py:804-819 without the ogb package: the published raw layout."""
    import gzip
    from sngnn_amd import datasets as DS
    rng = np.random.default_rng(0)
    n, e, f = 50, 200, 6
    ei = rng.integers(0, n, size=(e, 2))
    x = rng.normal(size=(n, f)).astype(np.float32)
    y = rng.integers(0, 4, size=n)
    perm = rng.permutation(n)
    root = tmp_path / "ogbn_toy"
    (root / "raw").mkdir(parents=True)
    (root / "split" / "time").mkdir(parents=True)

    def dump(path, arr, fmt):
        with gzip.open(path, "wt") as fh:
            np.savetxt(fh, arr, delimiter=",", fmt=fmt)
    dump(root / "raw" / "edge.csv.gz", ei, "%d")
    dump(root / "raw" / "node-feat.csv.gz", x, "%.9g")
    dump(root / "raw" / "node-label.csv.gz", y.reshape(-1, 1), "%d")
    dump(root / "raw" / "num-node-list.csv.gz", np.array([[n]]), "%d")
    for name, idx in (("train", perm[:30]), ("valid", perm[30:40]), ("test", perm[40:])):
        dump(root / "split" / "time" / f"{name}.csv.gz", idx.reshape(-1, 1), "%d")
    d = DS.load_ogb_raw(str(root))
    assert d.edge_index.dtype == torch.int64 and d.edge_index.shape == (2, e)
    assert np.array_equal(d.edge_index.numpy(), ei.T)              # stored order, not symmetrised
    assert np.array_equal(d.x.numpy(), x) and np.array_equal(d.y.numpy(), y)
    assert int(d.train_mask.sum()) == 30 and int(d.val_mask.sum()) == 10 and int(d.test_mask.sum()) == 10
    assert not (d.train_mask & d.val_mask).any() and not (d.train_mask & d.test_mask).any()
    assert set(torch.nonzero(d.test_mask).flatten().tolist()) == set(perm[40:].tolist())
    bad = ei.copy()
    bad[0, 0] = n
    dump(root / "raw" / "edge.csv.gz", bad, "%d")
    with pytest.raises(ValueError):
        DS.load_ogb_raw(str(root))


def test_cosine_tile_prefetch_addresses_stay_inside_the_row():
    """Index arithmetic of k_cosine_mfma's prefetch (csrc/toolbox.hip, SN_FETCH) replayed on the
    host with the launcher's split rule: for every lane vector kq and every K-step of every
    split, the 16-byte read lies inside [0, ld) of its row - also when the step is out of the
    split's range (its value is discarded, the read is real).  ADVICE r2 (medium)."""
    TB_K = 32
    for n, f in [(40, 4), (130, 16), (300, 28), (1500, 200), (1664, 200), (2277, 2325), (100, 33), (513, 96)]:
        ld = (f + 3) // 4 * 4
        nb = (n + 127) // 128
        tiles = nb * (nb + 1) // 2
        ks = 1
        if tiles <= 96:
            ks = min(min(8, (512 + tiles - 1) // tiles), max(1, ld // (2 * TB_K)))
        k_per = ((ld + ks - 1) // ks + TB_K - 1) // TB_K * TB_K
        ks = (ld + k_per - 1) // k_per
        for split in range(ks):
            k_begin = split * k_per
            k_end = min(ld, k_begin + k_per)
            assert k_begin < ld
            k0 = k_begin
            while k0 < k_end:
                for kq in range(8):
                    kk = k0 if 4 * kq + k0 < k_end else k_begin - 4 * kq
                    first = 4 * kq + kk                      # offset of the vector inside the row
                    assert 0 <= first and first + 4 <= ld, (n, f, split, k0, kq, first, ld)
                k0 += TB_K
