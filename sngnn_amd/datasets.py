"""Dataset ingestion without PyG: the geom-gcn text format and split files that the
reference's dataset classes read (datasets/datasets.py:147-304 - WebKB, Wikipedia
networks, Actor).  Pure host code (numpy), returns the minimal
``Data`` container; move it to the GPU with ``.to('cuda')``.

Layout of a raw directory (what the reference downloads into ``<root>/<name>/raw``):
    out1_node_feature_label.txt   "node_id <tab> features <tab> label"
    out1_graph_edges.txt          "src <tab> dst" per line
    <name>_split_0.6_0.2_<i>.npz  train_mask / val_mask / test_mask  (i = 0..9)
Actor ("film") lists the indices of its non-zero binary features
(datasets/datasets.py:263-275); the other datasets list dense feature values
(:209-213, :157-164).
"""
from __future__ import annotations

import glob
import os
import re
from typing import Optional

import numpy as np
import torch

from .synth import Data


def coalesce(edge_index: np.ndarray, num_nodes: int) -> np.ndarray:
    """Sort by (src, dst) and drop duplicates - what PyG's ``coalesce`` does for an
    edge list without attributes (datasets/datasets.py:170,221,284)."""
    key = edge_index[0].astype(np.int64) * num_nodes + edge_index[1].astype(np.int64)
    key = np.unique(key)
    return np.stack([key // num_nodes, key % num_nodes]).astype(np.int64)


def _read_lines(path: str):
    with open(path, "r") as f:
        return f.read().split("\n")[1:-1]          # header dropped, trailing newline dropped


def load_geom_gcn(raw_dir: str, name: Optional[str] = None, index_features: Optional[bool] = None,
                  num_features: Optional[int] = None) -> Data:
    """Parse one geom-gcn dataset.  ``index_features`` (Actor) is auto-detected from
    the header ("feature_amount") when not given.  Masks are stacked [n_splits, N]
    like the reference's classes do; pick one with :func:`select_split`."""
    feat_path = os.path.join(raw_dir, "out1_node_feature_label.txt")
    edge_path = os.path.join(raw_dir, "out1_graph_edges.txt")
    with open(feat_path, "r") as f:
        header = f.readline()
    m = re.search(r"feature_amount:(\d+)", header)
    if index_features is None:
        index_features = m is not None
    rows = [r.split("\t") for r in _read_lines(feat_path)]
    n = len(rows)
    y = np.empty(n, dtype=np.int64)
    if index_features:
        # Actor: the dense width is max index + 1 (SparseTensor(...).to_dense(), :270-271)
        ids, cols = [], []
        for nid, feats, label in rows:
            c = [int(v) for v in feats.split(",")] if feats else []
            ids += [int(nid)] * len(c)
            cols += c
            y[int(nid)] = int(label)
        width = num_features or (max(cols) + 1 if cols else 0)
        x = np.zeros((n, width), dtype=np.float32)
        x[np.asarray(ids, dtype=np.int64), np.asarray(cols, dtype=np.int64)] = 1.0
    else:
        # rows are used in FILE order as the reference does (:209-213)
        x = np.asarray([[float(v) for v in r[1].split(",")] for r in rows], dtype=np.float32)
        y = np.asarray([int(r[2]) for r in rows], dtype=np.int64)
    edges = np.asarray([[int(v) for v in r.split("\t")] for r in _read_lines(edge_path)],
                       dtype=np.int64).T
    edge_index = coalesce(edges, n)

    pattern = os.path.join(raw_dir, f"{name}_split_0.6_0.2_*.npz" if name else "*_split_0.6_0.2_*.npz")
    def split_id(p):
        return int(re.search(r"_(\d+)\.npz$", p).group(1))
    masks = {"train_mask": [], "val_mask": [], "test_mask": []}
    for p in sorted(glob.glob(pattern), key=split_id):
        z = np.load(p)
        for k in masks:
            masks[k].append(z[k].astype(bool))
    data = Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(edge_index), y=torch.from_numpy(y))
    for k, v in masks.items():
        setattr(data, k, torch.from_numpy(np.stack(v)) if v else None)
    return data


def select_split(data: Data, part_id: int) -> Data:
    """train.py:399-409: ``data.{train,val,test}_mask = mask[part_id]``."""
    kw = dict(data.__dict__)
    for k in ("train_mask", "val_mask", "test_mask"):
        if kw.get(k) is not None and kw[k].dim() == 2:
            kw[k] = kw[k][part_id]
    return Data(**kw)


def _read_csv_gz(path: str, dtype):
    import gzip
    with gzip.open(path, "rt") as f:
        return np.loadtxt(f, delimiter=",", dtype=dtype, ndmin=2)


def load_ogb_raw(root: str, split: Optional[str] = None) -> Data:
    """The on-disk layout ``ogb.nodeproppred.NodePropPredDataset`` unpacks - what
    ``load_ogb_dataset`` reads through that package (datasets/largescale_datasets.py:804-819;
    ogb is not vendored, the layout is its published one):
        <root>/raw/edge.csv.gz          "src,dst" per line (directed, as stored)
        <root>/raw/node-feat.csv.gz     one feature row per node
        <root>/raw/node-label.csv.gz    one label per node
        <root>/raw/num-node-list.csv.gz the node count
        <root>/split/<type>/{train,valid,test}.csv.gz   node indices
    The edge list is returned exactly as stored: the reference does not symmetrise it
    (SURVEY.md 8, config 4).  ``split`` names the split directory (default: the only one).
    """
    raw = os.path.join(root, "raw")
    ei = _read_csv_gz(os.path.join(raw, "edge.csv.gz"), np.int64).T
    x = _read_csv_gz(os.path.join(raw, "node-feat.csv.gz"), np.float32)
    y = _read_csv_gz(os.path.join(raw, "node-label.csv.gz"), np.int64).reshape(-1)
    nfile = os.path.join(raw, "num-node-list.csv.gz")
    n = int(_read_csv_gz(nfile, np.int64)[0, 0]) if os.path.exists(nfile) else x.shape[0]
    if x.shape[0] != n or y.shape[0] != n:
        raise ValueError(f"node-feat / node-label rows ({x.shape[0]}, {y.shape[0]}) != num nodes {n}")
    if ei.size and (ei.min() < 0 or ei.max() >= n):
        raise ValueError("edge.csv.gz names a node outside [0, num_nodes)")
    data = Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(np.ascontiguousarray(ei)),
                y=torch.from_numpy(y))
    sroot = os.path.join(root, "split")
    if os.path.isdir(sroot):
        kinds = sorted(os.listdir(sroot))
        kind = split if split is not None else (kinds[0] if len(kinds) == 1 else None)
        if kind is None:
            raise ValueError(f"several splits {kinds}: pass split=")
        for fname, attr in (("train", "train_mask"), ("valid", "val_mask"), ("test", "test_mask")):
            idx = _read_csv_gz(os.path.join(sroot, kind, fname + ".csv.gz"), np.int64).reshape(-1)
            mask = torch.zeros(n, dtype=torch.bool)
            mask[torch.from_numpy(idx)] = True
            setattr(data, attr, mask)
    return data
