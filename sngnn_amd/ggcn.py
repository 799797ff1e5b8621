"""GGCN's sparse layer on the gather skeleton: ``GGCNlayer_SP`` of
models/models.py:1453-1553 with the reference's constructor signature, parameter names
(``fcn``, ``deg_coeff``, ``coeff``, ``scale``), initial values and ``forward(h, adj,
degree_precompute)`` contract.

What the reference does per forward in the ``use_sign`` branch - two fancy-index gathers of
``Wh`` rows and ``F.cosine_similarity`` over all entries (get_sparse_att, :1512-1519), four
sparse tensors, four elementwise sparse products and two ``torch.sparse.mm`` (:1529-1537) -
is ONE gather here: ``sngnn_signed_forward`` reads every ``Wh_j`` once, forms the cosine and adds
the row into the output with the weight ``a_e (c_0 relu(s_e) - c_1 relu(-s_e))``; its autograd
(through the message values, both rows of every cosine, the degree scaling and the coefficients)
is ``sngnn_signed_backward`` + a few per-edge elementwise operations (ops._SignedPropagate).
The softmax of ``coeff``, the softplus of ``scale`` and the ``coeff[2] * Wh`` term are
elementwise on [N, C] / scalars and stay PyTorch.

The structure of ``adj`` (CSR by row without the diagonal, the CSC transpose for the backward,
the map from adj's entry order to CSR order) is built once per ``adj`` and cached on the layer,
where the reference caches ``adj_remove_diag``.  GPU tensors only (no CPU path).
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .graph import Graph


def precompute_degree_s(adj: torch.Tensor) -> torch.Tensor:
    """GGCN.precompute_degree_s (models.py:1691-1707) without the Python loop over the entries:
    value ``adj[i, i] / adj[i, j] - 1`` at every entry (i, j) of ``adj`` (coalesced sparse COO whose
    every row has its diagonal entry - the normalised adjacency with self-loops GGCN is fed)."""
    idx, val = adj._indices(), adj._values()
    diag = val[idx[0] == idx[1]]
    if diag.numel() != adj.size(0):
        raise ValueError("precompute_degree_s needs a diagonal entry in every row of adj")
    return torch.sparse_coo_tensor(idx, diag[idx[0]] / val - 1, adj.size())


class _AdjStructure:
    """Per-``adj`` cache: the device graph of the off-diagonal entries (source = column,
    target = row), and ``perm`` with ``coef_csr = coef_entries[perm]``."""

    def __init__(self, adj: torch.Tensor):
        if not adj.is_sparse or not adj.is_cuda:
            raise ValueError("adj must be a sparse COO tensor on the GPU (there is no CPU path)")
        if not adj.is_coalesced():
            raise ValueError("adj must be coalesced (the reference multiplies sparse tensors entry by entry)")
        idx = adj._indices()
        n = adj.size(0)
        if adj.size(1) != n:
            raise ValueError("adj must be square")
        self.key = (idx.data_ptr(), tuple(idx.shape), idx._version)
        self.idx = idx                                   # keeps the storage alive: the key cannot be recycled
        ei = torch.stack([idx[1], idx[0]]).contiguous()  # source = column j, target = row i
        self.graph = Graph(ei, n, False, True)           # remove_loops: adj_remove_diag (:1501-1506)
        off_diag = torch.nonzero(idx[0] != idx[1]).flatten()
        eid = torch.from_numpy(self.graph.array("eid").astype("int64")).to(idx.device)
        self.perm = off_diag[eid]
        self._ei, self._n = ei, n
        self._full = None

    def full(self):
        """The structure of ALL entries of adj, diagonal included (the plain propagation of use_sign=False,
        models.py:1544-1549): (graph, perm with ``coef_csr = coef_entries[perm]``, (csc_eid, tgt, src) of
        ``ops.weighted_propagate``).  Built on first use."""
        if self._full is None:
            g = Graph(self._ei, self._n, False, False)
            dev = self._ei.device
            perm = torch.from_numpy(g.array("eid").astype("int64")).to(dev)
            src = torch.from_numpy(g.array("col")).to(dev)                       # int32 [E'] source of each CSR entry
            rowptr = torch.from_numpy(g.array("rowptr").astype("int64")).to(dev)
            tgt = torch.repeat_interleave(torch.arange(self._n, device=dev, dtype=torch.int32), rowptr[1:] - rowptr[:-1])
            csc_eid = torch.from_numpy(g.array("csc_eid").astype("int64")).to(dev)
            self._full = (g, perm, (csc_eid, tgt.contiguous(), src.contiguous()))
        return self._full


class GGCNlayer_SP(nn.Module):
    """models.py:1453-1553."""

    def __init__(self, in_features, out_features, device=None, use_degree=True, use_sign=True, use_decay=True,
                 scale_init=0.5, deg_intercept_init=0.5):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.fcn = nn.Linear(in_features, out_features)
        self.use_degree, self.use_sign, self.use_decay = use_degree, use_sign, use_decay
        self.deg_intercept_init, self.scale_init = deg_intercept_init, scale_init
        self.device = device
        if use_degree:
            self.deg_coeff = nn.Parameter(torch.tensor([0.5 if use_decay else deg_intercept_init, 0.0]))
        if use_sign:
            self.coeff = nn.Parameter(torch.zeros(3))
            self.scale = nn.Parameter((2.0 if use_decay else scale_init) * torch.ones(1))
        self._structure = None

    def reset_parameters(self):
        """models.py:1483-1499 (fresh Parameters there; the same values in place here, so an
        optimizer built before the call keeps pointing at the live tensors)."""
        self.fcn.reset_parameters()
        with torch.no_grad():
            if self.use_degree:
                self.deg_coeff.copy_(torch.tensor([0.5 if self.use_decay else self.deg_intercept_init, 0.0]))
            if self.use_sign:
                self.coeff.zero_()
                self.scale.fill_(2.0 if self.use_decay else self.scale_init)

    def _adj(self, adj) -> _AdjStructure:
        idx = adj._indices()
        key = (idx.data_ptr(), tuple(idx.shape), idx._version)
        if self._structure is None or self._structure.key != key:
            self._structure = _AdjStructure(adj)
        return self._structure

    def forward(self, h, adj, degree_precompute):
        if not h.is_cuda:
            raise ValueError("h must live on the GPU (there is no CPU path)")
        val = adj._values()
        coef = val
        if self.use_degree:
            dv = degree_precompute._values()
            if dv.numel() != val.numel():
                raise ValueError("degree_precompute must have adj's entries (GGCN.precompute_degree_s)")
            coef = val * F.softplus(self.deg_coeff[0] * dv + self.deg_coeff[1])       # adj * sc (:1508-1510)
        wh = ops.linear(h, self.fcn)
        if not self.use_sign:
            # :1544-1549: a plain weighted sparse product (diagonal included), no cosine - the same gather-sum
            # kernels as SNGNN++'s adjacency branch, with one weight per entry (no torch.sparse.mm: fixed order,
            # no atomics, autograd through Wh and through the degree coefficients)
            graph, perm, aux = self._adj(adj).full()
            return ops.weighted_propagate(wh, coef[perm], graph, aux)
        st = self._adj(adj)
        c = F.softmax(self.coeff, dim=-1)
        scale = F.softplus(self.scale)
        prop = ops.signed_propagate(wh, coef[st.perm], c[:2], st.graph)       # c0 prop_pos + c1 prop_neg
        return scale * (prop + c[2] * wh)
