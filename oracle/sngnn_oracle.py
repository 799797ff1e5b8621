"""CPU oracle for SNGNN's similarity-navigated aggregation path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``sngnn_amd/`` may import this module;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and there only as the checker / the timed CPU baseline.

PARITY UNPINNED at the third-party boundary: the arithmetic of the reference's
hot path lives in packages that are not vendored in /root/reference and are not
installed here (torch-geometric 2.0.4, torch-scatter 2.0.9, torch-sparse 0.6.13,
requirements.txt:67-69); the reference ships no tests, golden vectors or
fixtures for this path, and the reference modules
cannot be imported (ordinary ModuleNotFoundError).  What follows is therefore a
restatement in core torch of
  * the reference's own code, line by line (models/models.py:89-158, 214-263,
    305-334, wrappers :35-86, :161-211, :265-303; SimGFAToolbox/dense.py,
    sparse.py), using the very same core-torch expressions wherever the
    reference uses core torch (F.normalize, index_select, (a*b).sum(-1),
    torch.where, Tensor.scatter, nn.Linear on a sparse COO tensor), and
  * the published algorithms of the third-party calls at the reference's call
    sites, each in its own function below.
It is pinned by hand-derived known-answer tests (SURVEY.md Appendix B,
tests/golden/kat_appendix_b.json), by a literal serial-loop C restatement of
the same algorithms (oracle/sngnn_oracle.c), by cross-checks between the
vectorised and literal-loop forms, and - tests/golden/pin_reference.py, build
container only - by runs of the reference's OWN files: models/models.py under stub
modules whose third-party functions are this file's restatements (the in-tree half
of the conv path, bit for bit), and SimGFAToolbox/dense.py + sparse.py with the real
scipy / scikit-learn (the toolbox section at the end of this file; fixtures
tests/golden/toolbox_*.npz).  The third-party kernels themselves stay unpinned.

Everything here runs on CPU tensors, fp32 values and int64 indices exactly as
the reference's CPU path does.
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor
from torch.nn.parameter import Parameter


# --------------------------------------------------------------------------
# Third-party call sites, restated
# --------------------------------------------------------------------------

def add_self_loops(edge_index: Tensor, num_nodes: int) -> Tensor:
    """PyG 2.0.4 ``add_self_loops`` as called at models.py:117,234,323.

    Appends (v, v) for every v in [0, num_nodes) at the END of the edge list;
    existing loops are kept, nothing is sorted (Appendix A-1).
    """
    loop = torch.arange(0, num_nodes, dtype=torch.long, device=edge_index.device)
    loop = loop.unsqueeze(0).repeat(2, 1)
    return torch.cat([edge_index, loop], dim=1)


def remove_self_loops(edge_index: Tensor) -> Tensor:
    """PyG 2.0.4 ``remove_self_loops(edge_index, None)`` (models.py:120,236):
    ``edge_index[:, row != col]``, order preserved (Appendix A-2)."""
    mask = edge_index[0] != edge_index[1]
    return edge_index[:, mask]


def scatter_max_loop(src: Tensor, index: Tensor) -> Tuple[Tensor, Tensor]:
    """torch-scatter 2.0.9 CPU ``scatter_max(src, index, dim=0)`` - literal form.

    Appendix A-5: output length ``index.max()+1``; ``out`` starts at lowest(),
    ``arg`` at ``src.size(0)``; one serial loop with a strict ``>`` so the first
    (lowest-position) edge wins ties; untouched outputs are then set to 0 and
    keep ``arg == src.size(0)`` (what models.py:150,255 tests for).
    Pure-Python loop: small cases only.
    """
    E = src.numel()
    M = int(index.max()) + 1 if E > 0 else 0
    lowest = torch.finfo(torch.float32).min
    out = [lowest] * M
    arg = [E] * M
    s = src.tolist()
    idx = index.tolist()
    for e in range(E):
        i = idx[e]
        if s[e] > out[i]:
            out[i] = s[e]
            arg[i] = e
    out = [0.0 if a == E else o for o, a in zip(out, arg)]
    return (torch.tensor(out, dtype=src.dtype), torch.tensor(arg, dtype=torch.long))


def scatter_max(src: Tensor, index: Tensor) -> Tuple[Tensor, Tensor]:
    """Vectorised equivalent of :func:`scatter_max_loop` (same outputs bit for bit
    for NaN-free input): per-group maximum, then the LOWEST edge position among
    the edges equal to it ("strict > in a forward serial loop" == first
    occurrence of the maximum)."""
    E = src.numel()
    if E == 0:
        return src.new_zeros(0), index.new_zeros(0)
    M = int(index.max()) + 1
    lowest = torch.finfo(src.dtype).min
    out = torch.full((M,), lowest, dtype=src.dtype)
    out = out.scatter_reduce(0, index, src, reduce="amax", include_self=True)
    pos = torch.arange(E, dtype=torch.long)
    # an edge can only win with a value strictly above lowest()
    is_max = (src == out[index]) & (src > lowest)
    cand = torch.where(is_max, pos, torch.full_like(pos, E))
    arg = torch.full((M,), E, dtype=torch.long)
    arg = arg.scatter_reduce(0, index, cand, reduce="amin", include_self=True)
    out = torch.where(arg == E, torch.zeros_like(out), out)
    return out, arg


def scatter_mean(msg: Tensor, index: Tensor, dim_size: int) -> Tensor:
    """torch-scatter 2.0.9 ``scatter(msg, index, dim=-2, dim_size=N, reduce='mean')``
    as used by PyG's ``aggr='mean'`` (models.py:92,217,307; Appendix A-4):
    scatter_add_ in edge order, count of ALL edges per target, clamp count to
    >= 1, true divide."""
    out = torch.zeros((dim_size,) + tuple(msg.shape[1:]), dtype=msg.dtype)
    out.index_add_(0, index, msg)
    count = torch.zeros(dim_size, dtype=msg.dtype)
    count.index_add_(0, index, torch.ones(index.numel(), dtype=msg.dtype))
    count = count.clamp_(min=1)
    if msg.dim() > 1:
        count = count.view(-1, *([1] * (msg.dim() - 1)))
    return out / count


def scatter_mean_1d(src: Tensor, index: Tensor) -> Tensor:
    """``torch_scatter.scatter_mean(src, index, dim=0)`` with dim_size=None
    (SimGFAToolbox/dense.py:163): output length ``index.max()+1``."""
    M = int(index.max()) + 1 if index.numel() else 0
    return scatter_mean(src, index, M)


def sparse_adj_coo(row: Tensor, col: Tensor, num_nodes: int) -> Tensor:
    """``SparseTensor(row, col, sparse_sizes=(N, N)).to_torch_sparse_coo_tensor()``
    (models.py:126-127; Appendix A-6): entries sorted by (row, col), duplicates
    kept, value 1.0 fp32 each."""
    key = row * num_nodes + col
    perm = torch.argsort(key, stable=True)
    idx = torch.stack([row[perm], col[perm]], dim=0)
    val = torch.ones(idx.size(1), dtype=torch.float32)
    return torch.sparse_coo_tensor(idx, val, (num_nodes, num_nodes))


# --------------------------------------------------------------------------
# The reference's own message functions, restated
# --------------------------------------------------------------------------

def edge_cosine(norm: Tensor, edge_index: Tensor) -> Tensor:
    """models.py:140,245,332: ``(norm_i * norm_j).sum(dim=-1)`` with
    ``norm_j = norm.index_select(0, edge_index[0])`` (source) and
    ``norm_i = norm.index_select(0, edge_index[1])`` (target) - Appendix A-3."""
    norm_j = norm.index_select(0, edge_index[0])
    norm_i = norm.index_select(0, edge_index[1])
    return (norm_i * norm_j).sum(dim=-1)


def topk_threshold_weights(s: Tensor, index: Tensor, top_k: int, thr: float,
                           use_loop: bool = False, smax=None) -> Tuple[Tensor, List[Tensor]]:
    """models.py:141-156 / 246-261: ``top_k`` rounds of scatter_max, the -2 mask
    for empty groups, the fp32 ``>= thr`` compare, the -1.1 knock-out, then
    ``weight.scatter(-1, idx, s[idx])`` per round.

    Returns (weight [E'], list of the per-round selected edge positions).
    ``smax``: another ``scatter_max(src, index) -> (out, arg)`` to run the rounds with - the real
    ``torch_scatter.scatter_max`` where that package imports (tests/third_party_cases.py).
    """
    if smax is None:
        smax = scatter_max_loop if use_loop else scatter_max
    tmp_weight = s.clone()
    weight = torch.zeros(s.shape, dtype=s.dtype)
    max_indexes: List[Tensor] = []
    for _ in range(top_k):
        max_weight, max_index = smax(tmp_weight, index)
        new_max_weight = torch.where(max_index == tmp_weight.shape[0],
                                     torch.full_like(max_weight, -2), max_weight)
        new_max_index = max_index[torch.where(new_max_weight >= thr)[0]]
        tmp_weight = tmp_weight.scatter(-1, new_max_index, -1.1)
        max_indexes.append(new_max_index)
    for i in range(top_k):
        weight = weight.scatter(-1, max_indexes[i], s[max_indexes[i]])
    return weight, max_indexes


def selected_sources(max_indexes: List[Tensor], edge_index: Tensor, num_nodes: int,
                     top_k: int) -> Tuple[Tensor, Tensor]:
    """Per-row selected source ids in rank order, -1 padded: [N, top_k] int64, and
    the selected edge positions (in the E' edge list) likewise.  Re-selections of
    an already knocked-out edge (possible only when thr <= -1.1) are ignored."""
    sel_src = torch.full((num_nodes, top_k), -1, dtype=torch.long)
    sel_pos = torch.full((num_nodes, top_k), -1, dtype=torch.long)
    fill = torch.zeros(num_nodes, dtype=torch.long)
    seen = torch.zeros(edge_index.size(1), dtype=torch.bool)
    for idx in max_indexes:
        idx = idx[~seen[idx]]
        seen[idx] = True
        tgt = edge_index[1, idx]
        sel_src[tgt, fill[tgt]] = edge_index[0, idx]
        sel_pos[tgt, fill[tgt]] = idx
        fill[tgt] += 1
    return sel_src, sel_pos


def sn_edge_list(edge_index: Tensor, num_nodes: int, add_loops: bool,
                 remove_loops: bool) -> Tensor:
    """Self-loop handling of the three convs: models.py:323 (SNConv: add only),
    :234-236 and :117-120 (add, then optionally remove ALL loops)."""
    ei = edge_index
    if add_loops:
        ei = add_self_loops(ei, num_nodes)
    if remove_loops:
        ei = remove_self_loops(ei)
    return ei


def propagate_mean(x: Tensor, norm: Tensor, ei: Tensor, top_k: Optional[int],
                   thr: float, use_loop: bool = False, smax=None, smean=None):
    """PyG ``propagate`` + ``message`` + mean ``aggregate`` for the three convs
    (models.py:132+139-158, :239+244-263, :326+331-334; Appendix A-3/A-4).

    ``top_k=None`` is SNConv (every edge weighted by its cosine).  Returns
    (out [N,C], s [E'], weight [E'], per-round index list or None).
    ``smax`` / ``smean``: the REAL ``torch_scatter.scatter_max(src, index)`` /
    ``scatter(msg, index, dim_size)`` where that package imports (bench.py's literal-reference
    CPU leg, tests/third_party_cases.py) instead of the restatements above.
    """
    N = x.size(0)
    s = edge_cosine(norm, ei)
    if top_k is None:
        weight, rounds = s, None
    else:
        weight, rounds = topk_threshold_weights(s, ei[1], top_k, thr, use_loop, smax)
    x_j = x.index_select(0, ei[0])
    msg = weight.view(-1, 1) * x_j
    out = scatter_mean(msg, ei[1], N) if smean is None else smean(msg, ei[1], N)
    return out, s, weight, rounds


def segment_softmax(src: Tensor, index: Tensor, num_nodes: int) -> Tensor:
    """torch_geometric.utils.softmax (2.0.4) as called at models.py:404 with
    ``ptr=None``: per target group subtract the group maximum, exponentiate, divide
    by the group sum + 1e-16.  (Published algorithm; PyG is not vendored.)"""
    src_max = src.new_full((num_nodes,), float("-inf")).scatter_reduce(
        0, index, src, reduce="amax", include_self=True)
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = src.new_zeros(num_nodes).scatter_add(0, index, out) + 1e-16
    return out / out_sum.index_select(0, index)


def agnn_edge_list(edge_index: Tensor, num_nodes: int) -> Tensor:
    """models.py:393-395: original self-loops removed FIRST, then one loop per node
    appended (the opposite order of the SNConv layers)."""
    return add_self_loops(remove_self_loops(edge_index), num_nodes)


def propagate_attention(x: Tensor, norm: Tensor, ei: Tensor):
    """PyG ``propagate`` + ``message`` (models.py:400-405) + ``aggr='add'`` (:381):
    alpha = softmax over each target's in-edges of the cosine, out_i = sum alpha * x_j."""
    N = x.size(0)
    s = edge_cosine(norm, ei)
    alpha = segment_softmax(s, ei[1], N)
    msg = x.index_select(0, ei[0]) * alpha.view(-1, 1)
    out = x.new_zeros(N, x.size(1)).index_add_(0, ei[1], msg)
    return out, s, alpha


def attention_reference(h: Tensor, edge_index: Tensor):
    """Operator-level contract of the attention mode (everything after ``lin`` in
    models.py:392-399): dict(out [N,C], ei [2,E'], s [E'], alpha [E'])."""
    N = h.size(0)
    ei = agnn_edge_list(edge_index, N)
    norm = F.normalize(h, p=2., dim=-1)
    out, s, alpha = propagate_attention(h, norm, ei)
    return dict(out=out, ei=ei, s=s, alpha=alpha)


# --------------------------------------------------------------------------
# Conv layers (same constructor signatures, parameter names and init order as
# the reference, so seeded construction gives identical parameters)
# --------------------------------------------------------------------------

class SNConv(nn.Module):
    """models.py:305-334."""

    def __init__(self, in_channels, out_channels, aggr='mean', bias: bool = True):
        super().__init__()
        assert aggr == 'mean'
        self.lin = nn.Linear(in_channels, out_channels)
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()
        if self.bias is not None:          # PyG inits.zeros: no-op on None
            self.bias.data.fill_(0)

    def forward(self, x, edge_index):
        ei = sn_edge_list(edge_index, x.size(0), True, False)
        x = self.lin(x)
        norm = F.normalize(x, p=2., dim=-1)
        out, *_ = propagate_mean(x, norm, ei, None, 0.0)
        if self.bias is not None:
            out = out + self.bias
        return out


class SNConv_plus(nn.Module):
    """models.py:214-263."""

    def __init__(self, in_channels, out_channels, num_nodes, top_k=2, thr=0.0,
                 is_remove_self_loops=True, bias: bool = False, aggr='mean'):
        super().__init__()
        assert aggr == 'mean'
        self.top_k = top_k
        self.thr = thr
        self.num_nodes = num_nodes
        self.is_remove_self_loops = is_remove_self_loops
        self.lin = nn.Linear(in_channels, out_channels)
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()
        if self.bias is not None:
            self.bias.data.fill_(0)

    def forward(self, x, edge_index):
        ei = sn_edge_list(edge_index, x.size(0), True, bool(self.is_remove_self_loops))
        x = self.lin(x)
        norm = F.normalize(x, p=2., dim=-1)
        out, *_ = propagate_mean(x, norm, ei, self.top_k, self.thr)
        if self.bias is not None:
            out = out + self.bias
        return out


class SNConv_plus_plus(nn.Module):
    """models.py:89-158."""

    def __init__(self, in_channels, out_channels, num_nodes, top_k=2, thr=0.0,
                 init_beta=0.5, is_remove_self_loops=True, bias: bool = False,
                 aggr='mean'):
        super().__init__()
        assert aggr == 'mean'
        self.top_k = top_k
        self.thr = thr
        self.w = nn.Linear(num_nodes, out_channels)
        self.num_nodes = num_nodes
        self.is_remove_self_loops = is_remove_self_loops
        self.lin = nn.Linear(in_channels, out_channels)
        self.beta = Parameter(torch.empty(1))
        self.init_beta = init_beta
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()
        if self.bias is not None:
            self.bias.data.fill_(0)
        self.w.reset_parameters()
        self.beta.data.fill_(self.init_beta)

    def forward(self, x, edge_index):
        ei = sn_edge_list(edge_index, x.size(0), True, bool(self.is_remove_self_loops))
        x = self.lin(x)
        norm = F.normalize(x, p=2., dim=-1)
        row, col = ei
        row = row - row.min()                       # models.py:125
        adj = sparse_adj_coo(row, col, self.num_nodes)
        out_0 = self.w(adj)                          # models.py:130
        out_1, *_ = propagate_mean(x, norm, ei, self.top_k, self.thr)
        out = self.beta * out_0 + (1 - self.beta) * out_1
        if self.bias is not None:
            out = out + self.bias
        return out


# --------------------------------------------------------------------------
# Model wrappers (models.py:35-86, 161-211, 265-303)
# --------------------------------------------------------------------------

class _Stack(nn.Module):
    def _run(self, data):
        x, edge_index = data.x, data.edge_index
        for i, lin in enumerate(self.lins[:-1]):
            x = lin(x, edge_index)
            x = F.relu(x, inplace=True)
            if self.bn:
                x = self.bns[i](x)
            x = self.dropout(x)
        x = self.lins[-1](x, edge_index)
        return F.log_softmax(x, dim=1)

    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        if self.bn:
            for bn in self.bns:
                bn.reset_parameters()

    def forward(self, data):
        return self._run(data)


class SNGNN(_Stack):
    """models.py:265-303."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, bn=False):
        super().__init__()
        self.bn = bn
        self.lins = nn.ModuleList()
        if self.bn:
            self.bns = nn.ModuleList()
        if num_layers == 1:
            self.lins.append(SNConv(in_channels, out_channels))
        else:
            self.lins.append(SNConv(in_channels, hidden_channels))
            if self.bn:
                self.bns.append(nn.BatchNorm1d(hidden_channels))
            for _ in range(num_layers - 2):
                self.lins.append(SNConv(hidden_channels, hidden_channels))
                if self.bn:
                    self.bns.append(nn.BatchNorm1d(hidden_channels))
            self.lins.append(SNConv(hidden_channels, out_channels))
        self.dropout = nn.Dropout(p=0.5)
        self.reset_parameters()


class SNGNN_Plus(_Stack):
    """models.py:161-211 (note: ``bn`` lands in the conv's ``bias`` slot, :177-178)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_nodes, num_layers,
                 top_k=2, thr=0.0, is_remove_self_loops=1, droput_rate=0.5, bn=False):
        super().__init__()
        self.top_k, self.thr, self.bn, self.num_nodes = top_k, thr, bn, num_nodes
        self.is_remove_self_loops = (is_remove_self_loops == 1)
        self.lins = nn.ModuleList()
        if self.bn:
            self.bns = nn.ModuleList()

        def conv(i, o):
            return SNConv_plus(i, o, self.num_nodes, self.top_k, self.thr,
                               self.is_remove_self_loops, self.bn)
        if num_layers == 1:
            self.lins.append(conv(in_channels, out_channels))
        else:
            self.lins.append(conv(in_channels, hidden_channels))
            if self.bn:
                self.bns.append(nn.BatchNorm1d(hidden_channels))
            for _ in range(num_layers - 2):
                self.lins.append(conv(hidden_channels, hidden_channels))
                if self.bn:
                    self.bns.append(nn.BatchNorm1d(hidden_channels))
            self.lins.append(conv(hidden_channels, out_channels))
        self.dropout = nn.Dropout(p=droput_rate)
        self.reset_parameters()


class SNGNN_Plus_Plus(_Stack):
    """models.py:35-86."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_nodes, num_layers,
                 top_k=2, thr=0.0, init_beta=0.5, is_remove_self_loops=1,
                 droput_rate=0.5, bn=False):
        super().__init__()
        self.top_k, self.thr, self.bn = top_k, thr, bn
        self.init_beta, self.num_nodes = init_beta, num_nodes
        self.is_remove_self_loops = (is_remove_self_loops == 1)
        self.lins = nn.ModuleList()
        if self.bn:
            self.bns = nn.ModuleList()

        def conv(i, o):
            return SNConv_plus_plus(i, o, self.num_nodes, self.top_k, self.thr,
                                    self.init_beta, self.is_remove_self_loops, self.bn)
        if num_layers == 1:
            self.lins.append(conv(in_channels, out_channels))
        else:
            self.lins.append(conv(in_channels, hidden_channels))
            if self.bn:
                self.bns.append(nn.BatchNorm1d(hidden_channels))
            for _ in range(num_layers - 2):
                self.lins.append(conv(hidden_channels, hidden_channels))
                if self.bn:
                    self.bns.append(nn.BatchNorm1d(hidden_channels))
            self.lins.append(conv(hidden_channels, out_channels))
        self.dropout = nn.Dropout(p=droput_rate)
        self.reset_parameters()


class AGNNConv(nn.Module):
    """models.py:377-405 (the reference's own AGNNConv: a Linear, no beta)."""

    def __init__(self, in_channels, out_channels, aggr='add', add_self_loops: bool = True):
        super().__init__()
        self.lin = nn.Linear(in_channels, out_channels)
        self.add_self_loops = add_self_loops      # stored, never read (models.py:386)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()

    def forward(self, x, edge_index):
        ei = agnn_edge_list(edge_index, x.size(0))
        x = self.lin(x)
        x_norm = F.normalize(x, p=2., dim=-1)
        return propagate_attention(x, x_norm, ei)[0]


class AGNN(_Stack):
    """models.py:336-374."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, bn=False):
        super().__init__()
        self.bn = bn
        self.lins = nn.ModuleList()
        if self.bn:
            self.bns = nn.ModuleList()
        if num_layers == 1:
            self.lins.append(AGNNConv(in_channels, out_channels))
        else:
            self.lins.append(AGNNConv(in_channels, hidden_channels))
            if self.bn:
                self.bns.append(nn.BatchNorm1d(hidden_channels))
            for _ in range(num_layers - 2):
                self.lins.append(AGNNConv(hidden_channels, hidden_channels))
                if self.bn:
                    self.bns.append(nn.BatchNorm1d(hidden_channels))
            self.lins.append(AGNNConv(hidden_channels, out_channels))
        self.dropout = nn.Dropout(p=0.5)
        self.reset_parameters()


# --------------------------------------------------------------------------
# GGCN's sparse layer (models.py:1453-1553) - the signed cosine attention that shares the
# gather skeleton.  All core torch in the reference: pinned by running the
# reference class itself (tests/golden/pin_reference.py: pin_ggcn), no third-party stand-in.
# --------------------------------------------------------------------------

def ggcn_degree_precompute(adj: Tensor) -> Tensor:
    """models.py:1691-1707 (GGCN.precompute_degree_s): a sparse matrix on adj's pattern with the
    value ``adj[i, i] / adj[i, j] - 1`` at entry (i, j) - the reference's Python loop, vectorised
    (``adj_diag`` lists the diagonal values in entry order and is indexed by ROW id, i.e. every
    row is assumed to have its diagonal entry, as a normalised adjacency with self-loops does)."""
    idx, val = adj._indices(), adj._values()
    diag = val[idx[0] == idx[1]]
    return torch.sparse_coo_tensor(idx, diag[idx[0]] / val - 1, adj.size())


def signed_attention_values(idx: Tensor, wh: Tensor) -> Tuple[Tensor, Tensor]:
    """models.py:1512-1519 (get_sparse_att): F.cosine_similarity of the rows at (idx[0], idx[1]),
    split into its positive and negative parts."""
    sim = F.cosine_similarity(wh[idx[0], :], wh[idx[1], :])
    return F.relu(sim), -F.relu(-sim)


class GGCNlayer_SP(nn.Module):
    """models.py:1453-1553."""

    def __init__(self, in_features, out_features, device, use_degree=True, use_sign=True, use_decay=True,
                 scale_init=0.5, deg_intercept_init=0.5):
        super().__init__()
        self.fcn = nn.Linear(in_features, out_features)
        self.use_degree, self.use_sign = use_degree, use_sign
        if use_degree:
            self.deg_coeff = Parameter(torch.tensor([0.5 if use_decay else deg_intercept_init, 0.0]))
        if use_sign:
            self.coeff = Parameter(torch.zeros(3))
            self.scale = Parameter((2.0 if use_decay else scale_init) * torch.ones(1))
        self.adj_remove_diag = None

    def forward(self, h, adj, degree_precompute):
        idx = adj._indices()
        if self.use_degree:        # :1508-1510 non_linear_degree
            sc = torch.sparse_coo_tensor(degree_precompute._indices(),
                                         F.softplus(self.deg_coeff[0] * degree_precompute._values() + self.deg_coeff[1]),
                                         degree_precompute.size())
        wh = self.fcn(h)
        if not self.use_sign:      # :1544-1549
            return torch.sparse.mm(adj * sc if self.use_degree else adj, wh)
        if self.adj_remove_diag is None:      # :1501-1506
            keep = idx[0] != idx[1]
            self.adj_remove_diag = torch.sparse_coo_tensor(idx[:, keep], adj._values()[keep], adj.size())
        pos, neg = signed_attention_values(idx, wh)
        e_pos = torch.sparse_coo_tensor(idx, pos, adj.size())
        e_neg = torch.sparse_coo_tensor(idx, neg, adj.size())
        if self.use_degree:
            att_pos, att_neg = self.adj_remove_diag * sc * e_pos, self.adj_remove_diag * sc * e_neg
        else:
            att_pos, att_neg = self.adj_remove_diag * e_pos, self.adj_remove_diag * e_neg
        prop_pos, prop_neg = torch.sparse.mm(att_pos, wh), torch.sparse.mm(att_neg, wh)
        coeff = F.softmax(self.coeff, dim=-1)
        scale = F.softplus(self.scale)
        return scale * (coeff[0] * prop_pos + coeff[1] * prop_neg + coeff[2] * wh)


# --------------------------------------------------------------------------
# Operator-level entry used by the parity tests: everything after ``lin``
# --------------------------------------------------------------------------

def aggregate_reference(h: Tensor, edge_index: Tensor, *, add_loops: bool = True,
                        remove_loops: bool = False, top_k: Optional[int] = None,
                        thr: float = 0.0, use_loop: bool = False, smax=None, smean=None):
    """The fused operator's contract, computed the reference's way.

    h: [N, C] fp32 (the output of ``self.lin``).  Returns a dict with
      out      [N, C]   scatter-mean of weight * h[src]
      ei       [2, E']  edge list after self-loop handling
      s        [E']     per-edge cosine
      weight   [E']     s where selected else 0
      sel_src  [N, k]   selected source ids in rank order, -1 padded (top_k only)
      sel_pos  [N, k]   selected positions in the E' edge list
    """
    N = h.size(0)
    ei = sn_edge_list(edge_index, N, add_loops, remove_loops)
    norm = F.normalize(h, p=2., dim=-1)
    out, s, weight, rounds = propagate_mean(h, norm, ei, top_k, thr, use_loop, smax, smean)
    res = dict(out=out, ei=ei, s=s, weight=weight)
    if top_k is not None:
        res["sel_src"], res["sel_pos"] = selected_sources(rounds, ei, N, top_k)
    return res


def adj_linear_reference(w_weight: Tensor, w_bias: Tensor, ei: Tensor, num_nodes: int) -> Tensor:
    """models.py:124-130: ``out_0 = self.w(adj)`` with adj built from the
    post-self-loop-handling edge list and ``row - row.min()``."""
    row, col = ei
    row = row - row.min()
    adj = sparse_adj_coo(row, col, num_nodes)
    return F.linear(adj, w_weight, w_bias)


# --------------------------------------------------------------------------
# Sim-GFA toolbox (SimGFAToolbox/dense.py, sparse.py) - small variants are the
# semantic spec
# --------------------------------------------------------------------------

def cosine_similarity_dense_small(x: Tensor) -> Tensor:
    """dense.py:138-141."""
    norm = F.normalize(x, p=2., dim=-1)
    return norm.mm(norm.t())


def node_similarity_dense_small(x: Tensor):
    """dense.py:144-149: all off-diagonal entries of S and their mean."""
    sim = cosine_similarity_dense_small(x)
    n = sim.shape[0]
    mask = ~torch.eye(n, dtype=torch.bool)
    sim = sim[mask]
    return sim, torch.mean(sim)


def linked_node_similarity_dense_small(x: Tensor, edge_index: Tensor):
    """dense.py:152-155."""
    sim = cosine_similarity_dense_small(x)
    sim = sim[edge_index[0], edge_index[1]]
    return sim.reshape(-1, 1), torch.mean(sim)


def neighborhood_similarity_dense_small(x: Tensor, edge_index: Tensor):
    """dense.py:158-164: per-edge cosine on raw features, scatter_mean by SOURCE
    (edge_index[0]); output length max(src)+1."""
    norm = F.normalize(x, p=2., dim=-1)
    sim_i = torch.index_select(norm, 0, edge_index[0])
    sim_j = torch.index_select(norm, 0, edge_index[1])
    sim = (sim_i * sim_j).sum(dim=-1)
    weight = scatter_mean_1d(sim, edge_index[0])
    return weight, torch.mean(weight)


def class_similarity_dense_small(x: Tensor, y: Tensor):
    """dense.py:167-179: mean of the S block for every ordered class pair.
    (``len(torch.unique(y))`` classes, labelled 0..n_classes-1.)"""
    sim = cosine_similarity_dense_small(x)
    n_classes = len(torch.unique(y))
    sim_matrix = torch.zeros(n_classes, n_classes)
    for i in range(n_classes):
        for j in range(n_classes):
            index_i = torch.where(y == i)[0]
            index_j = torch.where(y == j)[0]
            sim_matrix[i, j] = torch.mean(sim[index_i, :][:, index_j])
    return sim_matrix, torch.mean(sim_matrix)


def node_similarity_dense_large_parted(x: Tensor):
    """dense.py:9-30 including its operator-precedence quirk at :28:
    ``(sum - N) / (N - 1) * N`` (documented in SURVEY.md Appendix C)."""
    norm = F.normalize(x, p=2., dim=-1)
    n = norm.shape[0]
    sim_sum = norm.mm(norm.t()).sum()
    return None, (sim_sum - n) / (n - 1) * n


def class_similarity_dense_large(x: Tensor, y: Tensor) -> Tensor:
    """dense.py:104-130: block sums / block sizes == the small variant's means."""
    return class_similarity_dense_small(x, y)[0]


def sort_edge_index(edge_index: Tensor, num_nodes: Optional[int] = None) -> Tensor:
    """PyG 2.0.4 ``torch_geometric.utils.sort_edge_index(edge_index)`` as called at
    dense.py:34,66 and sparse.py:86: ``idx = row * num_nodes + col`` with
    ``num_nodes = edge_index.max() + 1``, ``perm = idx.argsort()``, columns permuted
    (sorted by source, then target)."""
    if num_nodes is None:
        num_nodes = int(edge_index.max()) + 1 if edge_index.numel() else 0
    idx = edge_index[0] * num_nodes + edge_index[1]
    return edge_index[:, idx.argsort(stable=True)]


def _out_neighbour_lists(edge_index: Tensor, num_nodes: int) -> List[Tensor]:
    """The scan of dense.py:38-48 (= :70-80, sparse.py:55-65, :91-101) on a SOURCE-SORTED
    edge list: node i's list is the run of edges whose source is i that starts at the scan
    cursor; the cursor only moves when a run is closed by an edge of a different source."""
    src = edge_index[0].tolist()
    dst = edge_index[1].tolist()
    lists, k, e = [], 0, len(src)
    for i in range(num_nodes):
        cur, j = [], k
        while j < e:
            if src[j] == i:
                cur.append(dst[j])
                j += 1
            else:
                k = j
                break
        lists.append(torch.tensor(cur, dtype=torch.long))
    return lists


def linked_node_similarity_dense_large(x: Tensor, edge_index: Tensor):
    """dense.py:33-62: S[k, linked(k)] listed node by node over the source-sorted edges."""
    ei = sort_edge_index(edge_index)
    norm = F.normalize(x, p=2., dim=-1)
    all_sim = []
    for k, linked in enumerate(_out_neighbour_lists(ei, norm.shape[0])):
        all_sim.append(torch.mm(norm[k, :].reshape(1, -1), norm.T).reshape(-1)[linked])
    all_sim = torch.concat(all_sim, -1)
    return all_sim.reshape(-1, 1), torch.mean(all_sim.reshape(-1, 1))


def neighborhood_similarity_dense_large(x: Tensor, edge_index: Tensor):
    """dense.py:65-101: per node the mean of S[k, linked(k)] (0 for a node without
    out-edges: ``sum(empty) / 1``), and ``sum_k avg_k * (1 / N)`` over ALL nodes."""
    ei = sort_edge_index(edge_index)
    norm = F.normalize(x, p=2., dim=-1)
    n = norm.shape[0]
    all_sim, mean_tmp = [], 0
    for k, linked in enumerate(_out_neighbour_lists(ei, n)):
        if linked.numel():
            avg = torch.sum(torch.mm(norm[k, :].reshape(1, -1), norm.T).reshape(-1)[linked]) / linked.numel()
        else:
            avg = torch.zeros(())
        mean_tmp = mean_tmp + avg * (1 / n)
        all_sim.append(avg)
    return torch.stack(all_sim).reshape(-1, 1), mean_tmp


# SimGFAToolbox/sparse.py: the arithmetic is scikit-learn's ``normalize(axis=0)`` (column
# L2 norms in float64, a zero column keeps norm 1) and scipy's sparse product, both float64;
# restated densely in float64 numpy, then read out the way the reference reads its rows
# (``torch.Tensor(sim.getrow(m).toarray())``: a cast to float32).

def cosine_similarity_sparse(mat):
    """sparse.py:8-14 for a scipy matrix or a dense array ``mat`` [rows, cols]:
    returns the dense float64 [cols, cols] array of ``Mn.T @ Mn``."""
    import numpy as np
    m = np.asarray(mat.todense() if hasattr(mat, "todense") else mat, dtype=np.float64)
    nrm = np.sqrt((m * m).sum(axis=0))
    nrm[nrm == 0.0] = 1.0
    mn = m / nrm
    return mn.T @ mn


def node_similarity_sparse(x):
    """sparse.py:17-42: all entries (diagonal included) as fp32 [N*N, 1] and their mean."""
    sim = torch.from_numpy(cosine_similarity_sparse(x)).to(torch.float32)
    return sim.reshape(-1, 1), torch.sum(sim) / sim.numel()


def linked_node_similarity_sparse(x, edge_index: Tensor):
    """sparse.py:45-77 (the edge list is NOT sorted there: source-sorted input expected)."""
    sim = torch.from_numpy(cosine_similarity_sparse(x)).to(torch.float32)
    all_sim = [sim[k][linked] for k, linked in enumerate(_out_neighbour_lists(edge_index, sim.shape[0]))]
    all_sim = torch.concat(all_sim, -1)
    return all_sim.reshape(-1, 1), torch.mean(all_sim.reshape(-1, 1))


def neighborhood_similarity_sparse(x, edge_index: Tensor):
    """sparse.py:80-120."""
    ei = sort_edge_index(edge_index)
    sim = torch.from_numpy(cosine_similarity_sparse(x)).to(torch.float32)
    n = sim.shape[0]
    all_sim, mean_tmp = [], 0
    for k, linked in enumerate(_out_neighbour_lists(ei, n)):
        avg = torch.sum(sim[k][linked]) / linked.numel() if linked.numel() else torch.zeros(())
        mean_tmp = mean_tmp + avg * (1 / n)
        all_sim.append(avg)
    return torch.stack(all_sim).reshape(-1, 1), mean_tmp


def class_similarity_sparse(x, y: Tensor) -> Tensor:
    """sparse.py:123-152: block sums / block sizes per ordered class pair."""
    sim = torch.from_numpy(cosine_similarity_sparse(x)).to(torch.float32)
    n_classes = len(torch.unique(y))
    out = torch.zeros(n_classes, n_classes)
    for i in range(n_classes):
        for j in range(n_classes):
            blk = sim[torch.where(y == i)[0], :][:, torch.where(y == j)[0]]
            out[i, j] = torch.sum(blk) / blk.numel()
    return out
