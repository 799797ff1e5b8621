"""Sim-GFA toolbox on the GPU: the dense-feature statistics of SimGFAToolbox/dense.py
under the reference's function names and return conventions, computed by
libsngnn_hip (no Python row / block loops, no N x N temporaries unless the function's
contract is to return them).  Inputs are GPU tensors; there is no CPU path.

SimGFAToolbox/sparse.py's five functions are covered too (second half of this file: the
column-normalised input stays sparse on the GPU); plot.py is not (matplotlib, out of scope).
"""
from __future__ import annotations

import torch

from . import _lib


def _x(x: torch.Tensor) -> torch.Tensor:
    if not x.is_cuda:
        raise ValueError("x must live on the GPU (there is no CPU path)")
    if x.dim() != 2:
        raise ValueError("x must be [N, F]")
    return x.to(torch.float32).contiguous()


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def cosine_similarity_dense_small(x: torch.Tensor) -> torch.Tensor:
    """dense.py:138-141: S = normalize(x) @ normalize(x).T, [N, N] (matrix cores: exact bf16 split of
    both operands, fp32 accumulation - an fp32 contraction's rounding; csrc/toolbox.hip)."""
    x = _x(x)
    n, f = x.shape
    s = torch.empty((n, n), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().sngnn_cosine_dense(x.data_ptr(), n, f, s.data_ptr(), _stream(x))
    _lib.check(rc, "sngnn_cosine_dense")
    return s


def edge_cosine(x: torch.Tensor, edge_index: torch.Tensor) -> torch.Tensor:
    """Per-edge cosine <n[ei[0]], n[ei[1]]> of raw feature rows, [E]."""
    x = _x(x)
    ei = edge_index.to(torch.int64).contiguous()
    if ei.device != x.device:
        raise ValueError("x and edge_index must be on the same device")
    e = ei.size(1)
    sim = torch.empty(e, dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().sngnn_edge_cosine(x.data_ptr(), x.size(0), x.size(1), ei.data_ptr(), e,
                                           sim.data_ptr(), _stream(x))
    _lib.check(rc, "sngnn_edge_cosine")
    return sim


def signed_edge_attention(indices: torch.Tensor, wh: torch.Tensor):
    """GGCNlayer_SP.get_sparse_att's values (models.py:1512-1519): the per-edge cosine
    of ``wh`` rows split into its positive and negative parts, (relu(s), -relu(-s)),
    each [nnz] in the order of ``indices`` ([2, nnz], the adjacency's COO indices).
    Forward only (no autograd); rows below 1e-8 in norm follow F.normalize's clamp."""
    s = edge_cosine(wh, indices)
    return torch.relu(s), -torch.relu(-s)


def class_block_sums(x: torch.Tensor, y: torch.Tensor, n_classes: int):
    """(sums [c, c] f64 of S over every class pair, diagonal sum f64) without forming S:
    sum_{i in A, j in B} <n_i, n_j> = <sum_A n, sum_B n>."""
    x = _x(x)
    y32 = y.to(device=x.device, dtype=torch.int32).contiguous()
    sums = torch.zeros((n_classes, n_classes), dtype=torch.float64, device=x.device)
    diag = torch.zeros(1, dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        rc = _lib.load().sngnn_cosine_class_sums(x.data_ptr(), x.size(0), x.size(1),
                                                 y32.data_ptr(), n_classes, sums.data_ptr(),
                                                 diag.data_ptr(), _stream(x))
    _lib.check(rc, "sngnn_cosine_class_sums")
    return sums, diag


def node_similarity_dense_small(x):
    """dense.py:144-149: every off-diagonal entry of S (row-major) and their mean."""
    s = cosine_similarity_dense_small(x)
    n = s.size(0)
    mask = ~torch.eye(n, dtype=torch.bool, device=s.device)
    sim = s[mask]
    return sim, torch.mean(sim)


def node_similarity_dense_large_parted(x, corrected: bool = False):
    """dense.py:9-30 without the 1000-row block loops.  The reference's last line
    has an operator-precedence slip, ``(sum - N) / (N - 1) * N`` (dense.py:28); that
    value is returned by default, ``corrected=True`` gives the mean over the
    N (N - 1) off-diagonal pairs."""
    x = _x(x)
    n = x.size(0)
    sums, diag = class_block_sums(x, torch.zeros(n, dtype=torch.int32, device=x.device), 1)
    total = sums[0, 0]
    if corrected:      # the true diagonal: all-zero rows contribute 0, not 1
        return None, ((total - diag[0]) / (n * (n - 1))).to(torch.float32)
    return None, ((total - n) / (n - 1) * n).to(torch.float32)


def linked_node_similarity_dense_small(x, edge_index):
    """dense.py:152-155: S[ei[0], ei[1]] - computed per edge, S is never formed."""
    sim = edge_cosine(x, edge_index)
    return sim.reshape(-1, 1), torch.mean(sim)


def linked_node_similarity_dense_large(x, edge_index):
    """dense.py:33-62: the same values listed by source node (edges sorted by
    (src, dst) as PyG's sort_edge_index does)."""
    ei = _sort_edge_index(edge_index)
    sim = edge_cosine(x, ei)
    return sim.reshape(-1, 1), torch.mean(sim.reshape(-1, 1))


def _sort_edge_index(edge_index):
    n = int(edge_index.max()) + 1 if edge_index.numel() else 1
    key = edge_index[0] * n + edge_index[1]
    return edge_index[:, torch.argsort(key, stable=True)]


def segment_mean(val: torch.Tensor, index: torch.Tensor, length: int):
    """``sngnn_segment_mean``: torch_scatter's ``scatter_mean(val, index, dim=0)`` (dense.py:163) in
    its own arithmetic - each group's values added in ENTRY order in fp32, divided by
    max(count, 1), 0 for an empty group - by a kernel without atomics (bit-identical runs, and
    bit-identical to the CPU's serial scatter).  Returns (mean fp32 [length], count int32 [length])."""
    val = val.to(torch.float32).contiguous()
    if not val.is_cuda:
        raise ValueError("val must live on the GPU (there is no CPU path)")
    idx = index.to(device=val.device, dtype=torch.int64).contiguous()
    if idx.numel() != val.numel() or val.dim() != 1:
        raise ValueError("val and index must be 1-D and of one length")
    mean = torch.empty(int(length), dtype=torch.float32, device=val.device)
    cnt = torch.empty(int(length), dtype=torch.int32, device=val.device)
    with torch.cuda.device(val.device):
        rc = _lib.load().sngnn_segment_mean(val.data_ptr(), idx.data_ptr(), val.numel(), int(length),
                                            mean.data_ptr(), cnt.data_ptr(), _stream(val))
    _lib.check(rc, "sngnn_segment_mean")
    return mean, cnt


def neighborhood_similarity_dense_small(x, edge_index):
    """dense.py:158-164: per-edge cosine, mean grouped by SOURCE (edge_index[0]) in edge order;
    output length max(src) + 1 like torch_scatter.scatter_mean without dim_size."""
    sim = edge_cosine(x, edge_index)
    src = edge_index[0].to(sim.device)
    length = int(src.max()) + 1 if src.numel() else 0
    weight, _ = segment_mean(sim, src, length)
    return weight, torch.mean(weight)


def neighborhood_similarity_dense_large(x, edge_index):
    """dense.py:65-101: one value per node - the mean over its out-edges in (src, dst) order
    (``sort_edge_index``, dense.py:66), 0 for a node without out-edges - and the mean over ALL nodes."""
    n = x.size(0)
    ei = _sort_edge_index(edge_index)
    per_node, _ = segment_mean(edge_cosine(x, ei), ei[0], n)
    return per_node.reshape(-1, 1), per_node.sum() / n


def class_similarity_dense_small(x, y):
    """dense.py:167-179: mean of S over every ordered class pair and the mean of that
    matrix (classes are ``0 .. len(unique(y)) - 1`` as in the reference)."""
    n_classes = len(torch.unique(y))
    sums, _ = class_block_sums(x, y, n_classes)
    cnt = torch.bincount(y.to(sums.device).long(), minlength=n_classes).double()
    mat = (sums / (cnt[:, None] * cnt[None, :])).to(torch.float32)
    return mat, torch.mean(mat)


def class_similarity_dense_large(x, y):
    """dense.py:104-130: block sums / block sizes (== the small variant's matrix)."""
    return class_similarity_dense_small(x, y)[0]


# ---------------------------------------------------------------------------
# SimGFAToolbox/sparse.py: the same statistics on the COLUMN-normalised sparse input
# (``cosine_similarity_sparse`` returns M_n^T M_n - node-to-node cosine when M is the
# [N, N] adjacency, as in toolbox-example.py:28-29).  The reference forms the scipy product
# and then reads it row by row in Python (``sim.getrow(k).toarray()``, sparse.py:30,68,106).
# Here the input stays sparse on the GPU (CSC: the normalised values by column): the linked /
# neighbourhood statistics are column-pair dot products (``sngnn_sparse_pair_dot``), the
# class matrix is <m_A, m_B> with m_A the sum of class A's normalised columns - O(nnz), the
# product is never formed.  Only the functions whose RESULT is the whole [cols, cols] matrix
# (``cosine_similarity_sparse``, ``node_similarity_sparse``) densify, through the MFMA cosine.
# ---------------------------------------------------------------------------
_DENSE_LIMIT = 1 << 31      # elements of the densified input / of the [M, M] result


class _SparseCols:
    """The column-normalised input in CSC form on the GPU (sklearn ``normalize(axis=0)``:
    a column of zeros stays zero)."""

    def __init__(self, mat, device=None):
        import numpy as np
        if isinstance(mat, torch.Tensor):
            t = mat.to_sparse() if mat.layout == torch.strided else mat
            t = t.coalesce()
            if device is not None:
                t = t.to(device)
            idx, val = t.indices(), t.values()
            self.shape = tuple(t.shape)
        else:                                   # scipy.sparse matrix
            coo = mat.tocoo()
            coo.sum_duplicates()
            if device is None:
                device = torch.device("cuda", torch.cuda.current_device())
            idx = torch.from_numpy(np.vstack([coo.row, coo.col]).astype(np.int64)).to(device)
            val = torch.from_numpy(coo.data.astype(np.float32)).to(device)
            self.shape = tuple(coo.shape)
        if not val.is_cuda:
            raise ValueError("the matrix must live on the GPU (or be a scipy.sparse matrix)")
        rows, cols = idx[0], idx[1]
        n_cols = self.shape[1]
        order = torch.argsort(cols * self.shape[0] + rows)          # by column, rows ascending
        self.rowidx = rows[order].to(torch.int32).contiguous()
        col_s, val_s = cols[order], val[order].to(torch.float32)
        counts = torch.bincount(col_s, minlength=n_cols)
        # (entries are in column order: a column's squares are one contiguous segment, added serially by one
        # thread - no float atomics, the same bits on every run)
        sq = torch.segment_reduce(val_s.double() ** 2, "sum", lengths=counts, unsafe=True)
        nrm = sq.sqrt()
        inv = torch.where(nrm > 0, 1.0 / nrm, torch.zeros_like(nrm))
        self.vals = (val_s.double() * inv[col_s]).to(torch.float32).contiguous()
        self.cols = col_s
        self.colptr = torch.zeros(n_cols + 1, dtype=torch.int64, device=val.device)
        self.colptr[1:] = torch.cumsum(counts, 0)
        self.device = val.device

    def pair_dot(self, a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        a = a.to(self.device, torch.int64).contiguous()
        b = b.to(self.device, torch.int64).contiguous()
        out = torch.empty(a.numel(), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = _lib.load().sngnn_sparse_pair_dot(self.colptr.data_ptr(), self.rowidx.data_ptr(),
                                                   self.vals.data_ptr(), self.shape[1], a.data_ptr(),
                                                   b.data_ptr(), a.numel(), out.data_ptr(),
                                                   torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "sngnn_sparse_pair_dot")
        return out

    def dense(self) -> torch.Tensor:
        if self.shape[0] * self.shape[1] > _DENSE_LIMIT or self.shape[1] ** 2 > _DENSE_LIMIT:
            raise ValueError("the full [cols, cols] similarity does not fit: use the linked / "
                             "neighbourhood / class statistics, which never form it")
        m = torch.zeros(self.shape, dtype=torch.float32, device=self.device)
        m[self.rowidx.long(), self.cols] = self.vals
        return m


def _group_sum_f64(val: torch.Tensor, key: torch.Tensor, n_groups: int) -> torch.Tensor:
    """``out[g] = sum of val[key == g]`` in float64 with a FIXED order of additions: entries are brought
    into (group, entry position) order by a stable sort and each group is added serially by one thread
    (``torch.segment_reduce``) - where ``index_add_`` would use float atomics, whose order changes from
    run to run."""
    order = torch.argsort(key, stable=True)
    counts = torch.bincount(key, minlength=n_groups)
    return torch.segment_reduce(val.double()[order], "sum", lengths=counts, unsafe=True)


def cosine_similarity_sparse(mat, device=None) -> torch.Tensor:
    """sparse.py:8-14: normalise the COLUMNS of ``mat`` and return ``M_n.T @ M_n``
    ([cols, cols], dense on the GPU; the whole matrix is the result, so it is formed)."""
    m = _SparseCols(mat, device).dense()
    return cosine_similarity_dense_small(m.t().contiguous())


def node_similarity_sparse(x, device=None):
    """sparse.py:17-42: every entry of the similarity (diagonal included, as the
    reference lists them) and their mean."""
    sim = cosine_similarity_sparse(x, device)
    return sim.reshape(-1, 1), torch.mean(sim)


def linked_node_similarity_sparse(x, edge_index, device=None):
    """sparse.py:45-77: the similarity at the linked pairs (edge order preserved)."""
    sp = _SparseCols(x, device)
    vals = sp.pair_dot(edge_index[0], edge_index[1])
    return vals.reshape(-1, 1), torch.mean(vals)


def neighborhood_similarity_sparse(x, edge_index, device=None):
    """sparse.py:80-120: per node the mean similarity to its out-neighbours (0 for a
    node without out-edges) and the mean over all nodes."""
    sp = _SparseCols(x, device)
    ei = _sort_edge_index(edge_index.to(sp.device))          # sparse.py:86
    n = sp.shape[1]
    per_node, _ = segment_mean(sp.pair_dot(ei[0], ei[1]), ei[0], n)
    return per_node.reshape(-1, 1), per_node.sum() / n


def class_similarity_sparse(x, y, device=None):
    """sparse.py:123-152: block sums / block sizes of the similarity per class pair -
    sum_{i in A, j in B} <c_i, c_j> = <m_A, m_B>, m_A = the sum of class A's normalised columns."""
    sp = _SparseCols(x, device)
    yl = y.to(sp.device).long()
    n_classes = len(torch.unique(yl))
    rows = sp.shape[0]
    m = _group_sum_f64(sp.vals, yl[sp.cols] * rows + sp.rowidx.long(), n_classes * rows).view(n_classes, rows)
    sums = m @ m.t()
    cnt = torch.bincount(yl, minlength=n_classes).double()
    return (sums / (cnt[:, None] * cnt[None, :])).to(torch.float32)


def edge_index_to_sparse_csc_tensor(x, edge_index):
    """SimGFAToolbox/utils.py:5-11: the [N, N] adjacency (ones; duplicates add up) as a
    scipy CSC matrix, N = len(x) - the input ``toolbox-example.py`` hands to the
    ``*_sparse`` functions.  Host code, as in the reference."""
    import numpy as np
    from scipy import sparse as sp
    n = len(x)
    ei = edge_index.detach().cpu().numpy()
    return sp.csc_matrix((np.full(ei.shape[1], 1), (ei[0], ei[1])), shape=(n, n))


# ---------------------------------------------------------------------------
# kNN similarity graph: the edges the aggregation layers consume, straight from x
# ---------------------------------------------------------------------------
def knn_graph(x: torch.Tensor, k: int, exclude_self: bool = True):
    """For every node the ``k`` most cosine-similar nodes (``sngnn_knn_graph``: tiled fp32
    MFMA with a fused per-row top-k; the [N, N] similarity of dense.py:138-141 is never
    stored).  Returns (idx int64 [N, k] in rank order, -1 padded; sim fp32 [N, k])."""
    x = _x(x)
    n, f = x.shape
    lib = _lib.load()
    idx = torch.empty((n, k), dtype=torch.int32, device=x.device)
    sim = torch.empty((n, k), dtype=torch.float32, device=x.device)
    ws = torch.empty(max(int(lib.sngnn_knn_workspace_bytes(n, int(k))), 256), dtype=torch.uint8, device=x.device)
    with torch.cuda.device(x.device):
        rc = lib.sngnn_knn_graph(x.data_ptr(), n, f, int(k), int(bool(exclude_self)), idx.data_ptr(),
                                 sim.data_ptr(), ws.data_ptr(), _stream(x))
    _lib.check(rc, "sngnn_knn_graph")
    return idx.long(), sim


def knn_edge_index(x: torch.Tensor, k: int, exclude_self: bool = True) -> torch.Tensor:
    """The kNN graph as an ``edge_index`` [2, E] for the conv layers: source = neighbour,
    target = node, targets ascending, a node's in-edges in similarity order."""
    idx, _ = knn_graph(x, k, exclude_self)
    n = idx.size(0)
    dst = torch.arange(n, device=idx.device).repeat_interleave(idx.size(1))
    src = idx.reshape(-1)
    keep = src >= 0
    return torch.stack([src[keep], dst[keep]])
