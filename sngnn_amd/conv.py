"""SNConv / SNConv_plus / SNConv_plus_plus with the reference's constructor
signatures, parameter names and init order (models/models.py:305-334, 214-263,
89-158), running the fused HIP aggregation instead of PyG's per-edge message
passing.  ``forward(x, edge_index) -> [N, out_channels]`` as in the reference.

What stays PyTorch: ``self.lin`` (one rocBLAS GEMM, models.py:121,237,324), the
bias add and the scalar blend.  Everything between ``lin`` and the bias is one
call into libsngnn_hip (plus one for the SNGNN++ adjacency branch).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import dist as sn_dist
from . import ops
from .graph import GLOBAL_CACHE, LOOPS_REPLACE


def _graph_for(x: torch.Tensor, edge_index: torch.Tensor, add_loops: bool, remove_loops: bool):
    if edge_index.device != x.device:
        raise ValueError("x and edge_index must be on the same device")
    part = sn_dist.current_partition()
    if part is None:
        return GLOBAL_CACHE.get(edge_index, x.size(0), add_loops, remove_loops)
    return GLOBAL_CACHE.get(edge_index, part.n_total, add_loops, remove_loops,
                            row_range=(part.row_begin, part.row_end))


def _lin_aligned(x: torch.Tensor, lin: nn.Linear):
    """``h = lin(x)`` with the channel count rounded up to a multiple of 4 by zero
    weights (returns h and the true width).  Rows of 4k floats are 16-byte aligned, so
    the kernels read them with 16-byte lane loads (2.5x faster than the dword path at
    C = 47); zero channels change neither a cosine nor a weighted sum, and autograd
    slices their (zero) gradients away."""
    c = lin.out_features
    cp = (c + 3) // 4 * 4
    if cp == c or c < 16 or not x.is_cuda:
        return ops.linear(x, lin), c
    w = torch.cat([lin.weight, lin.weight.new_zeros(cp - c, lin.in_features)], dim=0)
    b = None if lin.bias is None else torch.cat([lin.bias, lin.bias.new_zeros(cp - c)])
    return ops._Linear.apply(x, w, b), c


def _aggregate(h: torch.Tensor, graph, top_k, thr: float) -> torch.Tensor:
    """Fused aggregation of the local rows; under a node-range partition the
    feature shards are all-gathered first (RCCL), see sngnn_amd/dist.py."""
    part = sn_dist.current_partition()
    if part is not None:
        h = sn_dist.all_gather_rows(h, part)
    return ops.aggregate(h, graph, top_k, thr)


class SNConv(nn.Module):
    """models.py:305-334: self-loops added (never removed), every in-edge weighted
    by its cosine, mean over the full in-degree, ``bias=True`` by default."""

    def __init__(self, in_channels, out_channels, aggr='mean', bias: bool = True):
        super().__init__()
        if aggr != 'mean':
            raise ValueError("only aggr='mean' is implemented (the reference never uses another)")
        self.lin = nn.Linear(in_channels, out_channels)
        if bias:
            self.bias = Parameter(torch.empty(out_channels))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()
        if self.bias is not None:        # PyG inits.zeros(None) is a no-op
            self.bias.data.fill_(0)

    def forward(self, x, edge_index):
        graph = _graph_for(x, edge_index, True, False)
        h, c = _lin_aligned(x, self.lin)
        out = _aggregate(h, graph, None, 0.0)[:, :c]
        if self.bias is not None:
            out = out + self.bias
        return out


class AGNNConv(nn.Module):
    """models.py:377-405, the reference's own cosine-attention layer:
    original loops replaced by one per node, ``alpha`` = per-target softmax of the
    cosines, ``aggr='add'``.  Same kernel skeleton as SNConv, softmax instead of mean."""

    def __init__(self, in_channels, out_channels, aggr='add', add_self_loops: bool = True):
        super().__init__()
        if aggr != 'add':
            raise ValueError("only aggr='add' is implemented (the reference's setting)")
        self.lin = nn.Linear(in_channels, out_channels)
        self.add_self_loops = add_self_loops      # stored, never read (models.py:386)
        self.reset_parameters()

    def reset_parameters(self):
        self.lin.reset_parameters()

    def forward(self, x, edge_index):
        graph = _graph_for(x, edge_index, True, LOOPS_REPLACE)
        h, c = _lin_aligned(x, self.lin)
        part = sn_dist.current_partition()
        if part is not None:
            h = sn_dist.all_gather_rows(h, part)
        return ops.attention(h, graph)[:, :c]


class SNConv_plus(nn.Module):
    """models.py:214-263: per target keep the ``top_k`` highest-cosine in-edges that
    also reach ``thr``; ``is_remove_self_loops`` drops ALL loops (added and original)."""

    def __init__(self, in_channels, out_channels, num_nodes, top_k=2, thr=0.0,
                 is_remove_self_loops=True, bias: bool = False, aggr='mean'):
        super().__init__()
        if aggr != 'mean':
            raise ValueError("only aggr='mean' is implemented (the reference never uses another)")
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
        graph = _graph_for(x, edge_index, True, bool(self.is_remove_self_loops))
        h, c = _lin_aligned(x, self.lin)
        out = _aggregate(h, graph, int(self.top_k), float(self.thr))[:, :c]
        if self.bias is not None:
            out = out + self.bias
        return out


class _AdjLinearParams(nn.Module):
    """Holds ``w.weight`` [C, N] and ``w.bias`` [C] under the reference's names
    (models.py:95).  The weight is stored column-major - ``weight.t()`` is a
    contiguous [N, C] table - so the adjacency branch gathers whole rows; its
    shape, ``state_dict`` key and initial values (same RNG stream as
    ``nn.Linear(num_nodes, C).reset_parameters()``) are the reference's."""

    def __init__(self, num_nodes: int, out_channels: int):
        super().__init__()
        self.in_features, self.out_features = num_nodes, out_channels
        self.weight = Parameter(torch.empty(num_nodes, out_channels).t())
        self.bias = Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        w = torch.empty(self.out_features, self.in_features, device=self.weight.device)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))          # nn.Linear.reset_parameters
        with torch.no_grad():
            self.weight.copy_(w)
        bound = 1 / math.sqrt(self.in_features) if self.in_features > 0 else 0
        nn.init.uniform_(self.bias, -bound, bound)

    def _apply(self, fn, recurse=True):
        # .to()/.cuda() would re-materialise the weight row-major; restore the layout
        super()._apply(fn, recurse)
        w = self.weight
        if w.dim() == 2 and not w.t().is_contiguous():
            with torch.no_grad():
                fixed = w.data.t().contiguous().t()
            self.weight = Parameter(fixed, requires_grad=w.requires_grad)
        return self


class SNConv_plus_plus(nn.Module):
    """models.py:89-158: SNConv_plus blended with ``Linear(num_nodes, C)`` applied to
    the sparse adjacency, ``out = beta * out_0 + (1 - beta) * out_1``."""

    def __init__(self, in_channels, out_channels, num_nodes, top_k=2, thr=0.0, init_beta=0.5,
                 is_remove_self_loops=True, bias: bool = False, aggr='mean'):
        super().__init__()
        if aggr != 'mean':
            raise ValueError("only aggr='mean' is implemented (the reference never uses another)")
        self.top_k = top_k
        self.thr = thr
        self.w = _AdjLinearParams(num_nodes, out_channels)
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

    def _flipped(self, edge_index):
        """edge_index with source and target rows swapped, cached per tensor so the
        graph cache (keyed on tensor identity) sees a stable object."""
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape))
        hit = getattr(self, "_flip_cache", None)
        if hit is None or hit[0] != key:
            hit = (key, edge_index.flip(0).contiguous(), edge_index)
            self._flip_cache = hit
        return hit[1]

    def forward(self, x, edge_index):
        part = sn_dist.current_partition()
        if part is None and x.size(0) != self.num_nodes:
            raise ValueError(f"built for {self.num_nodes} nodes, got {x.size(0)} "
                             "(the adjacency branch is Linear(num_nodes, C))")
        graph = _graph_for(x, edge_index, True, bool(self.is_remove_self_loops))
        h, c = _lin_aligned(x, self.lin)
        if part is None:
            out_0 = ops.adj_linear(self.w.weight, self.w.bias, graph)
        else:
            # multi-GPU: edge_index must hold every edge incident to the owned nodes (the
            # full list is fine); the out-edges come from the partition of the flipped
            # list.  w.weight is replicated ([C, N_total]); its gradient is each rank's
            # partial sum, all-reduced with the other parameters (dist.allreduce_grads).
            if part.n_total != self.num_nodes:
                raise ValueError("num_nodes must be the partition's N_total")
            flipped = self._flipped(edge_index)
            graph_out = GLOBAL_CACHE.get(flipped, part.n_total, True, bool(self.is_remove_self_loops),
                                         row_range=(part.row_begin, part.row_end))
            out_0 = ops.adj_linear_partition(self.w.weight, self.w.bias, graph_out)
        out_1 = _aggregate(h, graph, int(self.top_k), float(self.thr))[:, :c]
        out = ops.blend(out_0, out_1, self.beta)
        if self.bias is not None:
            out = out + self.bias
        return out
