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
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import dist as sn_dist
from . import ops
from .graph import GLOBAL_CACHE, LOOPS_REPLACE


class _Shard:
    """What a conv layer needs on one rank of a node-range partition: the local graph and
    how to turn the rank's feature rows into the table that graph's columns address."""

    def __init__(self, edge_index, part, add_loops, remove_loops):
        self.part = part
        if part.exchange == "halo":
            self.plan = sn_dist.HaloPlan(edge_index, part)
            self.graph = GLOBAL_CACHE.get(self.plan.edge_index, self.plan.table_rows, add_loops, remove_loops,
                                          row_range=(0, part.n_local))
        else:
            self.plan = None
            self.graph = GLOBAL_CACHE.get(edge_index, part.n_total, add_loops, remove_loops,
                                          row_range=(part.row_begin, part.row_end))

    def table(self, rows_local: torch.Tensor, table: torch.Tensor = None) -> torch.Tensor:
        """The feature table the local graph's columns address.  ``table``: the preallocated
        [own | halo] buffer whose head ``rows_local`` already is (halo form)."""
        if self.plan is not None:
            return sn_dist.halo_exchange(rows_local, self.plan, table)
        return sn_dist.all_gather_rows(rows_local, self.part)


_SHARDS = {}


def _shard_for(edge_index, part, add_loops, remove_loops) -> _Shard:
    key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version, str(edge_index.device),
           id(part), part.exchange, bool(add_loops), int(remove_loops))
    hit = _SHARDS.get(key)
    if hit is None:
        if len(_SHARDS) >= 16:
            _SHARDS.pop(next(iter(_SHARDS)))
        hit = (_Shard(edge_index, part, add_loops, remove_loops), edge_index, part)   # keeps the key's objects alive
        _SHARDS[key] = hit
    return hit[0]


def _graph_for(x: torch.Tensor, edge_index: torch.Tensor, add_loops: bool, remove_loops: bool):
    """(graph, shard): shard is None on one GPU."""
    if edge_index.device != x.device:
        raise ValueError("x and edge_index must be on the same device")
    part = sn_dist.current_partition()
    if part is None:
        return GLOBAL_CACHE.get(edge_index, x.size(0), add_loops, remove_loops), None
    if x.size(0) != part.n_local:
        raise ValueError(f"under a partition x must be the rank's {part.n_local} rows, got {x.size(0)}")
    shard = _shard_for(edge_index, part, add_loops, remove_loops)
    return shard.graph, shard


class LinFold:
    """An affine map of the layer's INPUT channels folded into ``lin``: ``lin(x * scale + shift)`` computed as
    ``x (W diag(scale))^T + (W shift + b)`` - how an evaluation-mode BatchNorm1d in front of a conv layer
    (models.py:207-208: relu -> bn -> dropout -> next conv) costs no pass over [N, C] at all: its running
    statistics are constants, so it is a per-channel scale and shift, and those belong to the next ``lin``'s
    weights.  (Training-mode batch norm needs the batch's statistics and stays a kernel of its own.)"""

    def __init__(self, scale: torch.Tensor, shift: torch.Tensor):
        self.scale, self.shift = scale, shift

    @staticmethod
    def of_batch_norm(bn: nn.BatchNorm1d) -> "LinFold":
        inv = torch.rsqrt(bn.running_var + bn.eps)
        scale = inv if bn.weight is None else bn.weight * inv
        shift = -bn.running_mean * scale
        if bn.bias is not None:
            shift = shift + bn.bias
        return LinFold(scale, shift)

    def apply(self, lin: nn.Linear):
        w = lin.weight * self.scale.unsqueeze(0)
        b = torch.mv(lin.weight, self.shift)
        if lin.bias is not None:
            b = b + lin.bias
        return w, b


def _lin_aligned(x: torch.Tensor, lin: nn.Linear, shard=None, unit=None, act_in=None, fold: "Optional[LinFold]" = None):
    """``h = lin(x)`` with the channel count rounded up to a multiple of 4 by zero
    weights (returns h, the true width and the rank's feature table or None).  Rows of 4k
    floats are 16-byte aligned, so the kernels read them with 16-byte lane loads (2.5x faster
    than the dword path at C = 47); zero channels change neither a cosine nor a weighted sum,
    and their (zero) gradients are dropped (ops._Linear: persistent padded buffers, no
    per-forward cat).  Under a halo partition ``lin`` writes straight into the head of a fresh
    [own | halo] table (no copy, no concatenation later)."""
    c = lin.out_features
    cp = (c + 3) // 4 * 4
    pad = None if (cp == c or c < 16 or not x.is_cuda) else cp
    weight, bias = (lin.weight, lin.bias) if fold is None else fold.apply(lin)
    if shard is None or shard.plan is None or not x.is_cuda or x.dtype != torch.float32:
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and lin.weight.dtype == torch.float32):
            return F.linear(x, weight, bias), c, None
        # ``unit`` (single GPU): F.normalize of h from the same launch, for the aggregation that follows
        return ops._Linear.apply(x, weight, bias, pad, None, unit if shard is None else None,
                                 act_in if shard is None else None), c, None
    table = sn_dist.new_table(shard.plan, pad or c, x)
    head = table[:shard.plan.n_local]
    return ops._Linear.apply(x, weight, bias, pad, ops.OutBuffer(head), None, None), c, table


class _FilterHint:
    """Whether a layer's rows give the fp16 filter anything to prune.  The library switches the filter on
    by the call's knobs (a threshold >= 0.25); what it cannot see is the data: where nearly every in-edge
    passes the threshold - a deep layer's rows are nearly parallel - every edge is a candidate and the
    filter is a pass for nothing (an arxiv-sized second layer at thr 0.99: 56.6 us without, 69.3 with).
    The layer therefore looks at its own ``h`` now and then - the cosines of 4 096 sampled edges, outside
    graph captures, on its first forward and every 64th eager one after - and tells LATER forwards
    (``sngnn_epilogue_t.no_filter``).  A matter of time only: the results do not depend on the filter.

    The forward never waits for the answer (no host sync inside forward / backward - the
    reference has one at models.py:257): the probe's verdict is one byte the device copies into pinned
    host memory behind the sampled cosines; a later forward reads it once the copy's event has
    completed (``Event.query``: non-blocking) and keeps the previous verdict until then."""
    SAMPLE, EVERY = 4096, 64

    def __init__(self):
        self.no_filter = False
        self._calls = 0
        self._flag = None          # pinned uint8 [1]: the pending probe's verdict
        self._event = None         # recorded behind the copy into _flag; None = nothing pending

    def due(self) -> bool:
        self._calls += 1
        return (self._calls % self.EVERY == 1) and not torch.cuda.is_current_stream_capturing()

    def poll(self) -> None:
        """Take a finished probe's verdict (never blocks; a probe still in flight stays pending)."""
        # (an event query is not a legal call while a stream captures: the verdict waits for an eager forward)
        if self._event is not None and not torch.cuda.is_current_stream_capturing() and self._event.query():
            self.no_filter = bool(self._flag[0] != 0)          # (host memory: no device access)
            self._event = None

    @torch.no_grad()
    def probe(self, h: torch.Tensor, edge_index: torch.Tensor, thr: float, k_prunes: bool = False) -> None:
        """``k_prunes``: the library uses the filter on this graph whatever the threshold (top_k itself prunes
        its long rows: ``ops.filter_wanted(graph, C, top_k, 0.0)``) - then edges passing the threshold are no
        reason to switch it off; rows whose cosines do not SPREAD (the approximate scores cannot separate
        them: every edge is a candidate) are, in either case."""
        e = edge_index.size(1)
        if e == 0 or h.size(0) == 0 or self._event is not None:
            return
        gen = torch.Generator(device=edge_index.device)          # (its own generator: the global CUDA stream of
        gen.manual_seed(0x5EED + self._calls)                     # random numbers - dropout masks - is not touched)
        pick = torch.randint(0, e, (min(self.SAMPLE, e),), device=edge_index.device, generator=gen)
        src, dst = edge_index[0].index_select(0, pick), edge_index[1].index_select(0, pick)
        ok = (src < h.size(0)) & (dst < h.size(0))
        hs = h.index_select(0, src.clamp_max(h.size(0) - 1))
        hd = h.index_select(0, dst.clamp_max(h.size(0) - 1))
        s = F.cosine_similarity(hs, hd, dim=1)
        passing = ((s >= thr - 1.1e-3) & ok).float().mean() > 0.5
        srt = torch.sort(s).values                                  # (10th .. 90th percentile: static indices, no sync)
        m = srt.numel()
        flat = (srt[(9 * m) // 10 - (1 if m >= 10 else 0)] - srt[m // 10]) < 16 * 1.1e-3
        verdict = (flat if k_prunes else (flat | passing)).to(torch.uint8).reshape(1)
        if self._flag is None:
            self._flag = torch.zeros(1, dtype=torch.uint8).pin_memory()
        self._flag.copy_(verdict, non_blocking=True)
        self._event = torch.cuda.Event()
        self._event.record(torch.cuda.current_stream(h.device))


def _unit_for(lin: nn.Linear, graph, top_k, thr, hint: "Optional[_FilterHint]" = None) -> "ops.UnitRows":
    """The holder ``lin``'s normalising epilogue fills (ops.UnitRows); it asks for the fp16 filter
    rows only when the forward that follows will read them."""
    c = lin.out_features
    cp = c if (c % 4 == 0 or c < 16) else (c + 3) // 4 * 4
    want = ops.filter_wanted(graph, cp, int(top_k), float(thr))
    return ops.UnitRows(want, no_filter=want and hint is not None and hint.no_filter)


def _true_width(out: torch.Tensor, c: int) -> torch.Tensor:
    """Drop the zero channels `_lin_aligned` appended - only when it did: a full-range slice is
    still an autograd node whose backward zero-fills a [N, C] tensor and copies the gradient
    into it (two passes over 27-43 MB per layer at arxiv size for nothing)."""
    return out if out.size(1) == c else out[:, :c]


def _fuse_head(head, graph, h: torch.Tensor, c: int, shard, top_k) -> bool:
    """Whether this (last) layer's rows go through the classification head inside the aggregation's launches
    (ops.HeadEpilogue): one GPU, unpadded rows of 16-byte vectors, at most 64 classes, a graph whose
    split rows take the candidate finalize."""
    ok = (head is not None and shard is None and h.is_cuda and h.size(1) == c and c % 4 == 0 and c <= 64
          and ops.head_supported(graph, c, top_k))
    if head is not None:
        head.applied = ok
    return ok


def _fuse_epilogue(epilogue, h: torch.Tensor, c: int, shard) -> bool:
    """Whether this layer's output takes the fused store epilogue (ops.HiddenEpilogue): one GPU,
    no channel padding (the bias and the keep mask are [.., C] of the layer's own width)."""
    ok = epilogue is not None and shard is None and h.is_cuda and h.size(1) == c and c % 4 == 0
    if epilogue is not None:
        epilogue.applied = ok
    return ok


def _aggregate(h: torch.Tensor, graph, shard, top_k, thr: float, table=None, unit=None) -> torch.Tensor:
    """Fused aggregation of the local rows; under a node-range partition the feature rows the
    rank's in-edges reference are exchanged first (RCCL) - in the halo form overlapped with the
    aggregation of the rows that need none of them (sngnn_amd/dist.py:halo_aggregate)."""
    if shard is None:
        return ops.aggregate(h, graph, top_k, thr, unit)
    if shard.plan is not None:
        return sn_dist.halo_aggregate(h, shard.plan, graph, top_k, thr, table)
    return ops.aggregate(shard.table(h), graph, top_k, thr)


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

    def forward(self, x, edge_index, epilogue=None, act_in=None, head=None, fold=None):
        """``epilogue`` / ``act_in`` (both optional, the model wrappers' business): fuse the relu +
        dropout that follow this layer into its stores / tell ``lin`` that ``x`` is such an output
        (ops.HiddenEpilogue).  ``epilogue.applied`` says whether the layer did.  ``head`` (the LAST
        layer): the classification head inside the aggregation's launches (ops.HeadEpilogue; ``head.applied``).
        ``fold``: a ``LinFold`` - an evaluation-mode batch norm in front of this layer, folded into ``lin``."""
        graph, shard = _graph_for(x, edge_index, True, False)
        # (no selection: the aggregation scores straight from h - no unit rows wanted from lin)
        h, c, table = _lin_aligned(x, self.lin, shard, None, act_in, fold)
        if _fuse_head(head, graph, h, c, shard, None):
            return ops.aggregate(h, graph, None, 0.0, None, None, self.bias, head)
        if _fuse_epilogue(epilogue, h, c, shard):
            return ops.aggregate(h, graph, None, 0.0, None, epilogue, self.bias)
        out = _true_width(_aggregate(h, graph, shard, None, 0.0, table), c)
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
        graph, shard = _graph_for(x, edge_index, True, LOOPS_REPLACE)
        h, c, table = _lin_aligned(x, self.lin, shard)
        if shard is not None:
            h = shard.table(h, table)
        return _true_width(ops.attention(h, graph), c)


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

    def _filter_hint(self) -> _FilterHint:
        hint = getattr(self, "_filt_hint", None)
        if hint is None:
            hint = self._filt_hint = _FilterHint()
        return hint

    def forward(self, x, edge_index, epilogue=None, act_in=None, head=None, fold=None):
        """``epilogue`` / ``act_in`` / ``head`` / ``fold``: see SNConv.forward."""
        graph, shard = _graph_for(x, edge_index, True, bool(self.is_remove_self_loops))
        hint = self._filter_hint()
        hint.poll()
        unit = _unit_for(self.lin, graph, self.top_k, self.thr, hint)
        h, c, table = _lin_aligned(x, self.lin, shard, unit, act_in, fold)
        if shard is None and (unit.want_filter or unit.no_filter) and hint.due():
            hint.probe(h.detach(), edge_index, float(self.thr), ops.filter_wanted(graph, h.size(1), int(self.top_k), 0.0))
        if _fuse_head(head, graph, h, c, shard, int(self.top_k)):
            return ops.aggregate(h, graph, int(self.top_k), float(self.thr), unit, None, self.bias, head)
        if _fuse_epilogue(epilogue, h, c, shard):
            return ops.aggregate(h, graph, int(self.top_k), float(self.thr), unit, epilogue, self.bias)
        out = _true_width(_aggregate(h, graph, shard, int(self.top_k), float(self.thr), table, unit), c)
        if self.bias is not None:
            out = out + self.bias
        return out


class _AdjLinearParams(nn.Module):
    """Holds ``w.weight`` [C, N] and ``w.bias`` [C] under the reference's names
    (models.py:95).  The weight is stored column-major - ``weight.t()`` is a
    contiguous [N, C] table - so the adjacency branch gathers whole rows; its
    shape, ``state_dict`` key and initial values (same RNG stream as
    ``nn.Linear(num_nodes, C).reset_parameters()``) are the reference's.

    Built while a node-range partition is active (``sngnn_amd.dist.set_partition``), the
    module holds only the columns of the rank's own nodes, [C, n_local] - the same values
    the full initialisation gives those columns - and marks the parameter as sharded."""

    def __init__(self, num_nodes: int, out_channels: int):
        super().__init__()
        self.in_features, self.out_features = num_nodes, out_channels
        part = sn_dist.current_partition()
        self.shard_range = None
        if part is not None:
            if part.n_total != num_nodes:
                raise ValueError("num_nodes must be the partition's N_total")
            self.shard_range = (part.row_begin, part.row_end)
        n_cols = num_nodes if self.shard_range is None else part.n_local
        self.weight = Parameter(torch.empty(n_cols, out_channels).t())
        if self.shard_range is not None:
            sn_dist.mark_sharded(self.weight)
        self.bias = Parameter(torch.empty(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        w = torch.empty(self.out_features, self.in_features, device=self.weight.device)
        nn.init.kaiming_uniform_(w, a=math.sqrt(5))          # nn.Linear.reset_parameters
        with torch.no_grad():
            if self.shard_range is None:
                self.weight.copy_(w)
            else:
                self.weight.copy_(w[:, self.shard_range[0]:self.shard_range[1]])
        bound = 1 / math.sqrt(self.in_features) if self.in_features > 0 else 0
        nn.init.uniform_(self.bias, -bound, bound)

    # --- checkpoints of a sharded w (built under a partition: this rank holds [C, n_local]) ------
    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        """A reference-format checkpoint ([C, N] under ``...w.weight``) loads into a sharded
        module: the rank takes its own columns."""
        key = prefix + "weight"
        if self.shard_range is not None and key in state_dict:
            w = state_dict[key]
            if w.dim() == 2 and w.size(1) == self.in_features and self.in_features != self.weight.size(1):
                state_dict = dict(state_dict)
                state_dict[key] = w[:, self.shard_range[0]:self.shard_range[1]]
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def full_weight(self, part=None) -> torch.Tensor:
        """The reference-shaped [C, N] weight: the parameter itself when replicated, the
        all-gather of the ranks' column shards (a collective: every rank calls it) when sharded -
        what a checkpoint that must load into a single-process model, or into the reference,
        stores under ``w.weight`` (``sngnn_amd.dist.full_state_dict``)."""
        if self.shard_range is None:
            return self.weight.detach()
        part = part or sn_dist.current_partition()
        if part is None:
            raise ValueError("a sharded w needs its partition to be gathered")
        wt = self.weight.detach().t()
        rows = sn_dist.all_gather_rows(wt if wt.is_contiguous() else wt.contiguous(), part)
        return rows.t()

    def _apply(self, fn, recurse=True):
        # .to()/.cuda() would re-materialise the weight row-major; restore the layout
        super()._apply(fn, recurse)
        w = self.weight
        if w.dim() == 2 and not w.t().is_contiguous():
            with torch.no_grad():
                fixed = w.data.t().contiguous().t()
            self.weight = Parameter(fixed, requires_grad=w.requires_grad)
            if self.shard_range is not None:
                sn_dist.mark_sharded(self.weight)
        elif self.shard_range is not None:
            sn_dist.mark_sharded(self.weight)
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

    def _global_src_min(self, edge_index):
        """models.py:125's ``row - row.min()`` over the post-loop-handling edge list (cached per
        edge_index): 0 whenever loops are appended and kept."""
        key = (edge_index.data_ptr(), edge_index._version, tuple(edge_index.shape))
        hit = getattr(self, "_src_min_cache", None)
        if hit is None or hit[0] != key:
            if not self.is_remove_self_loops:
                m = 0
            else:
                keep = edge_index[0] != edge_index[1]
                m = edge_index[0][keep].min() if bool(keep.any()) else edge_index.new_tensor(2 ** 62)
                part = sn_dist.current_partition()
                if part is not None and torch.distributed.is_initialized():
                    # ranks may hold different subsets of the edges: the shift is a property of the
                    # WHOLE list, so every rank must arrive at the same value (and raise together)
                    m = m.clone()
                    if sn_dist._host_staged(m, part.group):
                        m = m.cpu()
                    torch.distributed.all_reduce(m, op=torch.distributed.ReduceOp.MIN, group=part.group)
                m = int(m)
                m = 0 if m >= 2 ** 62 else m
            hit = (key, m)
            self._src_min_cache = hit
        return hit[1]

    _filter_hint = SNConv_plus._filter_hint

    def forward(self, x, edge_index, epilogue=None, act_in=None, head=None, fold=None):
        """``epilogue`` / ``act_in`` / ``head`` / ``fold``: see SNConv.forward; here the BLEND is the layer's last
        kernel and takes the epilogue or the head (one GPU, no conv bias - i.e. no batch norm flag,
        models.py:52-53)."""
        if epilogue is not None:
            epilogue.applied = False
        if head is not None:
            head.applied = False
        part = sn_dist.current_partition()
        if part is None and x.size(0) != self.num_nodes:
            raise ValueError(f"built for {self.num_nodes} nodes, got {x.size(0)} "
                             "(the adjacency branch is Linear(num_nodes, C))")
        if part is None and self.w.shard_range is not None:
            raise ValueError("this layer holds a shard of w (built under a partition): run it under one")
        graph, shard = _graph_for(x, edge_index, True, bool(self.is_remove_self_loops))
        hint = self._filter_hint()
        hint.poll()
        unit = _unit_for(self.lin, graph, self.top_k, self.thr, hint)
        h, c, table = _lin_aligned(x, self.lin, shard, unit, act_in, fold)
        if shard is None and (unit.want_filter or unit.no_filter) and hint.due():
            hint.probe(h.detach(), edge_index, float(self.thr), ops.filter_wanted(graph, h.size(1), int(self.top_k), 0.0))
        if part is None:
            out_0 = ops.adj_linear(self.w.weight, self.w.bias, graph)
        else:
            # multi-GPU: edge_index must hold every edge incident to the owned nodes (the
            # full list is fine); a node's out-edges come from the partition of the FLIPPED list.
            if part.n_total != self.num_nodes:
                raise ValueError("num_nodes must be the partition's N_total")
            if self._global_src_min(edge_index) != 0:
                raise ValueError("the partitioned adjacency branch needs node 0 to have an out-edge "
                                 "(models.py:125's row shift would cross the node ranges otherwise)")
            flipped = self._flipped(edge_index)
            if self.w.shard_range is not None:
                # w sharded by node range: W^T rows travel like feature rows, over the flipped list
                if self.w.shard_range != (part.row_begin, part.row_end):
                    raise ValueError("w was sharded for a different node range")
                shard_f = _shard_for(flipped, part, True, bool(self.is_remove_self_loops))
                wt = self.w.weight.t()
                w_table = shard_f.table(wt if wt.is_contiguous() else wt.contiguous())
                out_0 = ops.gather_sum(w_table, self.w.bias, shard_f.graph)
            else:
                # w.weight replicated ([C, N_total]): its gradient is each rank's partial sum,
                # all-reduced with the other parameters (dist.allreduce_grads)
                graph_out = GLOBAL_CACHE.get(flipped, part.n_total, True, bool(self.is_remove_self_loops),
                                             row_range=(part.row_begin, part.row_end))
                out_0 = ops.adj_linear_partition(self.w.weight, self.w.bias, graph_out)
        out_1 = _true_width(_aggregate(h, graph, shard, int(self.top_k), float(self.thr), table, unit), c)
        if head is not None and part is None and self.bias is None:
            fused = ops.blend_head(out_0, out_1, self.beta, head)        # the head in the blend's own pass
            if fused is not None:
                return fused
        if epilogue is not None and part is None and self.bias is None:
            return ops.blend(out_0, out_1, self.beta, epilogue)
        out = ops.blend(out_0, out_1, self.beta)
        if self.bias is not None:
            out = out + self.bias
        return out
