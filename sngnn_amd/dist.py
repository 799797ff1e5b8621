"""Node-range partitioning over one process per GPU (RCCL over xGMI).

The reference is single-process / single-device; this is new design work
for graphs that exceed one GPU (BASELINE config 5).  Rank r owns the target nodes
[bounds[r], bounds[r + 1]) - any split of [0, N): equal rows (``Partition.even``) or equal
in-edges (``Partition.edge_balanced``) - i.e. its rows of x, of the CSR-by-target structure
and of the output.  One exchange per conv layer, two forms:

halo (default)   only the feature rows a rank's in-edges actually reference travel.
                 ``HaloPlan`` (built once per edge list) remaps the rank's columns to a
                 local table [own rows | halo rows, grouped by owner rank, ascending id];
                 forward = all-to-all-v of the requested rows, backward = its transpose
                 (the gradient of a halo row goes back to its owner and is added there in
                 rank order: deterministic).  RCCL: ``all_to_all_single`` with split sizes;
                 every peer is one xGMI hop away, so each pair's rows use their own link.
all-gather       the whole table on every rank (``all_gather_rows``): what the halo form
                 degenerates to on a graph without locality; kept as the simple baseline
                 and for the comparison bench.py prints.

SNGNN++'s ``Linear(num_nodes, C)`` is sharded by the same node ranges (the rank holds the
columns of ``w.weight`` of its own nodes): its rows are exchanged like feature rows over
the FLIPPED edge list, so neither the 460 MB table of products nor its dense gradient is
ever replicated or all-reduced.  Parameter gradients of the replicated parameters
(lin.weight / bias, conv bias, beta, batch-norm affine) are tiny and all-reduced by the
caller (``allreduce_grads``); batch statistics are reduced over the ranks (``sync_batch_norm``).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass
class Partition:
    """``Partition(rank, world, n_local)`` (equal shards) or ``Partition(rank, world,
    bounds=[0, ..., N])`` (any contiguous split)."""
    rank: int
    world: int
    n_local: Optional[int] = None
    group: Optional[object] = None
    bounds: Optional[Sequence[int]] = None
    exchange: str = "halo"              # "halo" | "allgather"

    def __post_init__(self):
        if self.bounds is None:
            if self.n_local is None:
                raise ValueError("give n_local (equal shards) or bounds")
            self.bounds = tuple(r * int(self.n_local) for r in range(self.world + 1))
        self.bounds = tuple(int(b) for b in self.bounds)
        if len(self.bounds) != self.world + 1 or self.bounds[0] != 0 or \
                any(a > b for a, b in zip(self.bounds, self.bounds[1:])):
            raise ValueError("bounds must be world + 1 non-decreasing offsets starting at 0")
        if not 0 <= self.rank < self.world:
            raise ValueError("rank out of range")
        if self.exchange not in ("halo", "allgather"):
            raise ValueError("exchange must be 'halo' or 'allgather'")
        self.n_local = self.bounds[self.rank + 1] - self.bounds[self.rank]

    @property
    def n_total(self) -> int:
        return self.bounds[-1]

    @property
    def row_begin(self) -> int:
        return self.bounds[self.rank]

    @property
    def row_end(self) -> int:
        return self.bounds[self.rank + 1]

    @property
    def sizes(self) -> List[int]:
        return [b - a for a, b in zip(self.bounds, self.bounds[1:])]

    @staticmethod
    def even_bounds(n_total: int, world: int) -> Tuple[int, ...]:
        """Rows split as evenly as possible (the first n_total % world ranks get one more)."""
        q, r = divmod(int(n_total), world)
        out = [0]
        for k in range(world):
            out.append(out[-1] + q + (1 if k < r else 0))
        return tuple(out)

    @staticmethod
    def edge_balanced_bounds(in_degree: torch.Tensor, world: int) -> Tuple[int, ...]:
        """Contiguous ranges with (nearly) equal numbers of in-edges + rows: rank k ends at
        the first node where the running cost reaches k / world of the total.  Cost of a
        node = its in-degree + 1 (the row itself is work too, and empty ranges are avoided)."""
        cost = in_degree.to(torch.int64).cpu() + 1
        run = torch.cumsum(cost, 0)
        total = int(run[-1]) if run.numel() else 0
        out = [0]
        for k in range(1, world):
            target = (total * k + world - 1) // world
            out.append(max(out[-1], int(torch.searchsorted(run, torch.tensor(target)))))
        out.append(int(in_degree.numel()))
        return tuple(out)

    @classmethod
    def even(cls, rank, world, n_total, **kw):
        return cls(rank, world, bounds=cls.even_bounds(n_total, world), **kw)

    @classmethod
    def edge_balanced(cls, rank, world, in_degree, **kw):
        return cls(rank, world, bounds=cls.edge_balanced_bounds(in_degree, world), **kw)


_current: Optional[Partition] = None


def set_partition(part: Optional[Partition]) -> None:
    """Make the conv layers treat their inputs as the local shard of ``part``
    (``None`` switches back to single-GPU behaviour)."""
    global _current
    _current = part


def current_partition() -> Optional[Partition]:
    return _current


def _host_staged(t: torch.Tensor, group=None) -> bool:
    """Whether the collective on ``t`` has to go through host copies: gloo moves host memory only,
    so a multi-rank REHEARSAL on a box with fewer GPUs than ranks (GPU tensors, gloo group) stages
    its buffers.  The collectives themselves are the same calls on every backend - RCCL on GPU
    tensors, gloo on CPU tensors (the world-2 CPU tests) and gloo on staged copies all run
    ``all_to_all_single`` / ``all_gather_into_tensor`` / ``reduce_scatter_tensor``: the CPU tests
    execute the branch the GPUs execute."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


# ---------------------------------------------------------------------------
# all-to-all-v of row blocks
# ---------------------------------------------------------------------------
def _all_to_all_rows(send: torch.Tensor, send_counts: Sequence[int], recv_counts: Sequence[int],
                     part: Partition) -> torch.Tensor:
    """``send`` = [sum(send_counts), C] row blocks ordered by destination rank; returns the
    [sum(recv_counts), C] blocks ordered by source rank."""
    c = send.size(1)
    staged = _host_staged(send, part.group)
    s = send.detach().contiguous()
    if staged:
        s = s.cpu()
    recv = s.new_empty((int(sum(recv_counts)), c))
    dist.all_to_all_single(recv, s, output_split_sizes=list(map(int, recv_counts)),
                           input_split_sizes=list(map(int, send_counts)), group=part.group)
    return recv.to(send.device) if staged else recv


def _exchange_counts(counts: torch.Tensor, part: Partition) -> torch.Tensor:
    """counts[p] = what I send to p  ->  what p sends to me (int64 [world])."""
    mine = counts.to(torch.int64)
    if _host_staged(mine, part.group):
        mine = mine.cpu()
    gathered = [torch.empty_like(mine) for _ in range(part.world)]
    dist.all_gather(gathered, mine, group=part.group)
    return torch.stack([g[part.rank] for g in gathered]).cpu()


class HaloPlan:
    """Which remote feature rows this rank's in-edges reference, and who wants which of its
    own.  Built once per (edge list, partition); the edge list it returns addresses the local
    table [own rows | halo rows] and keeps the original relative order of the edges (the
    reference's tie-break is the edge position)."""

    def __init__(self, edge_index: torch.Tensor, part: Partition):
        r0, r1 = part.row_begin, part.row_end
        dev = edge_index.device
        src, dst = edge_index[0], edge_index[1]
        mine = (dst >= r0) & (dst < r1)
        src_g = src[mine]
        own = (src_g >= r0) & (src_g < r1)
        need = torch.unique(src_g[~own])                       # sorted global ids: grouped by owner
        b = torch.tensor(part.bounds, dtype=torch.int64, device=dev)
        owner = torch.searchsorted(b, need, right=True) - 1
        need_counts = torch.bincount(owner, minlength=part.world)[:part.world]
        self.part = part
        self.n_local = part.n_local
        self.n_halo = int(need.numel())
        self.recv_counts = [int(v) for v in need_counts.tolist()]           # rows I receive per peer
        self.send_counts = [int(v) for v in _exchange_counts(need_counts, part).tolist()]
        # tell every owner which of its rows I need (ids relative to the owner's range)
        req = (need - b[owner]).to(torch.int64).view(-1, 1)
        got = _all_to_all_rows(req, self.recv_counts, self.send_counts, part)
        self.send_idx = got.view(-1).to(dev)                                   # my local rows, by peer
        if self.send_idx.numel() and (int(self.send_idx.min()) < 0 or int(self.send_idx.max()) >= self.n_local):
            raise RuntimeError("halo plan: a peer asked for a row outside this rank's range")
        # local edge list: targets -> [0, n_local), sources -> own id or n_local + halo position
        src_l = torch.where(own, src_g - r0, self.n_local + torch.searchsorted(need, src_g))
        self.edge_index = torch.stack([src_l, dst[mine] - r0]).contiguous()
        self.halo_ids = need
        # rows with at least one halo source ("boundary"): 1; rows whose sources are all local
        # ("interior"): 0 - the interior rows are aggregated while the halo is in flight
        flag = torch.zeros(self.n_local, dtype=torch.uint8, device=dev)
        flag[(dst[mine] - r0)[~own]] = 1
        self.row_boundary = flag
        self.n_boundary = int(flag.sum())

    @property
    def table_rows(self) -> int:
        return self.n_local + self.n_halo

    def exchanged_bytes(self, channels: int) -> int:
        """bytes this rank receives per forward exchange"""
        return self.n_halo * channels * 4


class _Pending:
    """An exchange in flight: ``wait()`` makes the current stream wait for it."""

    def __init__(self, work=None):
        self._work = work

    def wait(self):
        if self._work is not None:
            self._work.wait()
            self._work = None


def new_table(plan: HaloPlan, channels: int, like: torch.Tensor) -> torch.Tensor:
    """An uninitialised [own | halo] feature table for one forward (the caching allocator hands
    it out without any device work; a layer's ``lin`` writes its rows straight into the head)."""
    return like.new_empty((plan.table_rows, channels))


def start_halo_exchange(table: torch.Tensor, plan: HaloPlan) -> _Pending:
    """Send the own rows the peers asked for and receive this rank's halo rows STRAIGHT INTO
    ``table[n_local:]``; the own rows must already be in ``table[:n_local]``.  RCCL: one
    ``all_to_all_single`` issued asynchronously - kernels launched before ``wait()`` overlap it.
    (GPU tensors on a gloo group - rehearsals - are staged through the host and complete on return.)"""
    part = plan.part
    own = table[:plan.n_local]
    send = own.index_select(0, plan.send_idx)
    halo = table[plan.n_local:]
    if _host_staged(table, part.group):
        halo.copy_(_all_to_all_rows(send, plan.send_counts, plan.recv_counts, part))
        return _Pending()
    work = dist.all_to_all_single(halo, send, output_split_sizes=plan.recv_counts,
                                  input_split_sizes=plan.send_counts, group=part.group, async_op=True)
    return _Pending(work)


def return_halo_gradients(grad_table: torch.Tensor, plan: HaloPlan) -> torch.Tensor:
    """Transpose of the exchange: the gradient rows of the halo go back to their owners and are
    added to ``grad_table[:n_local]`` IN PLACE, peer by peer in rank order (inside one peer's
    block the rows are distinct: deterministic).  Returns that head view."""
    g_own = grad_table[:plan.n_local]
    back = _all_to_all_rows(grad_table[plan.n_local:].contiguous(), plan.recv_counts, plan.send_counts, plan.part)
    off = 0
    for n in plan.send_counts:
        if n:
            g_own.index_add_(0, plan.send_idx[off:off + n], back[off:off + n])
        off += n
    return g_own


class _HaloExchange(torch.autograd.Function):
    """[n_local, C] -> the [own | halo] table.  ``table`` (a ``_TableRef``) may already hold the
    own rows (``rows_local`` IS its head: the layer's ``lin`` wrote there) - nothing is copied
    then; otherwise they are copied in.  No concatenation either way."""

    @staticmethod
    def forward(ctx, rows_local, table_ref, plan: HaloPlan):
        ctx.plan = plan
        table = table_ref.t
        if rows_local.data_ptr() != table.data_ptr():
            table[:plan.n_local].copy_(rows_local)
        start_halo_exchange(table, plan).wait()
        return table

    @staticmethod
    def backward(ctx, grad_table):
        # (grad_table comes fresh out of the aggregation's backward: updated in place)
        return return_halo_gradients(grad_table.contiguous(), ctx.plan), None, None


class _TableRef:
    """Holder of a table tensor (kept out of autograd's sight as an input)."""

    def __init__(self, t: torch.Tensor):
        self.t = t


def halo_exchange(rows_local: torch.Tensor, plan: HaloPlan, table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Differentiable: [n_local, C] -> [n_local + n_halo, C] (own rows, then the halo).
    ``table``: the preallocated table whose head ``rows_local`` already is (``new_table``)."""
    if rows_local.size(0) != plan.n_local:
        raise ValueError(f"local shard has {rows_local.size(0)} rows, plan says {plan.n_local}")
    if table is None:
        table = new_table(plan, rows_local.size(1), rows_local)
    return _HaloExchange.apply(rows_local.contiguous(), _TableRef(table), plan)


# tests: a list - ``_HaloAggregate.forward`` then appends (label, HIP event recorded on the launch stream)
# at the four points that define the overlap: exchange issued, interior rows' kernels enqueued, exchange
# waited for, boundary rows' kernels enqueued (tests/test_dist_overlap_gpu.py)
TRACE = None


def _mark(label: str, t: torch.Tensor) -> None:
    if TRACE is not None and t.is_cuda:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream(t.device))
        TRACE.append((label, ev))


class _HaloAggregate(torch.autograd.Function):
    """Exchange + fused aggregation of one rank, overlapped: the rank's INTERIOR rows (every
    source local) are normalised and aggregated while the halo rows are on the wire, the
    BOUNDARY rows after they have arrived - two row-filtered launches of the same kernels on the
    same graph (``sngnn_agg_forward_rows``), bit-identical to one unfiltered call.  Backward: the
    aggregation's backward on the whole local table, then the transpose of the exchange."""

    @staticmethod
    def forward(ctx, rows_local, table_ref, plan: HaloPlan, graph, top_k, thr):
        from . import ops
        table = table_ref.t
        n_loc, c = plan.n_local, table.size(1)
        if rows_local.data_ptr() != table.data_ptr():
            table[:n_loc].copy_(rows_local)
        pending = start_halo_exchange(table, plan)
        _mark("exchange_issued", table)
        need_grad = ctx.needs_input_grad[0]
        dev = table.device
        out = torch.empty((n_loc, c), dtype=torch.float32, device=dev)
        wsel = torch.empty(graph.num_edges, dtype=torch.float32, device=dev) if need_grad else None
        inv = torch.empty(n_loc, dtype=torch.float32, device=dev) if need_grad else None
        on_the_fly = top_k is None          # nothing is selected: scored straight from h (as sngnn_agg_forward does)
        if on_the_fly:
            unit, nrm, filt = table, None, None
        else:
            unit = torch.empty_like(table)
            nrm = torch.empty(table.size(0), dtype=torch.float32, device=dev)
            fb = ops.filter_row_bytes(c) if ops.filter_wanted(graph, c, top_k, thr) else 0
            filt = torch.empty((table.size(0), fb), dtype=torch.uint8, device=dev) if fb else None
            ops.normalize_rows_into(table[:n_loc], unit[:n_loc], nrm[:n_loc], None if filt is None else filt[:n_loc])
        if plan.n_boundary < n_loc:
            ops.aggregate_forward_rows(graph, unit, nrm, filt, top_k, thr, plan.row_boundary, 0, out, wsel, inv)
        _mark("interior_enqueued", table)
        pending.wait()
        _mark("exchange_waited", table)
        if plan.n_halo and not on_the_fly:
            ops.normalize_rows_into(table[n_loc:], unit[n_loc:], nrm[n_loc:], None if filt is None else filt[n_loc:])
        if plan.n_boundary:
            ops.aggregate_forward_rows(graph, unit, nrm, filt, top_k, thr, plan.row_boundary, 1, out, wsel, inv)
        _mark("boundary_enqueued", table)
        if need_grad:
            ctx.plan, ctx.graph, ctx.top_k = plan, graph, top_k
            ctx.save_for_backward(table, wsel)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        from . import ops
        table, wsel = ctx.saved_tensors
        grad_table = ops.aggregate_backward(ctx.graph, table, grad_out.contiguous(), wsel, ctx.top_k)
        return return_halo_gradients(grad_table, ctx.plan), None, None, None, None, None


def halo_aggregate(rows_local: torch.Tensor, plan: HaloPlan, graph, top_k, thr: float,
                   table: Optional[torch.Tensor] = None) -> torch.Tensor:
    """One rank's conv after ``lin``: exchange the halo, aggregate the owned rows - the interior
    rows while the halo is in flight.  ``table``: see ``halo_exchange``.  [n_local, C] -> [n_local, C]."""
    if rows_local.size(0) != plan.n_local:
        raise ValueError(f"local shard has {rows_local.size(0)} rows, plan says {plan.n_local}")
    if table is None:
        table = new_table(plan, rows_local.size(1), rows_local)
    return _HaloAggregate.apply(rows_local.contiguous(), _TableRef(table), plan, graph, top_k, float(thr))


# ---------------------------------------------------------------------------
# full all-gather (baseline)
# ---------------------------------------------------------------------------
class _AllGatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h_local, part: Partition):
        ctx.part = part
        h_local = h_local.contiguous()
        sizes, c = part.sizes, h_local.size(1)
        staged = _host_staged(h_local, part.group)
        src = h_local.cpu() if staged else h_local
        if len(set(sizes)) > 1:
            # uneven shards: pad to the largest shard, gather, drop the padding
            mx = max(sizes)
            pad = src.new_zeros((mx, c))
            pad[:part.n_local] = src
            packed = src.new_empty((part.world * mx, c))
            dist.all_gather_into_tensor(packed, pad, group=part.group)
            full = torch.cat([packed[r * mx:r * mx + n] for r, n in enumerate(sizes)], dim=0)
        else:
            full = src.new_empty((part.n_total, c))
            dist.all_gather_into_tensor(full, src, group=part.group)
        return full.to(h_local.device) if staged else full

    @staticmethod
    def backward(ctx, grad_full):
        part = ctx.part
        staged = _host_staged(grad_full, part.group)
        g = grad_full.contiguous().cpu() if staged else grad_full.contiguous()
        sizes, c = part.sizes, g.size(1)
        if len(set(sizes)) > 1:
            # uneven shards: every rank's block padded to the largest shard, then the same reduce-scatter
            mx = max(sizes)
            packed = g.new_zeros((part.world * mx, c))
            off = 0
            for r, n in enumerate(sizes):
                packed[r * mx:r * mx + n] = g[off:off + n]
                off += n
            red = g.new_empty((mx, c))
            dist.reduce_scatter_tensor(red, packed, group=part.group)
            out = red[:part.n_local].clone()
        else:
            out = g.new_empty((part.n_local, c))
            dist.reduce_scatter_tensor(out, g, group=part.group)
        return (out.to(grad_full.device) if staged else out), None


def all_gather_rows(h_local: torch.Tensor, part: Partition) -> torch.Tensor:
    """Differentiable all-gather of the row shards: [n_local, C] -> [n_total, C]."""
    if h_local.size(0) != part.n_local:
        raise ValueError(f"local shard has {h_local.size(0)} rows, partition says {part.n_local}")
    return _AllGatherRows.apply(h_local, part)


# ---------------------------------------------------------------------------
# parameters and batch statistics
# ---------------------------------------------------------------------------
def mark_sharded(p: torch.nn.Parameter) -> torch.nn.Parameter:
    """A parameter that is split over the ranks (SNGNN++'s w.weight): its gradient is
    already complete on its owner and must not be all-reduced."""
    p._sngnn_sharded = True
    return p


def allreduce_grads(module: torch.nn.Module, part: Partition) -> None:
    """Sum the REPLICATED parameters' gradients over the ranks."""
    for p in module.parameters():
        g = p.grad
        if g is None or getattr(p, "_sngnn_sharded", False):
            continue
        if g.is_contiguous():
            dist.all_reduce(g, group=part.group)
        elif g.dim() == 2 and g.t().is_contiguous():      # SNGNN++'s column-major w.weight (replicated form)
            dist.all_reduce(g.t(), group=part.group)
        else:
            tmp = g.contiguous()
            dist.all_reduce(tmp, group=part.group)
            g.copy_(tmp)


class _SyncBatchNorm(torch.autograd.Function):
    """nn.BatchNorm1d's training-mode forward over the rows of ALL ranks: per-channel sum, sum
    of squares and row count are all-reduced (one [2C + 1] message), so the statistics, the
    running estimates and the gradients are those of the single-process batch."""

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, part: Partition):
        c = x.size(1)
        stats = torch.cat([x.sum(0), (x * x).sum(0), x.new_tensor([float(x.size(0))])])
        dist.all_reduce(stats, group=part.group)
        n = stats[-1]
        mean = stats[:c] / n
        var = (stats[c:2 * c] / n - mean * mean).clamp_min_(0)            # biased, as BN normalises
        invstd = torch.rsqrt(var + eps)
        xhat = (x - mean) * invstd
        if running_mean is not None:
            with torch.no_grad():
                unbiased = var * (n / (n - 1).clamp_min(1))
                running_mean.mul_(1 - momentum).add_(mean, alpha=momentum)
                running_var.mul_(1 - momentum).add_(unbiased, alpha=momentum)
        ctx.save_for_backward(xhat, invstd, weight)
        ctx.part, ctx.n = part, n
        out = xhat if weight is None else xhat * weight
        return out if bias is None else out + bias

    @staticmethod
    def backward(ctx, g):
        xhat, invstd, weight = ctx.saved_tensors
        gx = g if weight is None else g * weight
        c = g.size(1)
        sums = torch.cat([gx.sum(0), (gx * xhat).sum(0)])
        dist.all_reduce(sums, group=ctx.part.group)
        dx = (gx - sums[:c] / ctx.n - xhat * (sums[c:] / ctx.n)) * invstd
        # weight / bias gradients are this rank's partial sums: allreduce_grads completes them
        gw = (g * xhat).sum(0) if weight is not None else None
        gb = g.sum(0)
        return dx, gw, gb, None, None, None, None, None


def sync_batch_norm(bn: torch.nn.BatchNorm1d, x: torch.Tensor, part: Partition) -> torch.Tensor:
    """``bn(x)`` with batch statistics over every rank's rows (training mode); eval mode uses
    the running estimates and needs no communication."""
    if not bn.training:
        return bn(x)
    if bn.momentum is None:
        raise NotImplementedError("cumulative-average batch norm (momentum=None) under a partition")
    if bn.num_batches_tracked is not None:
        bn.num_batches_tracked += 1
    return _SyncBatchNorm.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps,
                                bn.momentum, part)


def full_state_dict(module: torch.nn.Module, part: Optional[Partition] = None) -> dict:
    """``module.state_dict()`` in the REFERENCE's layout: every sharded ``w.weight`` ([C, n_local]
    on this rank) replaced by the gathered [C, N] table, so the checkpoint loads into a
    single-process model (or the reference) - and, through ``_AdjLinearParams``'s load hook, back
    into a model sharded over any partition.  A collective when something is sharded: every rank
    calls it; write the result from one."""
    sd = dict(module.state_dict())
    for name, sub in module.named_modules():
        if hasattr(sub, "full_weight") and getattr(sub, "shard_range", None) is not None:
            sd[(name + "." if name else "") + "weight"] = sub.full_weight(part).contiguous()
    return sd
