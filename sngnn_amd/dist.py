"""Node-range partitioning over one process per GPU (RCCL over xGMI).

The reference is single-process / single-device; this is new
design work for graphs that exceed one GPU.  Rank r owns the target nodes
[r * n_local, (r + 1) * n_local): its rows of x, of the CSR-by-target structure
(column ids stay global) and of the output.  One exchange per conv layer:

    forward   h_full = all_gather(h_local)            [P * n_local, C]
    backward  grad_h_local = reduce_scatter(grad_h_full partials, sum)

Every peer is one xGMI hop away, so RCCL's all-gather moves each shard over its
own link; shard sizes here are n_local * C * 4 bytes.  Parameter gradients
(lin.weight / bias, conv bias) are tiny and all-reduced by the caller
(``allreduce_grads``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch
import torch.distributed as dist


@dataclass
class Partition:
    rank: int
    world: int
    n_local: int
    group: Optional[object] = None

    @property
    def n_total(self) -> int:
        return self.n_local * self.world

    @property
    def row_begin(self) -> int:
        return self.rank * self.n_local

    @property
    def row_end(self) -> int:
        return (self.rank + 1) * self.n_local


_current: Optional[Partition] = None


def set_partition(part: Optional[Partition]) -> None:
    """Make the conv layers treat their inputs as the local shard of ``part``
    (``None`` switches back to single-GPU behaviour)."""
    global _current
    _current = part


def current_partition() -> Optional[Partition]:
    return _current


def _use_fallback(t: torch.Tensor, group=None) -> bool:
    # gloo (CPU tests, and multi-rank rehearsals on a box with fewer GPUs than ranks) has no
    # reduce_scatter: list all-gather + all-reduce instead
    return (not t.is_cuda) or dist.get_backend(group) == "gloo"


class _AllGatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h_local, part: Partition):
        ctx.part = part
        h_local = h_local.contiguous()
        full = h_local.new_empty((part.n_total, h_local.size(1)))
        if _use_fallback(h_local, part.group):
            chunks = list(full.chunk(part.world, dim=0))
            dist.all_gather(chunks, h_local, group=part.group)
        else:
            dist.all_gather_into_tensor(full, h_local, group=part.group)
        return full

    @staticmethod
    def backward(ctx, grad_full):
        part = ctx.part
        grad_full = grad_full.contiguous()
        if _use_fallback(grad_full, part.group):
            dist.all_reduce(grad_full, group=part.group)
            return grad_full[part.row_begin:part.row_end].clone(), None
        out = grad_full.new_empty((part.n_local, grad_full.size(1)))
        dist.reduce_scatter_tensor(out, grad_full, group=part.group)
        return out, None


def all_gather_rows(h_local: torch.Tensor, part: Partition) -> torch.Tensor:
    """Differentiable all-gather of equal row shards: [n_local, C] -> [n_total, C]."""
    if h_local.size(0) != part.n_local:
        raise ValueError(f"local shard has {h_local.size(0)} rows, partition says {part.n_local}")
    return _AllGatherRows.apply(h_local, part)


def allreduce_grads(module: torch.nn.Module, part: Partition) -> None:
    """Sum the (replicated) parameters' gradients over the ranks."""
    for p in module.parameters():
        g = p.grad
        if g is None:
            continue
        if g.is_contiguous():
            dist.all_reduce(g, group=part.group)
        elif g.dim() == 2 and g.t().is_contiguous():      # SNGNN++'s column-major w.weight
            dist.all_reduce(g.t(), group=part.group)
        else:
            tmp = g.contiguous()
            dist.all_reduce(tmp, group=part.group)
            g.copy_(tmp)
