"""Operator-level Python entry points over the C ABI (include/sngnn_hip.h).

``aggregate`` is the fused replacement of everything the reference's conv layers
do after ``self.lin``: F.normalize, the per-edge cosine, the top-k / threshold
loop over torch_scatter.scatter_max and the mean aggregation
(models/models.py:122+132+139-158, :238-239+244-263, :325-326+331-334).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from .graph import Graph


def _check_rows(t: torch.Tensor, n: int, what: str) -> torch.Tensor:
    if t.dtype != torch.float32:
        raise ValueError(f"{what} must be float32 (the reference path is fp32 only)")
    if not t.is_cuda:
        raise ValueError(f"{what} must live on the GPU (there is no CPU path)")
    if t.dim() != 2 or t.size(0) != n:
        raise ValueError(f"{what} must have shape [{n}, C], got {tuple(t.shape)}")
    if not 1 <= t.size(1) <= _lib.MAX_CHANNELS:
        raise ValueError(f"{what}: C must be in [1, {_lib.MAX_CHANNELS}]")
    return t.contiguous()


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def aggregate_forward(graph: Graph, h: torch.Tensor, top_k: Optional[int], thr: float, *,
                      save_for_backward: bool = False, want_selection: bool = False):
    """Returns (out, wsel, inv_norm, sel_src, sel_w); the optional ones are None
    unless requested.  ``top_k=None`` is SNConv (no selection)."""
    lib = _lib.load()
    n = graph.num_nodes
    h = _check_rows(h, graph.num_total_nodes, "h")
    c = h.size(1)
    k = -1 if top_k is None else int(top_k)
    if top_k is not None and k < 0:
        raise ValueError("top_k must be >= 0")
    out = torch.empty((n, c), dtype=torch.float32, device=h.device)
    wsel = inv = sel_src = sel_w = None
    if save_for_backward:
        wsel = torch.empty(graph.num_edges, dtype=torch.float32, device=h.device)
        inv = torch.empty(n, dtype=torch.float32, device=h.device)
    if want_selection:
        if k < 0:
            raise ValueError("the selection is only defined for top_k >= 0")
        sel_src = torch.empty((n, k), dtype=torch.int32, device=h.device)
        sel_w = torch.empty((n, k), dtype=torch.float32, device=h.device)
    ws = graph.workspace(c)
    with torch.cuda.device(h.device):
        rc = lib.sngnn_agg_forward(graph.handle, h.data_ptr(), c, k, float(thr), out.data_ptr(),
                                   _lib.ptr(wsel), _lib.ptr(inv), _lib.ptr(sel_src),
                                   _lib.ptr(sel_w), ws.data_ptr(), _stream(h.device))
    _lib.check(rc, "sngnn_agg_forward")
    return out, wsel, inv, sel_src, sel_w


def normalize_rows(h: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """``sngnn_normalize_rows``: F.normalize(h, p=2, dim=-1) (models.py:122,238,325) with IEEE
    square root and division.  Returns (unit rows [rows, C], clamped norms [rows])."""
    h = _check_rows(h, h.size(0), "h")
    n = torch.empty_like(h)
    nrm = torch.empty(h.size(0), dtype=torch.float32, device=h.device)
    with torch.cuda.device(h.device):
        rc = _lib.load().sngnn_normalize_rows(h.data_ptr(), h.size(0), h.size(1), n.data_ptr(),
                                              nrm.data_ptr(), _stream(h.device))
    _lib.check(rc, "sngnn_normalize_rows")
    return n, nrm


def filter_row_bytes(c: int) -> int:
    """``sngnn_filter_row_bytes``: bytes of one fp16 filter row for C channels, 0 = no filter."""
    return int(_lib.load().sngnn_filter_row_bytes(int(c)))


def normalize_rows_filter(h: torch.Tensor):
    """``sngnn_normalize_rows_filter``: (unit rows, clamped norms, filter rows or None)."""
    h = _check_rows(h, h.size(0), "h")
    n = torch.empty_like(h)
    nrm = torch.empty(h.size(0), dtype=torch.float32, device=h.device)
    fb = filter_row_bytes(h.size(1))
    filt = torch.empty((h.size(0), fb), dtype=torch.uint8, device=h.device) if fb else None
    with torch.cuda.device(h.device):
        rc = _lib.load().sngnn_normalize_rows_filter(h.data_ptr(), h.size(0), h.size(1), n.data_ptr(),
                                                     nrm.data_ptr(), _lib.ptr(filt), _stream(h.device))
    _lib.check(rc, "sngnn_normalize_rows_filter")
    return n, nrm, filt


def filter_wanted(graph: Graph, c: int, top_k: Optional[int], thr: float) -> bool:
    """``sngnn_filter_wanted``: whether a forward with these arguments uses filter rows."""
    k = -1 if top_k is None else int(top_k)
    return bool(_lib.load().sngnn_filter_wanted(graph.handle, int(c), k, float(thr)))


def normalize_rows_into(h: torch.Tensor, n: torch.Tensor, nrm: torch.Tensor, filt: Optional[torch.Tensor]) -> None:
    """``sngnn_normalize_rows_filter`` into caller-owned (slices of) buffers: ``n`` [rows, C] like
    ``h``, ``nrm`` [rows], ``filt`` uint8 [rows, filter_row_bytes(C)] or None."""
    rows, c = h.shape
    if rows == 0:
        return
    if not (h.is_contiguous() and n.is_contiguous() and nrm.is_contiguous() and (filt is None or filt.is_contiguous())):
        raise ValueError("normalize_rows_into needs contiguous row blocks")
    if n.shape != h.shape or nrm.numel() != rows or (filt is not None and filt.size(0) != rows):
        raise ValueError("normalize_rows_into: buffer shapes do not match h")
    with torch.cuda.device(h.device):
        rc = _lib.load().sngnn_normalize_rows_filter(h.data_ptr(), rows, c, n.data_ptr(), nrm.data_ptr(),
                                                     _lib.ptr(filt), _stream(h.device))
    _lib.check(rc, "sngnn_normalize_rows_filter")


def aggregate_forward_rows(graph: Graph, n: torch.Tensor, nrm: torch.Tensor, filt: Optional[torch.Tensor],
                           top_k: Optional[int], thr: float, row_flag: torch.Tensor, want: int,
                           out: torch.Tensor, wsel: Optional[torch.Tensor] = None,
                           inv: Optional[torch.Tensor] = None) -> None:
    """``sngnn_agg_forward_rows``: the aggregation of the target rows whose ``row_flag`` (uint8 [N])
    equals ``want``, written into the caller's ``out`` / ``wsel`` / ``inv``; other rows untouched.
    ``nrm=None``: ``n`` holds the RAW rows h and the call scores on the fly (what
    ``sngnn_agg_forward`` does by itself when nothing is selected, top_k None)."""
    c = n.size(1)
    k = -1 if top_k is None else int(top_k)
    if row_flag.dtype != torch.uint8 or row_flag.numel() != graph.num_nodes or not row_flag.is_contiguous():
        raise ValueError("row_flag must be a contiguous uint8 tensor with one entry per owned row")
    if n.size(0) != graph.num_total_nodes or out.shape != (graph.num_nodes, c):
        raise ValueError("n must hold one row per feature-table row and out one per owned row")
    ws = graph.workspace(c)
    with torch.cuda.device(n.device):
        rc = _lib.load().sngnn_agg_forward_rows(graph.handle, n.data_ptr(), _lib.ptr(nrm), _lib.ptr(filt), c, k,
                                                float(thr), row_flag.data_ptr(), int(want), out.data_ptr(),
                                                _lib.ptr(wsel), _lib.ptr(inv), ws.data_ptr(), _stream(n.device))
    _lib.check(rc, "sngnn_agg_forward_rows")


def aggregate_forward_normalized(graph: Graph, n: torch.Tensor, nrm: torch.Tensor,
                                 top_k: Optional[int], thr: float, *, want_selection: bool = False,
                                 filt: Optional[torch.Tensor] = None):
    """``sngnn_agg_forward_prepared``: the aggregation on unit rows + norms (+ filter rows)
    that the caller already holds.  Returns (out, sel_src, sel_w)."""
    lib = _lib.load()
    n = _check_rows(n, graph.num_total_nodes, "n")
    if nrm.dtype != torch.float32 or nrm.numel() != graph.num_total_nodes or not nrm.is_cuda:
        raise ValueError("nrm must be a float32 GPU tensor with one entry per feature row")
    nrm = nrm.contiguous()
    c = n.size(1)
    k = -1 if top_k is None else int(top_k)
    out = torch.empty((graph.num_nodes, c), dtype=torch.float32, device=n.device)
    sel_src = sel_w = None
    if want_selection:
        sel_src = torch.empty((graph.num_nodes, k), dtype=torch.int32, device=n.device)
        sel_w = torch.empty((graph.num_nodes, k), dtype=torch.float32, device=n.device)
    ws = graph.workspace(c)
    with torch.cuda.device(n.device):
        rc = lib.sngnn_agg_forward_prepared(graph.handle, n.data_ptr(), nrm.data_ptr(), _lib.ptr(filt), c, k,
                                            float(thr), out.data_ptr(), None, None, _lib.ptr(sel_src),
                                            _lib.ptr(sel_w), ws.data_ptr(), _stream(n.device))
    _lib.check(rc, "sngnn_agg_forward_prepared")
    return out, sel_src, sel_w


def aggregate_backward(graph: Graph, h: torch.Tensor, grad_out: torch.Tensor,
                       wsel: torch.Tensor, top_k: Optional[int] = None) -> torch.Tensor:
    """``top_k``: the forward's top_k (a bound on the kept in-edges per row; lets the library
    take every node in one launch) or None when unknown / nothing was selected."""
    lib = _lib.load()
    h = _check_rows(h, graph.num_total_nodes, "h")
    grad_out = _check_rows(grad_out, graph.num_nodes, "grad_out")
    c = h.size(1)
    grad_h = torch.empty_like(h)
    ws = graph.workspace(c)
    with torch.cuda.device(h.device):
        rc = lib.sngnn_agg_backward_topk(graph.handle, h.data_ptr(), c, grad_out.data_ptr(),
                                         wsel.data_ptr(), -1 if top_k is None else int(top_k),
                                         grad_h.data_ptr(), ws.data_ptr(), _stream(h.device))
    _lib.check(rc, "sngnn_agg_backward_topk")
    return grad_h


def kept_bits_supported(graph: Graph, top_k: Optional[int], channels: int = 64) -> bool:
    """``sngnn_agg_kept_bits_supported``: whether a training forward on this graph with this top_k
    (at this width) can hand its backward the kept edges as bits it packs itself."""
    if top_k is None:
        return False
    return bool(_lib.load().sngnn_agg_kept_bits_supported(graph.handle, int(channels), int(top_k)))


def aggregate_backward_bits(graph: Graph, h: torch.Tensor, grad_out: torch.Tensor, kept_bits: torch.Tensor,
                            top_k: int) -> torch.Tensor:
    """``sngnn_agg_backward_bits``: the backward from the kept bits the forward wrote."""
    lib = _lib.load()
    h = _check_rows(h, graph.num_total_nodes, "h")
    grad_out = _check_rows(grad_out, graph.num_nodes, "grad_out")
    c = h.size(1)
    grad_h = torch.empty_like(h)
    with torch.cuda.device(h.device):
        rc = lib.sngnn_agg_backward_bits(graph.handle, h.data_ptr(), c, grad_out.data_ptr(), kept_bits.data_ptr(),
                                         int(top_k), grad_h.data_ptr(), graph.workspace(c).data_ptr(), _stream(h.device))
    _lib.check(rc, "sngnn_agg_backward_bits")
    return grad_h


class HiddenEpilogue:
    """What follows a hidden conv layer in the reference's wrappers (models.py:204-209, 79-84,
    296-301) - the conv's bias add, ``F.relu(x, inplace=True)`` and ``self.dropout(x)`` - as the
    aggregation's STORE epilogue (``sngnn_agg_forward_epilogue``) instead of three more passes over
    [N, C], and, backward, as the epilogue of the store that produces the gradient of the
    activated tensor (``sngnn_linear_forward_masked``: the next layer's ``lin`` input gradient)
    instead of threshold_backward + the dropout backward.

    ``p`` / ``training``: the wrapper's Dropout.  ``applied``: set by the conv layer when its shape
    took the fused store (the caller falls back to the plain elementwise ops otherwise).
    ``premasked``: set by the consumer's backward (to the address of the gradient tensor it returns)
    when it has already applied the mask to that gradient - the contract between ``_Linear.backward``
    and ``_Aggregate.backward`` / ``_Blend.backward``.  The wrapper hands the activated tensor to exactly
    one consumer (the next conv's ``lin``; models.py:205 -> :237); the producer's backward CHECKS that the
    gradient it receives is that very tensor (``_take_premasked``) and raises otherwise.  A hook on a
    hidden activation therefore sees the gradient of the PRE-activation (already masked)."""

    def __init__(self, relu: bool = True, p: float = 0.0, training: bool = False, seed: Optional[torch.Tensor] = None):
        """``seed``: int64 [1] device tensor - the dropout mask is then drawn inside the kernel from
        (seed, element index) by a counter-based hash: no mask tensor, no extra launch; the owner
        advances the seed between forwards (models._Stack does, on the device: graph-capture
        safe).  Without it the mask is torch's own Bernoulli draw, handed to the kernel."""
        self.relu, self.p, self.training = bool(relu), float(p), bool(training)
        self.seed = seed
        self.applied = False
        self.premasked = False
        self.scale = 1.0

    @property
    def drops(self) -> bool:
        return self.training and self.p > 0.0


def _take_premasked(epi: "HiddenEpilogue", grad_out: torch.Tensor) -> None:
    """The producer's side of the ``premasked`` hand-over: the gradient that arrives must be THE tensor the
    consumer's masked store wrote.  Anything else means the activated tensor had a second consumer (a hook,
    a skip connection, ``autograd.grad`` on it) whose contribution autograd added to the pre-masked one -
    the sum can no longer be masked correctly, so refuse loudly instead of returning a wrong gradient."""
    ptr, epi.premasked = epi.premasked, False
    if grad_out.data_ptr() != ptr:
        raise RuntimeError("the fused hidden epilogue (ops.HiddenEpilogue) needs the activated tensor to have exactly "
                           "one consumer - the next layer's lin; its gradient arrived accumulated with another "
                           "consumer's.  Set sngnn_amd.models.FUSE_HIDDEN = False for such a model.")


class HeadEpilogue:
    """What follows the LAST conv layer - the wrappers' ``log_softmax`` (models.py:86,211,303) and the
    harness' ``nll_loss`` on a mask + accuracy count (train.py:81-84, 98-102, 112-116) - inside the
    aggregation's own launches (``sngnn_epilogue_t.head_*``): the split rows go through the head in
    their finalize, all other rows are read back by extra workgroups of that same launch - a latency
    chain on an otherwise idle chip - instead of two more launches behind it (``head_nll`` /
    ``head_nll2``, which compute the same per-row bits).

    ``y`` int64 [N]; ``sel`` uint8 [N] - one split: != 0 marks its rows; two splits (``n_b`` given):
    bit 0 = split A, bit 1 = split B; ``metrics`` float32 [2] / [4] receives (mean NLL, correct
    count) per split.  ``grad``: one split, training - the layer's output tensor then holds
    d loss / d logits (zero rows outside the split) and ``G.backward(G.detach())`` is the backward
    of ``loss.backward()``; otherwise the output holds the logits.
    ``applied``: set by the conv layer when its shape took the fused form."""

    def __init__(self, y, sel, metrics, n_a: int, n_b: Optional[int] = None, grad: bool = False):
        if y.dtype != torch.int64 or sel.dtype != torch.uint8 or metrics.dtype != torch.float32:
            raise ValueError("y must be int64, sel uint8, metrics float32")
        sets = 1 if n_b is None else 2
        if metrics.numel() != 2 * sets or not metrics.is_contiguous():
            raise ValueError("metrics must be a contiguous float32 tensor of 2 elements per split")
        if grad and sets != 1:
            raise ValueError("grad: one split only")
        self.y, self.sel, self.metrics = y.contiguous(), sel.contiguous(), metrics
        self.sets, self.n_a, self.n_b = sets, int(n_a), int(n_b or 0)
        self.grad = bool(grad)
        self.applied = False

    @property
    def out_mode(self) -> int:
        return 2 if self.grad else 1


def head_supported(graph: Graph, channels: int, top_k: Optional[int]) -> bool:
    """``sngnn_agg_head_supported``: whether a forward on this graph at this width can take the head."""
    return bool(_lib.load().sngnn_agg_head_supported(graph.handle, int(channels), -1 if top_k is None else int(top_k)))


class _Aggregate(torch.autograd.Function):
    """autograd seam of the fused aggregation.  The output is freshly allocated
    and not saved, so the models' in-place ReLU on it is safe (models.py:81,206,298).
    With ``epi`` (a HiddenEpilogue) the stored rows are already bias-added / rectified /
    dropped out; the output is then saved, as the mask its own backward needs."""

    @staticmethod
    def forward(ctx, h, graph, top_k, thr, unit=None, epi=None, bias=None, head=None):
        need_grad = ctx.needs_input_grad[0]
        ctx.epi = epi
        # training calls: where the library can, the forward writes WHICH edges it kept as packed bits
        # itself (sngnn_epilogue_t.kept_bits) - no per-edge weights, no packing launch in the backward
        ctx.bits = need_grad and kept_bits_supported(graph, top_k, h.size(1))
        ctx.bias_grad = False
        if head is not None:
            # the classification head inside the forward's launches (HeadEpilogue): ``out`` = the gradient of
            # the split's mean NLL (training) or the logits
            out, wsel = _forward_epilogue(graph, h, unit, top_k, thr, need_grad, None, bias, ctx.bits, head)
            ctx.bias_grad = bias is not None and ctx.needs_input_grad[6]
        elif epi is not None or ctx.bits or (unit is not None and unit.no_filter):
            out, wsel = _forward_epilogue(graph, h, unit, top_k, thr, need_grad, epi, bias, ctx.bits)
            ctx.bias_grad = epi is not None and bias is not None and ctx.needs_input_grad[6]
        elif unit is not None and unit.n is not None:
            # the producer of h (``lin``'s epilogue) already wrote F.normalize(h): no pass over h
            out, wsel = _forward_prepared(graph, unit, top_k, thr, need_grad)
        else:
            out, wsel, _, _, _ = aggregate_forward(graph, h, top_k, thr,
                                                   save_for_backward=need_grad)
        if need_grad:
            ctx.graph, ctx.top_k = graph, top_k
            if epi is not None:
                ctx.save_for_backward(h, wsel, out)
            else:
                ctx.save_for_backward(h, wsel)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        epi = ctx.epi
        grad_bias = None
        if epi is None:
            h, wsel = ctx.saved_tensors
        else:
            h, wsel, out = ctx.saved_tensors
            if epi.premasked:          # the consumer's store already applied relu' and the dropout mask
                _take_premasked(epi, grad_out)
            else:
                g = grad_out.contiguous()
                grad_out = torch.empty_like(g)
                with torch.cuda.device(g.device):
                    rc = _lib.load().sngnn_epilogue_backward(g.data_ptr(), out.data_ptr(), float(epi.scale), g.numel(),
                                                             grad_out.data_ptr(), _stream(g.device))
                _lib.check(rc, "sngnn_epilogue_backward")
            if ctx.bias_grad:
                grad_bias = grad_out.sum(dim=0)
        if epi is None and ctx.bias_grad:          # (the head epilogue added the conv's bias)
            grad_bias = grad_out.sum(dim=0)
        if ctx.bits:         # (``wsel`` holds the kept bits)
            grad_h = aggregate_backward_bits(ctx.graph, h, grad_out.contiguous(), wsel, ctx.top_k)
        else:
            grad_h = aggregate_backward(ctx.graph, h, grad_out.contiguous(), wsel, ctx.top_k)
        return grad_h, None, None, None, None, None, grad_bias, None


class UnitRows:
    """What ``lin``'s normalising epilogue leaves behind for the aggregation that follows:
    ``n`` [N, C] unit rows, ``nrm`` [N] clamped norms, ``filt`` fp16 filter rows or None
    (``sngnn_linear_forward_normalized``).  ``want_filter``: whether the consumer will use them
    (``ops.filter_wanted``).  ``n`` stays None when the layer's shape took another route."""

    def __init__(self, want_filter: bool = False, no_filter: bool = False):
        self.want_filter = bool(want_filter) and not no_filter
        # the caller knows its rows do not prune (conv._FilterHint): the forward is told not to build
        # filter rows of its own either (sngnn_epilogue_t.no_filter)
        self.no_filter = bool(no_filter)
        self.n = self.nrm = self.filt = None


def _forward_prepared(graph: Graph, unit: "UnitRows", top_k, thr, need_grad):
    lib = _lib.load()
    n = unit.n
    c = n.size(1)
    k = -1 if top_k is None else int(top_k)
    out = torch.empty((graph.num_nodes, c), dtype=torch.float32, device=n.device)
    wsel = inv = None
    if need_grad:
        wsel = torch.empty(graph.num_edges, dtype=torch.float32, device=n.device)
        inv = torch.empty(graph.num_nodes, dtype=torch.float32, device=n.device)
    ws = graph.workspace(c)
    with torch.cuda.device(n.device):
        rc = lib.sngnn_agg_forward_prepared(graph.handle, n.data_ptr(), unit.nrm.data_ptr(), _lib.ptr(unit.filt), c, k,
                                            float(thr), out.data_ptr(), _lib.ptr(wsel), _lib.ptr(inv), None, None,
                                            ws.data_ptr(), _stream(n.device))
    _lib.check(rc, "sngnn_agg_forward_prepared")
    return out, wsel


def _forward_epilogue(graph: Graph, h, unit, top_k, thr, need_grad, epi: Optional["HiddenEpilogue"], bias,
                      bits: bool = False, head: Optional["HeadEpilogue"] = None):
    """``sngnn_agg_forward_epilogue`` / ``_prepared_epilogue``: the forward whose stores apply
    bias + relu + dropout (``epi``; draws the keep mask - the caller's Bernoulli(1 - p), torch's
    generator: graph-capture safe - when the epilogue drops and has no seed) and / or that writes
    the kept bits for its backward itself (``bits``: the second result is then that bit tensor
    instead of the per-edge weights)."""
    lib = _lib.load()
    h = _check_rows(h, graph.num_total_nodes, "h")
    n, c = graph.num_nodes, h.size(1)
    k = -1 if top_k is None else int(top_k)
    out = torch.empty((n, c), dtype=torch.float32, device=h.device)
    wsel = inv = keep = kbits = None
    if need_grad and bits:
        kbits = torch.empty(max(int(lib.sngnn_graph_kept_bits_bytes(graph.handle)), 16), dtype=torch.uint8, device=h.device)
    elif need_grad:
        wsel = torch.empty(graph.num_edges, dtype=torch.float32, device=h.device)
        inv = torch.empty(n, dtype=torch.float32, device=h.device)
    if epi is None:
        epi = HiddenEpilogue(False, 0.0, False)          # (a plain forward that saves the bits)
    epi.scale = 1.0
    seed = None
    if epi.drops:
        epi.scale = 1.0 / (1.0 - epi.p)
        if epi.seed is not None:
            if epi.seed.dtype != torch.int64 or epi.seed.numel() != 1 or epi.seed.device != h.device:
                raise ValueError("seed must be an int64 tensor of one element on h's device")
            seed = epi.seed
        else:
            keep = torch.empty((n, c), dtype=torch.uint8, device=h.device).bernoulli_(1.0 - epi.p)
    if bias is not None:
        if bias.dtype != torch.float32 or bias.numel() != c or bias.device != h.device:
            raise ValueError("bias must be a float32 tensor of C elements on h's device")
        bias = bias.detach().contiguous()
    st = _lib.Epilogue(_lib.ptr(bias), _lib.ptr(keep), float(epi.scale), int(epi.relu), _lib.ptr(seed), float(epi.p),
                       _lib.ptr(kbits))
    if unit is not None and getattr(unit, "no_filter", False):
        st.no_filter = 1
    if head is not None:
        if head.y.numel() != n or head.sel.numel() != n or head.y.device != h.device or head.sel.device != h.device:
            raise ValueError("head: y and sel must hold one entry per target row, on h's device")
        hws = graph.head_workspace()
        st.head_y, st.head_sel = head.y.data_ptr(), head.sel.data_ptr()
        st.head_sets, st.head_out_mode = head.sets, head.out_mode
        st.head_n_a, st.head_n_b = head.n_a, head.n_b
        st.head_metrics, st.head_workspace = head.metrics.data_ptr(), hws.data_ptr()
    ws = graph.workspace(c)
    import ctypes
    with torch.cuda.device(h.device):
        if unit is not None and unit.n is not None:
            rc = lib.sngnn_agg_forward_prepared_epilogue(graph.handle, unit.n.data_ptr(), unit.nrm.data_ptr(),
                                                         _lib.ptr(unit.filt), c, k, float(thr), ctypes.byref(st),
                                                         out.data_ptr(), _lib.ptr(wsel), _lib.ptr(inv), ws.data_ptr(),
                                                         _stream(h.device))
        else:
            rc = lib.sngnn_agg_forward_epilogue(graph.handle, h.data_ptr(), c, k, float(thr), ctypes.byref(st),
                                                out.data_ptr(), _lib.ptr(wsel), _lib.ptr(inv), ws.data_ptr(),
                                                _stream(h.device))
    _lib.check(rc, "sngnn_agg_forward_epilogue")
    return out, (kbits if kbits is not None else wsel)


def aggregate(h: torch.Tensor, graph: Graph, top_k: Optional[int], thr: float,
              unit: Optional["UnitRows"] = None, epilogue: Optional["HiddenEpilogue"] = None,
              bias: Optional[torch.Tensor] = None, head: Optional["HeadEpilogue"] = None) -> torch.Tensor:
    """Differentiable fused aggregation: [N_total, C] -> [N, C] (N_total == N unless
    ``graph`` is a node-range partition).  ``unit``: F.normalize(h) as left by ``lin``'s
    epilogue (``UnitRows``) - the normalisation pass is skipped then.  ``epilogue`` (+ ``bias``):
    the hidden layer's bias / relu / dropout applied by the forward's stores (``HiddenEpilogue``)."""
    if head is not None:
        # (``head``: the last layer's classification head inside the forward's launches - HeadEpilogue; returns
        # the gradient of the split's mean NLL, or the logits)
        return _Aggregate.apply(h, graph, top_k, thr, unit, None, bias, head)
    return _Aggregate.apply(h, graph, top_k, thr, unit, epilogue, bias if epilogue is not None else None, None)


def attention_forward(graph: Graph, h: torch.Tensor, save_for_backward: bool = True):
    """``sngnn_attn_forward``: softmax-of-cosine attention (AGNNConv after ``lin``,
    models.py:396-405).  Returns (out [N, C], alpha [E'] in CSR order or None)."""
    lib = _lib.load()
    h = _check_rows(h, graph.num_total_nodes, "h")
    c = h.size(1)
    out = torch.empty((graph.num_nodes, c), dtype=torch.float32, device=h.device)
    alpha = (torch.empty(graph.num_edges, dtype=torch.float32, device=h.device)
             if save_for_backward else None)
    with torch.cuda.device(h.device):
        rc = lib.sngnn_attn_forward(graph.handle, h.data_ptr(), c, out.data_ptr(), _lib.ptr(alpha),
                                    graph.workspace(c).data_ptr(), _stream(h.device))
    _lib.check(rc, "sngnn_attn_forward")
    return out, alpha


def attention_backward(graph: Graph, h: torch.Tensor, grad_out: torch.Tensor,
                       alpha: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    h = _check_rows(h, graph.num_total_nodes, "h")
    grad_out = _check_rows(grad_out, graph.num_nodes, "grad_out")
    c = h.size(1)
    grad_h = torch.empty_like(h)
    with torch.cuda.device(h.device):
        rc = lib.sngnn_attn_backward(graph.handle, h.data_ptr(), c, grad_out.data_ptr(),
                                     alpha.data_ptr(), grad_h.data_ptr(),
                                     graph.workspace(c).data_ptr(), _stream(h.device))
    _lib.check(rc, "sngnn_attn_backward")
    return grad_h


class _Attention(torch.autograd.Function):
    """autograd seam of the attention mode (output fresh and unsaved: in-place ReLU safe)."""

    @staticmethod
    def forward(ctx, h, graph):
        need_grad = ctx.needs_input_grad[0]
        out, alpha = attention_forward(graph, h, save_for_backward=need_grad)
        if need_grad:
            ctx.graph = graph
            ctx.save_for_backward(h, alpha)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        h, alpha = ctx.saved_tensors
        return attention_backward(ctx.graph, h, grad_out.contiguous(), alpha), None


def attention(h: torch.Tensor, graph: Graph) -> torch.Tensor:
    """Differentiable cosine-attention aggregation: [N_total, C] -> [N, C]."""
    return _Attention.apply(h, graph)


class _SignedPropagate(torch.autograd.Function):
    """GGCNlayer_SP's two signed propagations as ONE gather (models.py:1512-1519 + 1529-1537):
    ``c2[0] * prop_pos + c2[1] * prop_neg`` with prop_+- = (adj_remove_diag * sc * e_+-) @ Wh.
    ``coef`` [E'] = adj_e * sc_e in the graph's CSR order, ``c2`` [2] device scalars.  Gradients:
    Wh from the kernels (no atomics), coef and c2 finished here from the kernel's per-edge
    ``u_e = s_e <G_i, Wh_j>``."""

    @staticmethod
    def forward(ctx, wh, coef, c2, graph):
        lib = _lib.load()
        wh = _check_rows(wh, graph.num_total_nodes, "wh")
        c = wh.size(1)
        if coef.dtype != torch.float32 or coef.numel() != graph.num_edges or not coef.is_cuda:
            raise ValueError("coef must be a float32 GPU tensor with one entry per edge of the graph")
        if c2.dtype != torch.float32 or c2.numel() != 2 or not c2.is_cuda:
            raise ValueError("c2 must be a float32 GPU tensor of 2 elements")
        coef, c2 = coef.contiguous(), c2.contiguous()
        out = torch.empty((graph.num_nodes, c), dtype=torch.float32, device=wh.device)
        s = torch.empty(graph.num_edges, dtype=torch.float32, device=wh.device)
        with torch.cuda.device(wh.device):
            rc = lib.sngnn_signed_forward(graph.handle, wh.data_ptr(), c, coef.data_ptr(), c2.data_ptr(),
                                          out.data_ptr(), s.data_ptr(), graph.workspace(c).data_ptr(),
                                          _stream(wh.device))
        _lib.check(rc, "sngnn_signed_forward")
        ctx.graph = graph
        ctx.save_for_backward(wh, coef, c2, s)
        return out

    @staticmethod
    def backward(ctx, g):
        wh, coef, c2, s = ctx.saved_tensors
        graph = ctx.graph
        lib = _lib.load()
        g = _check_rows(g.contiguous(), graph.num_nodes, "grad_out")
        c = wh.size(1)
        grad_wh = torch.empty_like(wh)
        u = torch.empty_like(s)
        with torch.cuda.device(wh.device):
            rc = lib.sngnn_signed_backward(graph.handle, wh.data_ptr(), c, g.data_ptr(), coef.data_ptr(),
                                           s.data_ptr(), c2.data_ptr(), grad_wh.data_ptr(), u.data_ptr(),
                                           graph.workspace(c).data_ptr(), _stream(wh.device))
        _lib.check(rc, "sngnn_signed_backward")
        pos, neg = s > 0, s < 0
        zero = torch.zeros((), dtype=torch.float32, device=s.device)
        kappa = torch.where(pos, c2[0], torch.where(neg, c2[1], zero))
        au = coef * u
        grad_c2 = torch.stack([torch.where(pos, au, zero).sum(), torch.where(neg, au, zero).sum()])
        return grad_wh, kappa * u, grad_c2, None


def signed_propagate(wh: torch.Tensor, coef: torch.Tensor, c2: torch.Tensor, graph: Graph) -> torch.Tensor:
    """Differentiable signed cosine propagation: [N_total, C] -> [N, C] (see ``_SignedPropagate``)."""
    return _SignedPropagate.apply(wh, coef, c2, graph)


class _WeightedPropagate(torch.autograd.Function):
    """``torch.sparse.mm(A, X)`` for a sparse ``A`` whose pattern is a device graph (GGCNlayer_SP's plain
    propagation, models.py:1544-1549): ``out[i] = sum_e w_e X[src_e]`` over the in-edges of i, with autograd
    through X (the transpose gather-sum) and through the per-entry weights (``<G_i, X_j>`` per entry) -
    ``sngnn_weighted_gather_sum_rows`` / ``_scatter_sum_rows`` / ``sngnn_pair_dot_rows``: gathers with a fixed
    order, no atomics.  ``w_csr`` [E'] in the graph's CSR order; ``aux`` = (csc_eid int64 [E'], tgt int32 [E'],
    src int32 [E']) - the graph's CSC -> CSR map and the CSR entries' rows and columns, built once per graph."""

    @staticmethod
    def forward(ctx, x, w_csr, graph, aux):
        lib = _lib.load()
        x = _check_rows(x, graph.num_total_nodes, "x")
        c = x.size(1)
        if w_csr.dtype != torch.float32 or w_csr.numel() != graph.num_edges or not w_csr.is_cuda:
            raise ValueError("w_csr must be a float32 GPU tensor with one entry per edge of the graph")
        w_csr = w_csr.contiguous()
        out = torch.empty((graph.num_nodes, c), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            rc = lib.sngnn_weighted_gather_sum_rows(graph.handle, x.data_ptr(), w_csr.data_ptr(), c, out.data_ptr(),
                                                    graph.workspace(c).data_ptr(), _stream(x.device))
        _lib.check(rc, "sngnn_weighted_gather_sum_rows")
        ctx.graph, ctx.aux = graph, aux
        ctx.save_for_backward(x, w_csr)
        return out

    @staticmethod
    def backward(ctx, g):
        x, w_csr = ctx.saved_tensors
        graph = ctx.graph
        csc_eid, tgt, src = ctx.aux
        lib = _lib.load()
        g = _check_rows(g.contiguous(), graph.num_nodes, "grad_out")
        c = x.size(1)
        gx = gw = None
        with torch.cuda.device(x.device):
            if ctx.needs_input_grad[0]:
                gx = torch.empty_like(x)
                w_csc = w_csr.index_select(0, csc_eid)
                rc = lib.sngnn_weighted_scatter_sum_rows(graph.handle, g.data_ptr(), w_csc.data_ptr(), c, gx.data_ptr(),
                                                         graph.workspace(c).data_ptr(), _stream(x.device))
                _lib.check(rc, "sngnn_weighted_scatter_sum_rows")
            if ctx.needs_input_grad[1]:
                gw = torch.empty_like(w_csr)
                rc = lib.sngnn_pair_dot_rows(g.data_ptr(), tgt.data_ptr(), x.data_ptr(), src.data_ptr(), w_csr.numel(), c,
                                             gw.data_ptr(), _stream(x.device))
                _lib.check(rc, "sngnn_pair_dot_rows")
        return gx, gw, None, None


def weighted_propagate(x: torch.Tensor, w_csr: torch.Tensor, graph: Graph, aux) -> torch.Tensor:
    """Differentiable weighted gather-sum over a graph's in-edges (see ``_WeightedPropagate``)."""
    return _WeightedPropagate.apply(x, w_csr, graph, aux)


def adj_linear_forward(graph: Graph, wt: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    lib = _lib.load()
    wt = _check_rows(wt, graph.num_nodes, "wt")
    c = wt.size(1)
    out0 = torch.empty_like(wt)
    b = None if bias is None else bias.contiguous()
    with torch.cuda.device(wt.device):
        rc = lib.sngnn_adj_linear_forward(graph.handle, wt.data_ptr(), _lib.ptr(b), c,
                                          out0.data_ptr(), graph.workspace(c).data_ptr(),
                                          _stream(wt.device))
    _lib.check(rc, "sngnn_adj_linear_forward")
    return out0


def adj_linear_backward(graph: Graph, g0: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    g0 = _check_rows(g0, graph.num_nodes, "g0")
    dwt = torch.empty_like(g0)
    with torch.cuda.device(g0.device):
        rc = lib.sngnn_adj_linear_backward(graph.handle, g0.data_ptr(), g0.size(1),
                                           dwt.data_ptr(), graph.workspace(g0.size(1)).data_ptr(),
                                           _stream(g0.device))
    _lib.check(rc, "sngnn_adj_linear_backward")
    return dwt


class _AdjLinear(torch.autograd.Function):
    """``Linear(num_nodes, C)`` applied to the sparse adjacency (models.py:124-130).
    ``weight`` is the reference-shaped [C, N] parameter stored column-major, i.e.
    ``weight.t()`` is a contiguous [N, C] table whose rows are gathered."""

    @staticmethod
    def forward(ctx, weight, bias, graph):
        wt = weight.t()
        if not wt.is_contiguous():
            wt = wt.contiguous()
        ctx.graph = graph
        ctx.has_bias = bias is not None
        return adj_linear_forward(graph, wt, bias)

    @staticmethod
    def backward(ctx, g0):
        g0 = g0.contiguous()
        dwt = adj_linear_backward(ctx.graph, g0)
        db = g0.sum(dim=0) if ctx.has_bias else None
        return dwt.t(), db, None


def adj_linear(weight: torch.Tensor, bias: Optional[torch.Tensor], graph: Graph) -> torch.Tensor:
    return _AdjLinear.apply(weight, bias, graph)


class _AdjLinearPartition(torch.autograd.Function):
    """The adjacency branch of one rank of a node-range partition.  ``graph_out`` is the
    partition of the FLIPPED edge list (its owned rows are the rank's own source
    nodes, its columns the global targets), ``weight`` the full [C, N_total] table.
    Forward gathers W^T rows of the targets of each owned node's out-edges; backward
    scatters into the rank's PARTIAL dense gradient of the full table (summed over the
    ranks by the caller's all-reduce, sngnn_amd/dist.py)."""

    @staticmethod
    def forward(ctx, weight, bias, graph_out):
        lib = _lib.load()
        wt = weight.t()
        if not wt.is_contiguous():
            wt = wt.contiguous()
        c = wt.size(1)
        if wt.size(0) != graph_out.num_total_nodes:
            raise ValueError("w.weight must cover all N_total nodes")
        if graph_out.src_min != 0 and graph_out.num_edges > 0:
            raise ValueError("the partitioned adjacency branch needs a graph whose lowest target id is 0")
        out = torch.empty((graph_out.num_nodes, c), dtype=torch.float32, device=wt.device)
        b = None if bias is None else bias.contiguous()
        with torch.cuda.device(wt.device):
            rc = lib.sngnn_gather_sum_rows(graph_out.handle, wt.data_ptr(), _lib.ptr(b), c, out.data_ptr(),
                                           graph_out.workspace(c).data_ptr(), _stream(wt.device))
        _lib.check(rc, "sngnn_gather_sum_rows")
        ctx.graph = graph_out
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g0):
        lib = _lib.load()
        g0 = g0.contiguous()
        gr = ctx.graph
        c = g0.size(1)
        dwt = torch.empty((gr.num_total_nodes, c), dtype=torch.float32, device=g0.device)
        with torch.cuda.device(g0.device):
            rc = lib.sngnn_scatter_sum_rows(gr.handle, g0.data_ptr(), c, dwt.data_ptr(),
                                            gr.workspace(c).data_ptr(), _stream(g0.device))
        _lib.check(rc, "sngnn_scatter_sum_rows")
        db = g0.sum(dim=0) if ctx.has_bias else None
        return dwt.t(), db, None


def adj_linear_partition(weight, bias, graph_out: Graph) -> torch.Tensor:
    return _AdjLinearPartition.apply(weight, bias, graph_out)


class _GatherSum(torch.autograd.Function):
    """``out[i] = bias + sum_{q in row i} table[col_q]`` over a graph's CSR rows and its
    transpose as the gradient (``sngnn_gather_sum_rows`` / ``sngnn_scatter_sum_rows``): the
    adjacency branch of SNGNN++ on one rank when ``w`` is sharded - ``table`` is then the
    rank's own rows of W^T followed by the halo rows (sngnn_amd/dist.py), ``graph`` the local
    form of the flipped edge list."""

    @staticmethod
    def forward(ctx, table, bias, graph):
        lib = _lib.load()
        table = _check_rows(table, graph.num_total_nodes, "table")
        c = table.size(1)
        out = torch.empty((graph.num_nodes, c), dtype=torch.float32, device=table.device)
        b = None if bias is None else bias.contiguous()
        with torch.cuda.device(table.device):
            rc = lib.sngnn_gather_sum_rows(graph.handle, table.data_ptr(), _lib.ptr(b), c, out.data_ptr(),
                                           graph.workspace(c).data_ptr(), _stream(table.device))
        _lib.check(rc, "sngnn_gather_sum_rows")
        ctx.graph = graph
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, g0):
        lib = _lib.load()
        g0 = g0.contiguous()
        gr = ctx.graph
        c = g0.size(1)
        dtab = torch.empty((gr.num_total_nodes, c), dtype=torch.float32, device=g0.device)
        with torch.cuda.device(g0.device):
            rc = lib.sngnn_scatter_sum_rows(gr.handle, g0.data_ptr(), c, dtab.data_ptr(),
                                            gr.workspace(c).data_ptr(), _stream(g0.device))
        _lib.check(rc, "sngnn_scatter_sum_rows")
        db = g0.sum(dim=0) if ctx.has_bias else None
        return dtab, db, None


def gather_sum(table: torch.Tensor, bias: Optional[torch.Tensor], graph: Graph) -> torch.Tensor:
    return _GatherSum.apply(table, bias, graph)


# ---------------------------------------------------------------------------
# Callers on either side of the aggregation
# ---------------------------------------------------------------------------
_ws_cache = {}


def _workspace(key, nbytes, device):
    ws = _ws_cache.get((key, str(device)))
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)
        _ws_cache[(key, str(device))] = ws
    return ws


class OutBuffer:
    """Holder of a preallocated result tensor for ``_Linear`` (``.t``)."""

    def __init__(self, t: torch.Tensor):
        self.t = t


class _Linear(torch.autograd.Function):
    """``self.lin(x)`` (models.py:121,237,324): hand-written streaming MFMA forward for
    narrow layers on big graphs (rocBLAS otherwise), hand-written weight
    gradient (the [C, F] result reduces over all N rows - a shape the BLAS heuristic
    handles poorly), grad_x through rocBLAS only when x needs it."""

    @staticmethod
    def forward(ctx, x, weight, bias, pad_to=None, out=None, unit=None, act=None):
        """``pad_to``: produce [N, pad_to] with zero channels behind the layer's own (16-byte
        rows for the aggregation kernels).  The padded weight / bias live in two persistent
        buffers attached to the weight and are refreshed by two small copies - no
        concatenation, no allocation and no autograd node per forward.
        ``out``: ``OutBuffer`` around a contiguous [N, C] tensor to write into (the head of a rank's
        [own | halo] feature table, sngnn_amd/dist.py) - that tensor is returned as the result.
        ``unit``: a ``UnitRows`` to fill with F.normalize of the result from the same launch
        (models.py:237-238 are adjacent lines); left empty when the shape takes the BLAS.
        ``act``: the ``HiddenEpilogue`` whose activated output ``x`` is (the wrapper hands that
        tensor to this layer only): the backward then applies relu' and the dropout mask inside
        the kernel that writes grad_x (``sngnn_linear_forward_masked``) and tells the epilogue so."""
        ctx.save_for_backward(x, weight)
        ctx.act = act if (act is not None and act.applied) else None
        ctx.has_bias = bias is not None
        n, f = x.shape
        c = weight.size(0)
        ctx.c = c
        if pad_to is not None and pad_to > c:
            pad = getattr(weight, "_sngnn_pad", None)
            if pad is None or pad[0].size(0) != pad_to or pad[0].device != weight.device:
                pad = (weight.new_zeros((pad_to, f)), weight.new_zeros(pad_to))
                weight._sngnn_pad = pad
            pad[0][:c].copy_(weight)
            if bias is not None:
                pad[1][:c].copy_(bias)
            weight, bias = pad[0], (pad[1] if bias is not None else None)
            c = pad_to
        # The hand-written forward is a streaming kernel for NARROW inputs on big graphs (F <= 128:
        # all of W^T in registers, x the only stream; 26.8 us against rocBLAS' 48 us at
        # 169 343 x 128 -> 40).  A wide layer, a tiny graph or a wide INPUT (the first layer of the
        # real datasets: F = 932 .. 2 325) goes to the BLAS, which tiles the contraction
        # (7 600 x 932 -> 32: 18.7 us against 36.7; 2 277 x 2 325 -> 32: 19 us against 83).
        if out is not None:
            out = out.t          # (a holder object, not a tensor argument: autograd must see the result
            #                       as this node's fresh output, not as an input modified in place)
            if out.shape != (n, c) or not out.is_contiguous() or out.dtype != torch.float32:
                raise ValueError("out must be a contiguous float32 [N, C] buffer")
        if c > 64 or n < 4096 or f > 128:
            if out is None:
                return torch.nn.functional.linear(x, weight, bias)
            if bias is None:
                return torch.mm(x, weight.t(), out=out)
            return torch.addmm(bias, x, weight.t(), out=out)
        xc, wc = x.contiguous(), weight.contiguous()
        bc = None if bias is None else bias.contiguous()
        h = torch.empty((n, c), dtype=torch.float32, device=x.device) if out is None else out
        lib = _lib.load()
        if unit is not None and lib.sngnn_linear_normalized_supported(n, f, c) and xc.data_ptr() % 16 == 0 \
                and wc.data_ptr() % 16 == 0 and h.data_ptr() % 16 == 0:
            unit.n = torch.empty((n, c), dtype=torch.float32, device=x.device)
            unit.nrm = torch.empty(n, dtype=torch.float32, device=x.device)
            fb = int(lib.sngnn_filter_row_bytes(c)) if unit.want_filter else 0
            unit.filt = torch.empty((n, fb), dtype=torch.uint8, device=x.device) if fb == 128 else None
            with torch.cuda.device(x.device):
                rc = lib.sngnn_linear_forward_normalized(xc.data_ptr(), wc.data_ptr(), _lib.ptr(bc), n, f, c,
                                                         h.data_ptr(), unit.n.data_ptr(), unit.nrm.data_ptr(),
                                                         _lib.ptr(unit.filt), _stream(x.device))
            _lib.check(rc, "sngnn_linear_forward_normalized")
            return h
        with torch.cuda.device(x.device):
            rc = lib.sngnn_linear_forward(xc.data_ptr(), wc.data_ptr(), _lib.ptr(bc), n, f, c,
                                          h.data_ptr(), _stream(x.device))
        _lib.check(rc, "sngnn_linear_forward")
        return h

    @staticmethod
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g = g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            n, fin = g.size(0), weight.size(1)
            act = ctx.act
            if g.is_cuda and g.size(1) == ctx.c and ctx.c <= 128 and fin <= 64 and n >= 4096:
                # grad_x = g W is the same narrow streaming product as the forward, with W^T as the
                # weight: 35 us against rocBLAS' 55 us at 169 343 x 40 -> 64 (gpurun_out/gx_cmp.log)
                wt = weight.t().contiguous()
                gx = torch.empty((n, fin), dtype=torch.float32, device=g.device)
                # x is an activated tensor (relu + dropout of the layer before): their backward is the
                # store epilogue of this product where the panel kernel runs it anyway (ctx.c not a
                # row-tile width); the row-tile widths keep their faster kernel and leave the mask
                # to the producer's backward
                masked = act is not None and ctx.c not in (16, 32, 64, 128) and x.is_contiguous()
                with torch.cuda.device(g.device):
                    if masked:
                        rc = _lib.load().sngnn_linear_forward_masked(g.data_ptr(), wt.data_ptr(), None, n, ctx.c, fin,
                                                                     x.data_ptr(), float(act.scale), gx.data_ptr(),
                                                                     _stream(g.device))
                        act.premasked = gx.data_ptr()          # (which tensor: _take_premasked checks it)
                    else:
                        rc = _lib.load().sngnn_linear_forward(g.data_ptr(), wt.data_ptr(), None, n, ctx.c, fin,
                                                              gx.data_ptr(), _stream(g.device))
                _lib.check(rc, "sngnn_linear_forward (input gradient)")
            else:
                gx = g[:, :ctx.c].mm(weight)         # (the padded channels carry no gradient)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            lib = _lib.load()
            xc = x.contiguous()
            n, f = xc.shape
            c = g.size(1)
            gw = torch.empty((c, f), dtype=torch.float32, device=g.device)
            gb = torch.empty(c, dtype=torch.float32, device=g.device) if ctx.has_bias else None
            ws = _workspace("wgrad", lib.sngnn_linear_wgrad_workspace_bytes(n, c, f), g.device)
            with torch.cuda.device(g.device):
                rc = lib.sngnn_linear_wgrad(g.data_ptr(), xc.data_ptr(), n, c, f, gw.data_ptr(),
                                            _lib.ptr(gb), ws.data_ptr(), _stream(g.device))
            _lib.check(rc, "sngnn_linear_wgrad")
            gw = gw[:ctx.c]
            gb = None if gb is None else gb[:ctx.c]
        return gx, gw, gb, None, None, None, None


def linear(x: torch.Tensor, lin: torch.nn.Linear) -> torch.Tensor:
    """Apply ``lin`` with the hand-written weight gradient (fp32 GPU tensors), or
    plain ``lin(x)`` for anything else."""
    if x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and lin.weight.dtype == torch.float32:
        return _Linear.apply(x, lin.weight, lin.bias, None, None, None, None)
    return lin(x)


class _Blend(torch.autograd.Function):
    """``beta * out_0 + (1 - beta) * out_1`` (models.py:134) in one pass each way.  With ``epi`` (a
    HiddenEpilogue: relu + seeded dropout, no bias) the hidden layer's activation behind the blend
    (models.py:81-84) is applied in the same store, and undone in the same backward pass."""

    @staticmethod
    def forward(ctx, out0, out1, beta, epi=None):
        lib = _lib.load()
        out = torch.empty_like(out0)
        ctx.epi = epi
        with torch.cuda.device(out0.device):
            if epi is None:
                rc = lib.sngnn_blend_forward(out0.data_ptr(), out1.data_ptr(), beta.data_ptr(), out0.numel(),
                                             out.data_ptr(), _stream(out0.device))
            else:
                import ctypes
                epi.scale = 1.0 / (1.0 - epi.p) if epi.drops else 1.0
                seed = epi.seed if epi.drops else None
                st = _lib.Epilogue(None, None, float(epi.scale), int(epi.relu), _lib.ptr(seed), float(epi.p), None)
                rc = lib.sngnn_blend_forward_epilogue(out0.data_ptr(), out1.data_ptr(), beta.data_ptr(), out0.numel(),
                                                      ctypes.byref(st), out.data_ptr(), _stream(out0.device))
        _lib.check(rc, "sngnn_blend_forward")
        if epi is None:
            ctx.save_for_backward(out0, out1, beta)
        else:
            ctx.save_for_backward(out0, out1, beta, out)
        return out

    @staticmethod
    def backward(ctx, g):
        epi = ctx.epi
        act = None
        if epi is None:
            out0, out1, beta = ctx.saved_tensors
        else:
            out0, out1, beta, out = ctx.saved_tensors
            if epi.premasked:          # the consumer's store already applied relu' and the dropout mask
                _take_premasked(epi, g)
            else:
                act = out
        lib = _lib.load()
        g = g.contiguous()
        g0, g1 = torch.empty_like(g), torch.empty_like(g)
        gbeta = torch.empty_like(beta)
        ws = _workspace("blend", lib.sngnn_blend_workspace_bytes(), g.device)
        with torch.cuda.device(g.device):
            if act is None:
                rc = lib.sngnn_blend_backward(g.data_ptr(), out0.data_ptr(), out1.data_ptr(), beta.data_ptr(),
                                              g.numel(), g0.data_ptr(), g1.data_ptr(), gbeta.data_ptr(),
                                              ws.data_ptr(), _stream(g.device))
            else:
                rc = lib.sngnn_blend_backward_epilogue(g.data_ptr(), out0.data_ptr(), out1.data_ptr(), beta.data_ptr(),
                                                       g.numel(), act.data_ptr(), float(epi.scale), g0.data_ptr(),
                                                       g1.data_ptr(), gbeta.data_ptr(), ws.data_ptr(), _stream(g.device))
        _lib.check(rc, "sngnn_blend_backward")
        return g0, g1, gbeta, None


class _BlendHead(torch.autograd.Function):
    """The LAST SNGNN++ layer's blend (models.py:134) with the classification head behind it
    (``HeadEpilogue``) in one pass: the blended logits are formed in registers (the blend kernel's own
    rounding) and go straight through log_softmax / NLL / accuracy - ``sngnn_head_nll_blend``.  The result is
    what the head was asked to leave: d loss / d logits (training; ``G.backward(G.detach())`` is then the
    backward of ``loss.backward()``: this node's backward is the blend's) or the logits."""

    @staticmethod
    def forward(ctx, out0, out1, beta, head):
        lib = _lib.load()
        n, c = out0.shape
        out = torch.empty_like(out0)
        ws = _workspace("head", lib.sngnn_head_workspace_bytes(n), out0.device)
        with torch.cuda.device(out0.device):
            rc = lib.sngnn_head_nll_blend(out0.data_ptr(), out1.data_ptr(), beta.data_ptr(), head.y.data_ptr(),
                                          head.sel.data_ptr(), n, c, head.sets, head.n_a, head.n_b,
                                          out.data_ptr() if head.grad else None, None if head.grad else out.data_ptr(),
                                          head.metrics.data_ptr(), ws.data_ptr(), _stream(out0.device))
        _lib.check(rc, "sngnn_head_nll_blend")
        ctx.save_for_backward(out0, out1, beta)
        return out

    @staticmethod
    def backward(ctx, g):
        out0, out1, beta = ctx.saved_tensors
        lib = _lib.load()
        g = g.contiguous()
        g0, g1 = torch.empty_like(g), torch.empty_like(g)
        gbeta = torch.empty_like(beta)
        ws = _workspace("blend", lib.sngnn_blend_workspace_bytes(), g.device)
        with torch.cuda.device(g.device):
            rc = lib.sngnn_blend_backward(g.data_ptr(), out0.data_ptr(), out1.data_ptr(), beta.data_ptr(), g.numel(),
                                          g0.data_ptr(), g1.data_ptr(), gbeta.data_ptr(), ws.data_ptr(), _stream(g.device))
        _lib.check(rc, "sngnn_blend_backward")
        return g0, g1, gbeta, None


def blend_head(out0: torch.Tensor, out1: torch.Tensor, beta: torch.Tensor, head: "HeadEpilogue") -> Optional[torch.Tensor]:
    """The blend with the head behind it (``_BlendHead``) where the shapes allow - contiguous fp32 [N, C] rows of
    16-byte vectors, C <= 64, labels and split flags for every row; ``head.applied`` says whether it ran
    (None is returned when it did not: the caller blends and runs the head itself)."""
    ok = (out0.is_cuda and out0.dtype == torch.float32 and out1.dtype == torch.float32 and out0.shape == out1.shape
          and out0.dim() == 2 and out0.is_contiguous() and out1.is_contiguous() and beta.numel() == 1
          and beta.dtype == torch.float32 and beta.is_cuda and head.y.numel() == out0.size(0)
          and head.sel.numel() == out0.size(0) and head.y.device == out0.device
          and bool(_lib.load().sngnn_head_nll_blend_supported(out0.size(1))))
    head.applied = ok
    return _BlendHead.apply(out0, out1, beta, head) if ok else None


def blend(out0: torch.Tensor, out1: torch.Tensor, beta: torch.Tensor,
          epilogue: Optional["HiddenEpilogue"] = None) -> torch.Tensor:
    """SNGNN++'s ``beta * out_0 + (1 - beta) * out_1``; fused for contiguous fp32 GPU tensors of
    one shape with a one-element fp32 ``beta``, the plain expression otherwise.  ``epilogue``: the
    hidden layer's relu + dropout behind it in the same pass (``epilogue.applied`` says whether the
    fused kernel took it; needs a seed for the dropout)."""
    fused = (out0.is_cuda and out0.dtype == torch.float32 and out1.dtype == torch.float32 and out0.shape == out1.shape
             and out0.is_contiguous() and out1.is_contiguous() and beta.numel() == 1
             and beta.dtype == torch.float32 and beta.is_cuda)
    if epilogue is not None:
        epilogue.applied = fused and (not epilogue.drops or epilogue.seed is not None)
        if not epilogue.applied:
            epilogue = None
    if fused:
        return _Blend.apply(out0, out1, beta, epilogue)
    return beta * out0 + (1 - beta) * out1


class _HeadNLL(torch.autograd.Function):
    """mean NLL of log_softmax(logits) over the masked rows; also returns the number
    of correctly classified masked rows (no gradient)."""

    @staticmethod
    def forward(ctx, logits, y, row_mask_u8, n_masked, out=None):
        lib = _lib.load()
        z = logits.contiguous()
        n, c = z.shape
        need_grad = ctx.needs_input_grad[0]
        grad = torch.empty_like(z) if need_grad else None
        if out is None:
            out = torch.empty(2, dtype=torch.float32, device=z.device)
        ws = _workspace("head", lib.sngnn_head_workspace_bytes(n), z.device)
        with torch.cuda.device(z.device):
            rc = lib.sngnn_head_nll(z.data_ptr(), y.data_ptr(), row_mask_u8.data_ptr(), n, c,
                                    int(n_masked), _lib.ptr(grad), out.data_ptr(), ws.data_ptr(),
                                    _stream(z.device))
        _lib.check(rc, "sngnn_head_nll")
        if need_grad:
            ctx.save_for_backward(grad)
        loss, correct = out[0], out[1]
        ctx.mark_non_differentiable(correct)
        return loss, correct

    @staticmethod
    def backward(ctx, g_loss, _g_correct):
        (grad,) = ctx.saved_tensors
        return grad * g_loss, None, None, None, None


def head_nll_with_grad(logits: torch.Tensor, y: torch.Tensor, row_mask_u8: torch.Tensor, n_masked: int,
                       out: Optional[torch.Tensor] = None):
    """The head kernel outside autograd: returns ((loss, n_correct) as views of ``out``,
    d loss / d logits).  A trainer calls ``logits.backward(grad)`` with it - the same
    gradient ``loss.backward()`` produces through :func:`head_nll`, without the ones-fill and
    the ``grad * 1`` pass over [N, C] that the generic autograd seam costs."""
    lib = _lib.load()
    z = logits.detach().contiguous()
    n, c = z.shape
    grad = torch.empty_like(z)
    if out is None:
        out = torch.empty(2, dtype=torch.float32, device=z.device)
    ws = _workspace("head", lib.sngnn_head_workspace_bytes(n), z.device)
    with torch.cuda.device(z.device):
        rc = lib.sngnn_head_nll(z.data_ptr(), y.data_ptr(), row_mask_u8.data_ptr(), n, c, int(n_masked),
                                grad.data_ptr(), out.data_ptr(), ws.data_ptr(), _stream(z.device))
    _lib.check(rc, "sngnn_head_nll")
    return (out[0], out[1]), grad


@torch.no_grad()
def head_nll2(logits: torch.Tensor, y: torch.Tensor, row_sets_u8: torch.Tensor, n_a: int, n_b: int,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``sngnn_head_nll2``: (mean NLL, n_correct) of TWO splits read off one forward - the
    validation and test metrics of train.py:92-117.  ``row_sets_u8``: uint8 [N], bit 0 = split A,
    bit 1 = split B.  Returns fp32 [4] = (loss A, correct A, loss B, correct B) (``out`` if
    given).  No autograd (evaluation); C <= 64."""
    if logits.dtype != torch.float32 or not logits.is_cuda or logits.dim() != 2:
        raise ValueError("logits must be a float32 GPU tensor [N, C]")
    if y.dtype != torch.int64 or row_sets_u8.dtype != torch.uint8:
        raise ValueError("y must be int64 and row_sets uint8")
    if out is None:
        out = torch.empty(4, dtype=torch.float32, device=logits.device)
    elif out.dtype != torch.float32 or out.numel() != 4 or not out.is_contiguous() or out.device != logits.device:
        raise ValueError("out must be a contiguous float32 tensor of 4 elements on the logits' device")
    lib = _lib.load()
    z = logits.detach().contiguous()
    n, c = z.shape
    ws = _workspace("head", lib.sngnn_head_workspace_bytes(n), z.device)
    with torch.cuda.device(z.device):
        _lib.check(lib.sngnn_head_nll2(z.data_ptr(), y.contiguous().data_ptr(), row_sets_u8.contiguous().data_ptr(),
                                       n, c, int(n_a), int(n_b), out.data_ptr(), ws.data_ptr(), _stream(z.device)),
                   "sngnn_head_nll2")
    return out


def head_nll(logits: torch.Tensor, y: torch.Tensor, row_mask_u8: torch.Tensor, n_masked: int,
             out: Optional[torch.Tensor] = None):
    """(loss, n_correct) of the masked rows: fused log_softmax + nll_loss + accuracy
    (models.py:86 + train.py:81-84).  ``row_mask_u8``: uint8 [N].  ``out``: optional
    contiguous fp32 [2] the kernel writes (loss, n_correct) into - the returned tensors are
    views of it (a trainer's metrics buffer, no copy)."""
    if out is not None and (out.dtype != torch.float32 or out.numel() != 2 or not out.is_contiguous()
                            or out.device != logits.device):
        raise ValueError("out must be a contiguous float32 tensor of 2 elements on the logits' device")
    if logits.dtype != torch.float32 or not logits.is_cuda or logits.dim() != 2:
        raise ValueError("logits must be a float32 GPU tensor [N, C]")
    if y.dtype != torch.int64 or row_mask_u8.dtype != torch.uint8:
        raise ValueError("y must be int64 and row_mask uint8")
    return _HeadNLL.apply(logits, y.contiguous(), row_mask_u8.contiguous(), n_masked, out)
