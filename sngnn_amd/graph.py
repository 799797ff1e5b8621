"""Device graph structure, built once per (edge_index, self-loop mode).

The reference redoes ``add_self_loops`` / ``remove_self_loops`` and the implicit
per-target grouping on every forward (models/models.py:117-120, 234-236, 323);
here the C library builds CSR-by-target / CSC-by-source once and the conv layers
look it up in a small cache keyed on the identity and version of ``edge_index``.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Tuple

import numpy as np
import torch

from . import _lib

_ARRAYS = {"rowptr": 0, "col": 1, "eid": 2, "cscptr": 3, "csc_eid": 4, "rperm": 5}


LOOPS_REPLACE = 2      # SNGNN_LOOPS_REPLACE: drop the original loops, then append one per node


class Graph:
    """Owns a ``sngnn_graph_t`` handle."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, add_loops: bool,
                 remove_loops: bool, row_range=None):
        """``row_range=(begin, end)`` builds the node-range partition that owns the
        targets [begin, end) of a ``num_nodes``-node graph (global ids in
        ``edge_index``); ``None`` is the whole graph."""
        if edge_index.dim() != 2 or edge_index.size(0) != 2 or edge_index.dtype != torch.int64:
            raise ValueError("edge_index must be an int64 tensor of shape [2, E]")
        if not edge_index.is_cuda:
            raise ValueError("edge_index must live on the GPU (there is no CPU path)")
        lib = _lib.load()
        ei = edge_index.contiguous()
        self.device = ei.device
        # remove_loops: False/True as in the SNConv layers, or LOOPS_REPLACE (AGNNConv's order)
        self.add_loops, self.remove_loops = bool(add_loops), int(remove_loops)
        handle = C.c_void_p()
        with torch.cuda.device(self.device):
            stream = torch.cuda.current_stream(self.device).cuda_stream
            r0, r1 = (0, int(num_nodes)) if row_range is None else map(int, row_range)
            rc = lib.sngnn_graph_create_partition(ei.data_ptr(), ei.size(1), int(num_nodes),
                                                  r0, r1, int(add_loops), int(remove_loops),
                                                  stream, C.byref(handle))
        _lib.check(rc, "sngnn_graph_create_partition")
        self._h = handle
        self.num_nodes = int(lib.sngnn_graph_num_nodes(handle))            # owned rows
        self.num_total_nodes = int(lib.sngnn_graph_num_total_nodes(handle))
        self.row_offset = int(lib.sngnn_graph_row_offset(handle))
        self.num_edges = int(lib.sngnn_graph_num_edges(handle))
        self.max_in_degree = int(lib.sngnn_graph_max_in_degree(handle))
        self.num_fused_nodes = int(lib.sngnn_graph_num_fused_nodes(handle))    # in- and out-degree <= 16 (node-centric backward)
        self.src_min = int(lib.sngnn_graph_src_min(handle))
        self._ws: Dict[Tuple[int, int], torch.Tensor] = {}

    @property
    def handle(self):
        return self._h

    def workspace(self, channels: int) -> torch.Tensor:
        """Scratch for forward/backward at ``channels``: one buffer per (width, stream), so
        calls on the same graph from different streams never share scratch (the entry
        points are re-entrant per stream).  Allocated on first use: ``GraphedEpoch`` runs an
        eager pass on the very stream it then captures on, so the capture finds its buffers."""
        key = (channels, torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = int(_lib.load().sngnn_graph_workspace_bytes(self._h, channels))
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def head_workspace(self) -> torch.Tensor:
        """Scratch of the classification head inside this graph's forward (``sngnn_epilogue_t.head_workspace``):
        one buffer per stream, owned by the graph like :meth:`workspace` - a captured epoch that holds the
        graph holds the buffer its launches point into."""
        key = ("head", torch.cuda.current_stream(self.device).cuda_stream)
        ws = self._ws.get(key)
        if ws is None:
            nbytes = int(_lib.load().sngnn_agg_head_workspace_bytes(self._h))
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def array(self, name: str) -> np.ndarray:
        """Host copy of one of the structure arrays (tests / inspection)."""
        which = _ARRAYS[name]
        n = {0: self.num_nodes + 1, 3: self.num_total_nodes + 1,
             5: self.num_nodes}.get(which, self.num_edges)
        out = np.empty(n, dtype=np.int32)
        _lib.check(_lib.load().sngnn_graph_copy_array(self._h, which, out.ctypes.data),
                   "sngnn_graph_copy_array")
        return out

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _lib.load().sngnn_graph_destroy(h)
            except Exception:
                pass
            self._h = None


class GraphCache:
    """Per-module cache: (edge_index storage, shape, version, loop mode) -> Graph."""

    def __init__(self, max_entries: int = 4):
        self._entries: Dict[Tuple, Tuple[Graph, torch.Tensor]] = {}
        self._max = max_entries
        self._recording = None          # list of graphs handed out while record() is active

    def get(self, edge_index: torch.Tensor, num_nodes: int, add_loops: bool,
            remove_loops: bool, row_range=None) -> Graph:
        key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version,
               int(num_nodes), bool(add_loops), int(remove_loops), str(edge_index.device),
               None if row_range is None else tuple(map(int, row_range)))
        hit = self._entries.get(key)
        if hit is None:
            if len(self._entries) >= self._max:
                self._entries.pop(next(iter(self._entries)))
            # the entry keeps edge_index alive, so its address cannot be recycled
            # for a different tensor while the key is cached
            hit = (Graph(edge_index, num_nodes, add_loops, remove_loops, row_range), edge_index)
            self._entries[key] = hit
        if self._recording is not None and not any(g is hit[0] for g in self._recording):
            self._recording.append(hit[0])
        return hit[0]

    def record(self):
        """Context manager: collects the Graph objects ``get`` hands out inside it (the graphs a
        captured epoch launches on - what its owner must keep alive)."""
        cache = self

        class _Rec:
            def __enter__(self):
                cache._recording = self.graphs = []
                return self.graphs

            def __exit__(self, *exc):
                cache._recording = None
                return False
        return _Rec()

    def snapshot(self):
        """The Graph objects currently cached.  Whoever captured raw pointers of them (a HIP
        graph holds the CSR arrays and the workspaces of the graphs its kernels were launched
        on) keeps this list, so eviction from the cache cannot free them under the capture."""
        return [g for g, _ in self._entries.values()]

    def clear(self):
        self._entries.clear()


# Graphs are shared between the layers of a model (same edge_index, same mode).
GLOBAL_CACHE = GraphCache(max_entries=16)
