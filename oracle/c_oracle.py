"""ctypes front end of the plain-C oracle (oracle/sngnn_oracle.c).

TEST INFRASTRUCTURE ONLY (see oracle/sngnn_oracle.py's header; parity unpinned at
the third-party boundary).  Builds ``oracle/_build/libsngnn_oracle.so`` with gcc on
first use when it is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libsngnn_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sngnn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.sno_aggregate.restype = C.c_int64
        _lib.sno_aggregate.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                       C.c_int, C.c_int, C.c_int, C.c_double, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib.sno_adj_linear.restype = None
        _lib.sno_adj_linear.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                        C.c_void_p, C.c_int64, C.c_void_p]
        _lib.sno_cosine_dense.restype = None
        _lib.sno_cosine_dense.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def aggregate(h: np.ndarray, edge_index: np.ndarray, *, add_loops=True, remove_loops=False,
              top_k=None, thr=0.0, want_sel=True):
    """Same contract as ``sngnn_oracle.aggregate_reference`` on numpy arrays."""
    h = np.ascontiguousarray(h, dtype=np.float32)
    ei = np.ascontiguousarray(edge_index, dtype=np.int64)
    N, Cc = h.shape
    E = ei.shape[1]
    out = np.empty((N, Cc), np.float32)
    s = np.empty(E + N, np.float32)
    w = np.empty(E + N, np.float32)
    eo = np.empty(2 * (E + N), np.int64)
    k = -1 if top_k is None else int(top_k)
    sel = np.empty((N, max(k, 0)), np.int64) if (k >= 0 and want_sel) else None
    Ep = lib().sno_aggregate(_p(h), N, Cc, _p(ei), E, int(add_loops), int(remove_loops), k,
                             float(thr), _p(out), _p(s), _p(w), _p(sel), _p(eo))
    res = dict(out=out, s=s[:Ep].copy(), weight=w[:Ep].copy(),
               ei=np.stack([eo[:Ep], eo[Ep:2 * Ep]]))
    if sel is not None:
        res["sel_src"] = sel
    return res


def adj_linear(W: np.ndarray, b: np.ndarray, ei: np.ndarray, N: int) -> np.ndarray:
    W = np.ascontiguousarray(W, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    src = np.ascontiguousarray(ei[0], np.int64)
    dst = np.ascontiguousarray(ei[1], np.int64)
    Cc = W.shape[0]
    out = np.empty((N, Cc), np.float32)
    lib().sno_adj_linear(_p(W), _p(b), N, Cc, _p(src), _p(dst), src.shape[0], _p(out))
    return out


def cosine_dense(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    N, F = x.shape
    S = np.empty((N, N), np.float32)
    lib().sno_cosine_dense(_p(x), N, F, _p(S))
    return S
