"""ctypes binding of ``libsngnn_hip.so`` (the C ABI declared in include/sngnn_hip.h).

The product path has NO fallback: if the shared library is missing or a call
fails, an exception is raised.  (The CPU oracle under ``oracle/`` is test
infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SNGNN_LIB_PATH: an experimental build of the same library for an A/B measurement; the product is the in-tree file)
LIB_PATH = os.environ.get("SNGNN_LIB_PATH") or os.path.join(_HERE, "libsngnn_hip.so")

OK, EINVAL, ERANGE, EHIP, ENOMEM = 0, -1, -2, -3, -4
UNSELECTED = -4.0
MAX_CHANNELS = 512

_vp, _i64, _i32, _f32 = C.c_void_p, C.c_int64, C.c_int, C.c_float

# name -> (restype, argtypes); mirrors include/sngnn_hip.h one to one
SIGNATURES = {
    "sngnn_last_error": (C.c_char_p, []),
    "sngnn_build_info": (C.c_char_p, []),
    "sngnn_graph_create": (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, C.POINTER(_vp)]),
    "sngnn_graph_create_partition": (_i32, [_vp, _i64, _i64, _i64, _i64, _i32, _i32, _vp,
                                            C.POINTER(_vp)]),
    "sngnn_graph_destroy": (None, [_vp]),
    "sngnn_graph_num_nodes": (_i64, [_vp]),
    "sngnn_graph_num_total_nodes": (_i64, [_vp]),
    "sngnn_graph_row_offset": (_i64, [_vp]),
    "sngnn_graph_num_edges": (_i64, [_vp]),
    "sngnn_graph_max_in_degree": (_i64, [_vp]),
    "sngnn_graph_src_min": (_i64, [_vp]),
    "sngnn_graph_num_fused_nodes": (_i64, [_vp]),
    "sngnn_graph_workspace_bytes": (_i64, [_vp, _i32]),
    "sngnn_graph_copy_array": (_i32, [_vp, _i32, _vp]),
    "sngnn_graph_array_dev": (_vp, [_vp, _i32]),
    "sngnn_agg_forward": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_normalize_rows": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp]),
    "sngnn_filter_row_bytes": (_i64, [_i32]),
    "sngnn_normalize_rows_filter": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _vp]),
    "sngnn_agg_forward_prepared": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp]),
    "sngnn_agg_forward_epilogue": (_i32, [_vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_agg_forward_prepared_epilogue": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp,
                                                   _vp]),
    "sngnn_filter_enable": (_i32, [_i32]),
    "sngnn_filter_wanted": (_i32, [_vp, _i32, _i32, _f32]),
    "sngnn_agg_forward_rows": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _f32, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_tuning_set": (_i32, [_i32, _i32]),
    "sngnn_last_forward_finalize_workgroups": (_i32, []),
    "sngnn_filter_pair_scores": (_i32, [_vp, _i32, _vp, _vp, _i64, _vp, _vp]),
    "sngnn_agg_forward_normalized": (_i32, [_vp, _vp, _vp, _i32, _i32, _f32, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _vp]),
    "sngnn_agg_backward": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_agg_backward_topk": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sngnn_agg_kept_bits_supported": (_i32, [_vp, _i32, _i32]),
    "sngnn_agg_head_supported": (_i32, [_vp, _i32, _i32]),
    "sngnn_agg_head_workspace_bytes": (_i64, [_vp]),
    "sngnn_graph_kept_bits_bytes": (_i64, [_vp]),
    "sngnn_agg_backward_bits": (_i32, [_vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sngnn_attn_forward": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    "sngnn_attn_backward": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_signed_forward": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_signed_backward": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_blend_workspace_bytes": (_i64, []),
    "sngnn_blend_forward": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "sngnn_blend_backward": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_blend_forward_epilogue": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "sngnn_blend_backward_epilogue": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _f32, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_knn_workspace_bytes": (_i64, [_i64, _i32]),
    "sngnn_knn_graph": (_i32, [_vp, _i64, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "sngnn_gather_sum_rows": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sngnn_scatter_sum_rows": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "sngnn_weighted_gather_sum_rows": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sngnn_weighted_scatter_sum_rows": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sngnn_pair_dot_rows": (_i32, [_vp, _vp, _vp, _vp, _i64, _i32, _vp, _vp]),
    "sngnn_profile_enable": (_i32, [_i32]),
    "sngnn_profile_last_forward": (_i32, [C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_f32), C.POINTER(_f32)]),
    "sngnn_gather_floor_workspace_bytes": (_i64, []),
    "sngnn_gather_floor": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "sngnn_adj_linear_forward": (_i32, [_vp, _vp, _vp, _i32, _vp, _vp, _vp]),
    "sngnn_adj_linear_backward": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp]),
    "sngnn_head_workspace_bytes": (_i64, [_i64]),
    "sngnn_head_nll": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _vp, _vp, _vp, _vp]),
    "sngnn_head_nll2": (_i32, [_vp, _vp, _vp, _i64, _i32, _i64, _i64, _vp, _vp, _vp]),
    "sngnn_head_nll_blend_supported": (_i32, [_i32]),
    "sngnn_head_nll_blend": (_i32, [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32, _i64, _i64, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_linear_forward": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp]),
    "sngnn_linear_forward_masked": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _f32, _vp, _vp]),
    "sngnn_epilogue_backward": (_i32, [_vp, _vp, _f32, _i64, _vp, _vp]),
    "sngnn_linear_normalized_supported": (_i32, [_i64, _i32, _i32]),
    "sngnn_linear_forward_normalized": (_i32, [_vp, _vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "sngnn_linear_wgrad_workspace_bytes": (_i64, [_i64, _i32, _i32]),
    "sngnn_linear_wgrad": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp, _vp, _vp, _vp]),
    "sngnn_cosine_dense": (_i32, [_vp, _i64, _i64, _vp, _vp]),
    "sngnn_cosine_class_sums": (_i32, [_vp, _i64, _i64, _vp, _i32, _vp, _vp, _vp]),
    "sngnn_edge_cosine": (_i32, [_vp, _i64, _i64, _vp, _i64, _vp, _vp]),
    "sngnn_segment_mean": (_i32, [_vp, _vp, _i64, _i64, _vp, _vp, _vp]),
    "sngnn_sparse_pair_dot": (_i32, [_vp, _vp, _vp, _i64, _vp, _vp, _i64, _vp, _vp]),
}



class Epilogue(C.Structure):
    """``sngnn_epilogue_t`` (include/sngnn_hip.h)."""
    _fields_ = [("bias", C.c_void_p), ("keep", C.c_void_p), ("keep_scale", C.c_float), ("relu", C.c_int),
                ("seed", C.c_void_p), ("p", C.c_float), ("kept_bits", C.c_void_p),
                ("head_y", C.c_void_p), ("head_sel", C.c_void_p), ("head_sets", C.c_int), ("head_out_mode", C.c_int),
                ("head_n_a", C.c_int64), ("head_n_b", C.c_int64), ("head_metrics", C.c_void_p),
                ("head_workspace", C.c_void_p), ("no_filter", C.c_int)]


_lib = None


class SngnnError(RuntimeError):
    pass


def load():
    """Load the library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SngnnError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                f"g.build()'` or `make -C sngnn_amd/csrc` (there is no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().sngnn_last_error().decode("utf-8", "replace")
        exc = ValueError if rc in (EINVAL, ERANGE) else SngnnError
        raise exc(f"{what} failed ({rc}): {msg}")


def ptr(t):
    """Device/host pointer of a tensor (or None)."""
    return None if t is None else t.data_ptr()
