"""CPU: the C-ABI library loads and exports every symbol include/sngnn_hip.h declares;
argument validation that needs no GPU; the product has no CPU fallback."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sngnn_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sngnn_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_exported_and_bound():
    from sngnn_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in sngnn_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes signatures out of sync with the header"
    assert b"gfx950" in lib.sngnn_build_info()


def test_constants_match_header():
    from sngnn_amd import _lib
    text = open(os.path.join(ROOT, "include", "sngnn_hip.h")).read()
    assert f"#define SNGNN_MAX_CHANNELS  {_lib.MAX_CHANNELS}" in text
    assert "#define SNGNN_UNSELECTED   (-4.0f)" in text and _lib.UNSELECTED == -4.0


def test_null_and_bad_arguments_are_rejected_without_a_gpu():
    from sngnn_amd import _lib
    lib = _lib.load()
    assert lib.sngnn_graph_create(None, 0, 0, 1, 0, None, None) == _lib.EINVAL
    assert b"NULL" in lib.sngnn_last_error()
    h = C.c_void_p()
    assert lib.sngnn_graph_create(None, 5, 10, 1, 0, None, C.byref(h)) == _lib.EINVAL
    assert lib.sngnn_graph_create_partition(None, 0, 10, 4, 2, 1, 0, None, C.byref(h)) == _lib.EINVAL
    assert lib.sngnn_agg_forward(None, None, 8, 1, 0.0, None, None, None, None, None, None,
                                 None) == _lib.EINVAL
    assert lib.sngnn_graph_num_nodes(None) == -1
    lib.sngnn_graph_destroy(None)          # no-op


def test_no_cpu_fallback():
    from sngnn_amd import SNConv_plus, ops
    from sngnn_amd.graph import Graph
    conv = SNConv_plus(4, 3, 10)
    x = torch.randn(10, 4)
    ei = torch.randint(0, 10, (2, 20))
    with pytest.raises(ValueError, match="GPU"):
        conv(x, ei)
    with pytest.raises(ValueError, match="GPU"):
        Graph(ei, 10, True, False)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "sngnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "sngnn_oracle" not in src, f


def test_epilogue_struct_layout_matches_the_header(tmp_path):
    """``_lib.Epilogue`` (ctypes) against ``sngnn_epilogue_t`` as a C compiler lays it out: size and
    the offset of every field (a mismatch would hand the library garbage pointers silently)."""
    import ctypes as C
    import shutil
    import subprocess
    from sngnn_amd import _lib
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        pytest.skip("no C compiler")
    fields = [name for name, _ in _lib.Epilogue._fields_]
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "sngnn_hip.h"', 'int main(void) {',
             '  printf("size %zu\\n", sizeof(sngnn_epilogue_t));']
    lines += [f'  printf("{f} %zu\\n", offsetof(sngnn_epilogue_t, {f}));' for f in fields]
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call([cc, "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = dict(line.split() for line in subprocess.check_output([str(exe)], text=True).splitlines())
    assert int(out["size"]) == C.sizeof(_lib.Epilogue)
    for f in fields:
        assert int(out[f]) == getattr(_lib.Epilogue, f).offset, f


def test_hot_path_sources_hold_no_host_synchronisation():
    """forward / backward only enqueue.  The translation units behind the conv layers'
    forward and backward entries (aggregation, attention, signed propagation, adjacency branch, blend,
    ``lin`` / head) call no blocking HIP API; the one exception is the profiling read-back
    ``sngnn_profile_last_forward`` (bench.py's roofline leg, never on a model's path).  Graph construction
    (graph.hip), the toolbox statistics and the kNN builder are setup / analysis calls and may block."""
    hot = ("agg_fwd.hip", "agg_fwd_impl.h", "agg_fwd_filter.h", "agg_bwd.hip", "agg_bwd_impl.h", "attn.hip",
           "attn_impl.h", "signed.hip", "signed_impl.h", "adj_linear.hip", "adj_linear_impl.h", "blend.hip",
           "linear.hip", "head.hip", "head_row.h", "device_utils.h")
    blocking = re.compile(r"\bhip(StreamSynchronize|DeviceSynchronize|EventSynchronize|Memcpy|MemcpyDtoH|Malloc|Free|"
                          r"HostMalloc|MemcpyAsync)\s*\(")
    for name in hot:
        src = open(os.path.join(ROOT, "sngnn_amd", "csrc", name)).read()
        for m in blocking.finditer(src):
            line = src[:m.start()].count("\n") + 1
            fn_start = src.rfind('extern "C"', 0, m.start())
            owner = src[fn_start:src.find("(", fn_start)] if fn_start >= 0 else ""
            assert name == "agg_fwd.hip" and "sngnn_profile_last_forward" in owner, (name, line, m.group(0))
