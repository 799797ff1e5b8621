"""Event counters of the kNN builder's selection (measurement build: hipcc -DSNGNN_KNN_EXP=4 knn.hip, loaded through
SNGNN_LIB_PATH): tiles x waves, registers passing the fast test, registers with a real candidate, candidates parked
by their lane, merges - per scan at arxiv size."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sngnn_amd import _lib, toolbox as T  # noqa: E402

lib = _lib.load()
x = torch.randn(int(os.environ.get("N", 169343)), 128, device="cuda:0")
lib.sngnn_knn_debug_counters.argtypes = [C.c_void_p, C.c_int]
buf = (C.c_ulonglong * 8)()
lib.sngnn_knn_debug_counters(None, 1)
T.knn_graph(x, 16)
torch.cuda.synchronize()
lib.sngnn_knn_debug_counters(buf, 0)
tw, passed, real, parked, merges = (int(v) for v in buf[:5])
print(f"tiles x waves {tw}; registers passing the fast test {passed} ({passed / tw:.2f} per tile-wave); with a real candidate "
      f"{real} ({real / tw:.2f}); parked by one lane {parked}; merges {merges} ({merges / tw:.3f} per tile-wave)")
