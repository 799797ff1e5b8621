import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from sngnn_amd import _lib, toolbox as T
lib = _lib.load()
dev = torch.device("cuda:0")
for n, f in ((169343, 128), (100000, 64), (100000, 96), (100000, 32), (40000, 128)):
    x = torch.randn(n, f, device=dev)
    for route, name in ((3, "128x128 tiles, 1 wave/SIMD"), (4, "256x64 tiles, 2 waves/SIMD")):
        lib.sngnn_tuning_set(6, route)
        T.knn_graph(x, 16); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): T.knn_graph(x, 16)
        torch.cuda.synchronize()
        print(f"N={n} F={f}: {name}: {(time.perf_counter() - t0) / 3 * 1e3:8.2f} ms", flush=True)
lib.sngnn_tuning_set(6, 0)
