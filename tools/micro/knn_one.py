import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from sngnn_amd import toolbox as T
dev = torch.device("cuda:0")
x = torch.randn(169343, 128, device=dev)
T.knn_graph(x, 16); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): T.knn_graph(x, 16)
torch.cuda.synchronize()
print(os.environ.get("SNGNN_LIB_PATH", "product"), "arxiv-size kNN %.2f ms" % ((time.perf_counter() - t0) / 3 * 1e3), flush=True)
