"""Device time of the classification head's launches (sngnn_head_nll / _nll2) at arxiv size, replayed
from a HIP graph of 20 calls (the eager loop is bound by the host)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from sngnn_amd import ops    # noqa: E402

dev = torch.device("cuda:0")
N, C = 169343, 40
z = torch.randn(N, C, device=dev)
y = torch.randint(0, C, (N,), device=dev)
r = torch.rand(N, device=dev)
m8 = (r < 0.6).to(torch.uint8)
sets = ((r >= 0.6) & (r < 0.8)).to(torch.uint8) + 2 * (r >= 0.8).to(torch.uint8)
nm, na, nb = int(m8.sum()), int((sets == 1).sum()), int((sets == 2).sum())


def graphed(f, calls=20, reps=30):
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        f()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    cg = torch.cuda.CUDAGraph()
    with torch.cuda.graph(cg, stream=side):
        for _ in range(calls):
            f()
    for _ in range(3):
        cg.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        cg.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (reps * calls) * 1e6


with torch.no_grad():
    t1 = graphed(lambda: ops.head_nll(z, y, m8, nm))
    t2 = graphed(lambda: ops.head_nll2(z, y, sets, na, nb))
    t3 = graphed(lambda: ops.head_nll_with_grad(z, y, m8, nm))
print(f"head (kernel + reduce, graph replay): eval {t1:6.2f} us  eval-two {t2:6.2f} us  train(grad) {t3:6.2f} us")
