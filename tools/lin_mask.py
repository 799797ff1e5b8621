import sys, time, torch
sys.path.insert(0, '/root/repo')
from sngnn_amd import _lib
lib = _lib.load()
dev = torch.device("cuda:0")
n, f, c = 169343, 40, 64
g = torch.randn(n, f, device=dev); w = torch.randn(c, f, device=dev); act = torch.randn(n, c, device=dev)
out = torch.empty(n, c, device=dev)
st = torch.cuda.current_stream().cuda_stream
def t(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e6
print("plain ", t(lambda: lib.sngnn_linear_forward(g.data_ptr(), w.data_ptr(), None, n, f, c, out.data_ptr(), st)))
ref = out.clone()
print("masked", t(lambda: lib.sngnn_linear_forward_masked(g.data_ptr(), w.data_ptr(), None, n, f, c, act.data_ptr(), 2.0, out.data_ptr(), st)))
print("equal:", torch.equal(out, torch.where(act > 0, ref * 2.0, torch.zeros_like(ref))))
print("where ", t(lambda: torch.where(act > 0, ref * 2.0, torch.zeros((), device=dev))))
