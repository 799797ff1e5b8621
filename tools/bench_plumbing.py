#!/usr/bin/env python3
"""Tuning aid: the kernels around the aggregation inside an epoch at arxiv size
(self.lin forward, its weight gradient, the fused classification head), each against
the bytes it has to move."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sngnn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 169343))


def timed(fn, reps=100, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


for F, C in ((128, 40), (128, 32), (32, 40), (64, 64), (16, 7)):
    x = torch.randn(N, F, device=dev)
    lin = torch.nn.Linear(F, C).to(dev)
    g = torch.randn(N, C, device=dev)
    with torch.no_grad():
        out0 = ops.linear(x, lin)
        t_f = timed(lambda: ops.linear(x, lin))
        t_b = timed(lambda: torch.nn.functional.linear(x, lin.weight, lin.bias))
        if C % 4 == 0:       # the same launch with the F.normalize epilogue (unit rows + norms [+ filter rows])
            def lin_norm(filt):
                u = ops.UnitRows(filt)
                ops._Linear.apply(x, lin.weight, lin.bias, None, None, u, None)
                return u
            t_n = timed(lambda: lin_norm(False))
            t_nf = timed(lambda: lin_norm(True)) if C > 32 else float("nan")
            t_pass = timed(lambda: ops.normalize_rows(out0))
        else:
            t_n = t_nf = t_pass = float("nan")
    out = ops.linear(x, lin)

    def wg():
        lin.zero_grad(set_to_none=True)
        out.backward(g, retain_graph=True)
    t_w = timed(wg, reps=50)
    mb_f = (N * F + N * C) * 4 / 1e6
    print(f"F={F:4d} C={C:3d}  lin fwd {t_f:6.1f} us (rocBLAS {t_b:6.1f}; {mb_f:.0f} MB -> {mb_f / 8e6 * 1e6:5.1f} us at 8 TB/s)"
          f"   wgrad {t_w:6.1f} us   | lin + normalise epilogue {t_n:6.1f} us (+ filter rows {t_nf:6.1f}); "
          f"separate normalise pass {t_pass:5.1f} us", flush=True)

C = 40
z = torch.randn(N, C, device=dev)
y = torch.randint(0, C, (N,), device=dev)
mask = (torch.rand(N, device=dev) < 0.6)
m8 = mask.to(torch.uint8)
nm = int(mask.sum())
with torch.no_grad():
    t_h = timed(lambda: ops.head_nll(z, y, m8, nm))
zz = z.clone().requires_grad_(True)
t_hg = timed(lambda: ops.head_nll(zz, y, m8, nm))
print(f"head C={C}: eval {t_h:6.1f} us, with grad {t_hg:6.1f} us   ({N * C * 4 / 1e6:.0f} MB logits)")
