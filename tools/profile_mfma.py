#!/usr/bin/env python3
"""The matrix-core kernels of the repo, each launched a few times at the sizes DESIGN.md
quotes, for `rocprofv3 --kernel-trace --stats` / `--pmc` (tools/profile_mfma.sh):
  k_cosine_mfma   dense cosine S = n n^T      (SimGFAToolbox/dense.py:138-141)  Chameleon / Cora / Actor sizes
  k_knn_mfma      kNN similarity-graph builder (fused top-k epilogue)            20 000 x 128, 169 343 x 128
  k_linear_rows   self.lin forward             (models.py:121,237,324)           169 343 x 128 -> 40
  k_wgrad_mfma    its weight gradient                                            same
Prints each kernel's wall time and the flop count the MFMA rate is priced on."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sngnn_amd import ops, toolbox  # noqa: E402

dev = torch.device("cuda:0")
REPS = int(os.environ.get("REPS", 5))
FP32_MFMA_PEAK = 157.3e12


def timed(fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


g = torch.Generator(device=dev).manual_seed(0)
for name, n, f in (("chameleon", 2277, 2325), ("cora", 2708, 1433), ("actor", 7600, 932)):
    x = (torch.rand(n, f, generator=g, device=dev) < 0.02).float()
    t = timed(lambda: toolbox.cosine_similarity_dense_small(x))
    fl = 2.0 * n * n * f
    # the kernel computes the upper triangle only (N^2 F flops) and mirrors it: "nominal" prices
    # the whole call at the reference's 2 N^2 F (what a full GEMM would do), "done" at what ran
    print(f"k_cosine_mfma {name:10s} N={n:6d} F={f:5d}: {t * 1e6:9.1f} us (whole call: pad + norms + tiles)  "
          f"nominal {fl / t / 1e12:6.1f} TFLOP/s on 2 N^2 F = {fl / 1e9:.1f} GF;  "
          f"done {fl / 2 / t / 1e12:6.1f} TFLOP/s = {fl / 2 / t / FP32_MFMA_PEAK:5.1%} of the fp32 MFMA peak", flush=True)

for n, f, k in ((2277, 2325, 10), (7600, 932, 10), (20000, 128, 16), (169343, 128, 16)):
    x = torch.randn(n, f, generator=g, device=dev)
    t = timed(lambda: toolbox.knn_graph(x, k), reps=3 if n > 50000 else REPS)
    fl = 2.0 * n * n * f
    print(f"k_knn_mfma    N={n:6d} F={f} k={k}: {t * 1e3:9.2f} ms  {fl / t / 1e12:6.1f} TFLOP/s "
          f"({fl / t / FP32_MFMA_PEAK:5.1%}; 2 N^2 F = {fl / 1e12:.2f} TF)", flush=True)

n, f, c = 169343, 128, 40
x = torch.randn(n, f, generator=g, device=dev)
lin = torch.nn.Linear(f, c).to(dev)
gout = torch.randn(n, c, generator=g, device=dev)
with torch.no_grad():
    t = timed(lambda: ops.linear(x, lin), reps=50)
print(f"k_linear_rows {n}x{f}->{c}: {t * 1e6:7.1f} us  ({2.0 * n * f * 48 / t / 1e12:5.1f} TFLOP/s on 48 padded columns)", flush=True)
out = ops.linear(x, lin)


def wg():
    lin.zero_grad(set_to_none=True)
    out.backward(gout, retain_graph=True)


t = timed(wg, reps=50)
print(f"k_wgrad_mfma  {n}x{f}, C={c}: {t * 1e6:7.1f} us (with k_sum_partials)  ({2.0 * n * f * c / t / 1e12:5.1f} TFLOP/s)", flush=True)
