#!/usr/bin/env python3
"""The dense cosine kernel (toolbox.cosine_similarity_dense_small: fp32 MFMA, upper triangle +
mirror) against the library route F.normalize + rocBLAS SGEMM at the shapes DESIGN.md 4.5
quotes.  Rates are priced on the flops each side DOES: the hand-written kernel computes the
tiles on and above the diagonal (2 F 128^2 per tile), the library the full 2 N^2 F product.
Run on the GPU box:  python tools/bench_cosine_cmp.py"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sngnn_amd import synth, toolbox  # noqa: E402

dev = torch.device("cuda:0")
PEAK = 157.3          # TFLOP/s, dense fp32 MFMA (MI355X_MICROARCH.md); the hand-written kernel's products run on the
                      # bf16 matrix cores (exact split, 8 partial products): its column is fp32-equivalent flops / this peak


def timed(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def lib(x):
    n = F.normalize(x, p=2., dim=-1)
    return n.mm(n.t())


shapes = [(name,) + tuple(synth.make_dataset(name).x.shape) for name in ("cora", "chameleon", "actor")]
shapes += [("", 20000, 128), ("", 20000, 512), ("", 8192, 1024), ("", 16384, 256), ("", 3000, 33), ("", 5000, 7)]
print(f"{'shape':28s} {'hand-written':>14s} {'of peak (done)':>15s} {'norm + rocBLAS':>15s} {'of peak':>8s}  max |diff|")
for name, n, f in shapes:
    x = (synth.make_dataset(name).x if name else torch.randn(n, f, generator=torch.Generator().manual_seed(n))).to(dev)
    ours = timed(lambda: toolbox.cosine_similarity_dense_small(x))
    ref = timed(lambda: lib(x))
    nb = (n + 127) // 128
    done = 2.0 * f * 128 * 128 * (nb * (nb + 1) // 2)
    diff = (toolbox.cosine_similarity_dense_small(x) - lib(x)).abs().max().item()
    line = (f"{name:10s} {n:6d} x {f:5d}    {ours:9.3f} ms  {done / ours / 1e9 / PEAK * 100:11.1f} %  "
            f"{ref:11.3f} ms  {2.0 * n * n * f / ref / 1e9 / PEAK * 100:6.1f} %  {diff:.1e}")
    if os.environ.get("KS"):          # sweep of the contraction split (sngnn_tuning_set(7, ks)): "ks:ms" pairs
        from sngnn_amd import _lib
        for ks in (int(v) for v in os.environ["KS"].split(",")):
            _lib.load().sngnn_tuning_set(7, ks)
            line += f"  ks{ks}:{timed(lambda: toolbox.cosine_similarity_dense_small(x)):.3f}"
        _lib.load().sngnn_tuning_set(7, 0)
    print(line, flush=True)
