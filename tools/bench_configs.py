#!/usr/bin/env python3
"""Per-config latencies of BASELINE.json's configs 1-3 (small graphs: launch-bound,
parity/plumbing configs) and the toolbox MFMA cosine.  Run on the GPU box."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sngnn_amd  # noqa: E402
from sngnn_amd import synth, toolbox  # noqa: E402
from sngnn_amd import train as T  # noqa: E402

dev = torch.device("cuda:0")


def timed(fn, reps=50, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


CONFIGS = [
    ("cora", "SNGNN", lambda f, n, c: (f, 32, c, 1)),
    ("chameleon", "SNGNN_Plus", lambda f, n, c: (f, 32, c, n, 1, 10, 0.9, 1, 0.5)),
    ("chameleon", "SNGNN_Plus", lambda f, n, c: (f, 32, c, n, 2, 10, 0.9, 1, 0.5)),
    ("actor", "SNGNN_Plus_Plus", lambda f, n, c: (f, 32, c, n, 1, 10, 0.9, 0.0, 1, 0.5)),
    ("actor", "SNGNN_Plus_Plus", lambda f, n, c: (f, 32, c, n, 2, 10, 0.9, 0.0, 1, 0.5)),
]
print(f"{'dataset':10s} {'model':16s} layers  fwd_ms  fwd+bwd_ms  epoch_eager_ms  epoch_graph_ms")
for name, kind, mk in CONFIGS:
    data = synth.make_dataset(name).to(dev)
    n, f = data.x.shape
    c = synth.num_classes(name)
    args = mk(f, n, c)
    torch.manual_seed(1234)
    model = getattr(sngnn_amd, kind)(*args).to(dev)
    model.eval()
    with torch.no_grad():
        fwd = timed(lambda: model(data))
    model.train()

    def fb():
        model.zero_grad()
        torch.nn.functional.nll_loss(model(data)[data.train_mask], data.y[data.train_mask]).backward()
    fwbw = timed(fb, reps=30)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)

    def ep():
        T.train_step(model, data, opt)
        T.eval_step(model, data, data.val_mask)
        T.eval_step(model, data, data.test_mask)
    eager = timed(ep, reps=20)
    torch.manual_seed(1234)
    model2 = getattr(sngnn_amd, kind)(*args).to(dev)
    opt2 = torch.optim.Adam(model2.parameters(), lr=0.01, weight_decay=5e-4)
    ge = T.GraphedEpoch(model2, data, opt2)
    graphed = timed(ge.run, reps=30)
    layers = args[3] if kind == "SNGNN" else args[4]
    print(f"{name:10s} {kind:16s} {layers:6d}  {fwd:6.3f}  {fwbw:10.3f}  {eager:14.3f}  {graphed:14.3f}",
          flush=True)

# the kernel computes the upper triangle of the symmetric S and mirrors it: N (N + 128) F flop are
# DONE (tiles of 128 on and above the diagonal); the rate is priced on those, not on the nominal
# 2 N^2 F of a full product (which would read above 100 % of the peak at Actor's size)
print("\ndense cosine S = n n^T (bf16 matrix cores after an exact split, fp32 rounding): fp32-equivalent flops done = 2 F 128^2 x (upper-triangle tiles)")
for name in ("cora", "chameleon", "actor"):
    data = synth.make_dataset(name)
    x = data.x.to(dev)
    n, f = x.shape
    ms = timed(lambda: toolbox.cosine_similarity_dense_small(x), reps=10, warm=2)
    nb = (n + 127) // 128
    done = 2.0 * f * 128 * 128 * (nb * (nb + 1) // 2)
    print(f"{name:10s} N={n} F={f}: {ms:7.3f} ms  {done / ms / 1e9:7.1f} TFLOP/s on the flops done "
          f"({done / ms / 1e9 / 157.3 * 100:4.1f} % of the 157.3 TF fp32 MFMA peak; nominal 2 N^2 F: "
          f"{2 * n * n * f / 1e9:.1f} GF)", flush=True)
    y = data.y.to(dev)
    ms = timed(lambda: toolbox.class_similarity_dense_small(x, y), reps=10, warm=2)
    print(f"{'':10s} class similarity without S: {ms:7.3f} ms", flush=True)
