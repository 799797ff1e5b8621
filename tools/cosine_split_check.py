#!/usr/bin/env python3
"""Measurement aid: the dense cosine at Chameleon's / Cora's size with the contraction split forced
(sngnn_tuning_set(7, ks); 0 = the library's own rule), alternating, same process."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sngnn_amd import _lib, synth, toolbox  # noqa: E402

lib = _lib.load()
dev = torch.device("cuda:0")
for name in ("chameleon", "cora"):
    x = synth.make_dataset(name).x.to(dev)
    for rnd in range(3):
        line = f"{name} round {rnd}:"
        for ks in (0, 1, 2, 3):
            lib.sngnn_tuning_set(7, ks)
            for _ in range(5):
                toolbox.cosine_similarity_dense_small(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                toolbox.cosine_similarity_dense_small(x)
            torch.cuda.synchronize()
            line += f"  ks={ks}: {(time.perf_counter() - t0) / 50 * 1e3:.3f} ms"
        print(line, flush=True)
lib.sngnn_tuning_set(7, 0)
