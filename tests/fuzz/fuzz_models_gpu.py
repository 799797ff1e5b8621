#!/usr/bin/env python3
"""Randomised drop-in check at model level: SNGNN / SNGNN_Plus / SNGNN_Plus_Plus / AGNN built
with random constructor arguments on random graphs; same seeded parameters as the oracle's
restated classes, forward log-probs and every parameter gradient compared.
usage: fuzz_models_gpu.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import sngnn_amd  # noqa: E402
from oracle import sngnn_oracle as O  # noqa: E402
from sngnn_amd.synth import Data  # noqa: E402
from tests.helpers import random_graph  # noqa: E402

dev = torch.device("cuda:0")
if os.environ.get("SNGNN_FUZZ_FIN"):      # where split rows are finalized (sngnn_tuning_set knob 9; fuzz_gpu.py)
    from sngnn_amd import _lib  # noqa: E402
    _lib.load().sngnn_tuning_set(9, int(os.environ["SNGNN_FUZZ_FIN"]))
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
cases = skipped = flipped = ref32_off = 0


def grads_match_fp64(kind, args, seed, x, ei, y, mod):
    """The arbiter when the fp32 oracle and the GPU disagree on a gradient: the same oracle in
    float64.  True when every GPU gradient is within the tolerance of the fp64 one (i.e. the
    fp32 CPU autograd was the inaccurate side - a row whose norm nearly vanishes behind a
    ReLU / batch norm makes F.normalize's backward cancel catastrophically in fp32)."""
    torch.manual_seed(seed)
    ref64 = getattr(O, kind)(*args)
    if kind in ("SNGNN", "AGNN"):
        ref64.dropout.p = 0.0
    ref64 = ref64.double().train()
    F.nll_loss(ref64(Data(x=x.double(), edge_index=ei)), y).backward()
    for p64, q in zip(ref64.parameters(), mod.parameters()):
        if p64.grad is None:
            continue
        sc = max(p64.grad.abs().max().item(), 1e-5)
        if (q.grad.cpu().double() - p64.grad).abs().max().item() > 2e-3 * sc + 5e-6:
            return False
    return True

while time.time() < t_end:
    rng = np.random.default_rng(seed)
    n = int(rng.integers(20, 500))
    f = int(rng.integers(3, 48))
    hid = int(rng.choice([4, 8, 16, 20, 32, 47]))
    classes = int(rng.integers(2, 11))
    layers = int(rng.integers(1, 4))
    k = int(rng.choice([1, 2, 3, 5, 8, 16, 40]))
    thr = float(rng.choice([-1.5, 0.0, 0.1, 0.5]))
    rem = int(rng.integers(0, 2))
    bn = bool(rng.integers(0, 2))
    kind = str(rng.choice(["SNGNN", "SNGNN_Plus", "SNGNN_Plus_Plus", "AGNN"]))
    hubs = tuple((int(rng.integers(0, n)), int(rng.integers(1, n))) for _ in range(int(rng.integers(0, 3))))
    ei = random_graph(n, int(rng.integers(n, 10 * n)), seed=seed, hubs=hubs)
    if kind == "SNGNN_Plus_Plus":                       # the reference's `row - row.min()` needs node 0
        ei = torch.unique(torch.cat([ei, torch.tensor([[0], [n - 1]])], 1), dim=1)   # as a source
    gen = torch.Generator().manual_seed(seed)
    x = torch.randn(n, f, generator=gen)
    y = torch.randint(0, classes, (n,), generator=gen)
    args = {"SNGNN": (f, hid, classes, layers, bn), "AGNN": (f, hid, classes, layers, bn),
            "SNGNN_Plus": (f, hid, classes, n, layers, k, thr, rem, 0.0, bn),
            "SNGNN_Plus_Plus": (f, hid, classes, n, layers, k, thr, float(rng.uniform(0, 1)), rem, 0.0, bn)}[kind]
    tag = f"seed={seed} {kind}{args} n={n} E={ei.size(1)}"
    try:
        torch.manual_seed(seed)
        ref = getattr(O, kind)(*args)
        torch.manual_seed(seed)
        mod = getattr(sngnn_amd, kind)(*args)
        for (ka, va), (kb, vb) in zip(ref.state_dict().items(), mod.state_dict().items()):
            assert ka == kb and torch.equal(va, vb), f"init {ka}"
        mod = mod.to(dev)
        if kind in ("SNGNN", "AGNN"):                   # their dropout is fixed at 0.5: switch it off
            ref.dropout.p = mod.dropout.p = 0.0
        ref.train(), mod.train()
        out_r = ref(Data(x=x, edge_index=ei))
        out_g = mod(Data(x=x.to(dev), edge_index=ei.to(dev)))
        if not torch.isfinite(out_r).all():
            skipped += 1
            seed += 1
            continue
        # A selection layer fed by a ReLU sees rows with very few non-zero channels, whose
        # cosines tie (or nearly tie) at +-1: the reference breaks such ties by the last ulp of
        # ITS arithmetic, this library by the last ulp of its own (tests/helpers.py near-tie
        # rule).  Deep selecting models are therefore compared row-wise with a small budget of
        # rows that may differ (measured: ~0.5 % of such models show any); everything else must agree
        # everywhere.
        # (also a 1-layer model whose conv is only 2-3 classes wide: in so few dimensions a
        # neighbour is easily parallel to the node itself, and ties its self-loop at cosine 1)
        tie_prone = kind in ("SNGNN_Plus", "SNGNN_Plus_Plus") and (layers >= 2 or classes <= 4)
        diff = (out_g.cpu() - out_r).abs()
        tol = 2e-4 * max(1.0, out_r.abs().max().item())
        bad_rows = int((diff.max(dim=1).values > tol).sum())
        if tie_prone:
            assert bad_rows <= max(2, n // 10), f"{bad_rows} of {n} rows differ"
            if bad_rows:
                flipped += 1
        else:
            assert bad_rows == 0, f"log-probs err {diff.max().item():.2e} in {bad_rows} rows"
        if bad_rows == 0:
            F.nll_loss(out_r, y).backward()
            F.nll_loss(out_g, y.to(dev)).backward()
            for (name, p), q in zip(ref.named_parameters(), mod.parameters()):
                if p.grad is None:
                    continue
                ge = (q.grad.cpu() - p.grad).abs().max().item()
                sc = max(p.grad.abs().max().item(), 1e-5)
                gt = 2e-3 * sc + 5e-6                      # (+ fp32 noise floor: some gradients vanish)
                if tie_prone and ge > gt:                  # a tie that flipped without moving the output
                    flipped += 1
                    break
                if ge > gt and grads_match_fp64(kind, args, seed, x, ei, y, mod):
                    ref32_off += 1                         # the GPU agrees with the fp64 oracle
                    break
                assert ge <= gt, f"grad {name}: err {ge:.2e} scale {sc:.2e}"
    except Exception as ex:      # noqa: BLE001
        print("FAIL", tag, "->", repr(ex)[:300], flush=True)
        sys.exit(1)
    cases += 1
    seed += 1
    if cases % 50 == 0:
        print(f"{cases} cases ok (last {tag})", flush=True)
print(f"done: {cases} random models passed ({skipped} skipped, {flipped} with near-tie flips in deep selecting models, "
      f"{ref32_off} where the fp32 oracle's gradient was off and the fp64 oracle sided with the GPU), next seed {seed}")
