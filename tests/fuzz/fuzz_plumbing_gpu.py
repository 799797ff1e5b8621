#!/usr/bin/env python3
"""Randomised check (not part of the suite) of the kernels around the aggregation: self.lin
forward / weight gradient (MFMA paths and fallbacks), the fused head (stand-alone and inside the
aggregation's launches) and the blend, on random shapes against PyTorch.  usage: fuzz_plumbing_gpu.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sngnn_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
t_end = time.time() + budget
cases = head_cases = 0
while time.time() < t_end:
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 15, 16, 17, 63, 1000, 1023, 1024, 1025, 4095, 4096, 4097, 5000, 16384, 30001]))
    if rng.integers(0, 2):
        n = int(rng.integers(1, 40000))
    f = int(rng.choice([16, 32, 64, 128, 1, 7, 33, 100, 129, 300]))
    c = int(rng.integers(1, 65)) if rng.integers(0, 4) else int(rng.choice([65, 70, 128]))
    tag = f"seed={seed} n={n} f={f} c={c}"
    try:
        gen = torch.Generator().manual_seed(seed)
        x = torch.randn(n, f, generator=gen).to(dev)
        lin = torch.nn.Linear(f, c).to(dev)
        gout = torch.randn(n, c, generator=gen).to(dev)
        ref = lin(x)
        (ref * gout).sum().backward()
        gw, gb = lin.weight.grad.clone(), lin.bias.grad.clone()
        lin.zero_grad()
        out = ops.linear(x, lin)
        (out * gout).sum().backward()
        assert (out - ref).abs().max() <= 3e-6 * max(1.0, float(ref.abs().max())), "linear forward"
        # sums over n rows of signed terms: judged against fp64 and the size of the terms
        gd, xd = gout.double(), x.double()
        tol_w = 2e-6 * (gd.abs().t() @ xd.abs()) + 1e-7
        assert ((lin.weight.grad.double() - gd.t() @ xd).abs() <= tol_w).all(), "wgrad"
        assert ((lin.bias.grad.double() - gd.sum(0)).abs() <= 2e-6 * gd.abs().sum(0) + 1e-7).all(), "bias grad"
        # lin with the F.normalize epilogue (round 3): h, unit rows, norms, filter rows bit for
        # bit what the separate calls give
        from sngnn_amd import _lib
        lib = _lib.load()
        if lib.sngnn_linear_normalized_supported(n, f, c):
            st = torch.cuda.current_stream().cuda_stream
            w_, b_ = lin.weight.detach().contiguous(), lin.bias.detach().contiguous()
            h0 = torch.empty(n, c, device=dev)
            _lib.check(lib.sngnn_linear_forward(x.data_ptr(), w_.data_ptr(), b_.data_ptr(), n, f, c, h0.data_ptr(), st), "lin")
            h1, un, nrm = torch.empty(n, c, device=dev), torch.empty(n, c, device=dev), torch.empty(n, device=dev)
            fb = ops.filter_row_bytes(c)
            filt = torch.empty((n, fb), dtype=torch.uint8, device=dev) if fb == 128 else None
            _lib.check(lib.sngnn_linear_forward_normalized(x.data_ptr(), w_.data_ptr(), b_.data_ptr(), n, f, c, h1.data_ptr(),
                                                           un.data_ptr(), nrm.data_ptr(), _lib.ptr(filt), st), "lin+norm")
            un2, nrm2, filt2 = ops.normalize_rows_filter(h0)
            assert torch.equal(h0, h1) and torch.equal(un, un2) and torch.equal(nrm, nrm2), "normalising epilogue"
            assert filt is None or torch.equal(filt, filt2), "filter rows of the epilogue"
        # head
        z = torch.randn(n, c, generator=gen).to(dev).requires_grad_(True)
        y = torch.randint(0, c, (n,), generator=gen).to(dev)
        mask = (torch.rand(n, generator=gen) < 0.6).to(dev)
        if int(mask.sum()) > 0:
            lref = F.nll_loss(F.log_softmax(z, 1)[mask], y[mask])
            lref.backward()
            gz = z.grad.clone()
            z.grad = None
            loss, corr = ops.head_nll(z, y, mask.to(torch.uint8), int(mask.sum()))
            loss.backward()
            assert abs(float(loss) - float(lref)) <= 3e-6 * max(1.0, abs(float(lref))), "head loss"
            assert int(corr) == int((z.detach()[mask].max(1)[1] == y[mask]).sum()), "head accuracy"
            assert (z.grad - gz).abs().max() <= 2e-6 * max(1e-3, float(gz.abs().max())) + 1e-9, "head grad"
        # the head inside the aggregation's launches (ops.HeadEpilogue) on a random graph over the same rows:
        # gradient rows bit-equal to the stand-alone head's on the stored logits, metrics to 2e-6
        if c % 4 == 0 and c <= 64 and 2 <= n <= 20000:
            from sngnn_amd.graph import Graph
            from tests.helpers import random_graph
            nh = int(rng.integers(0, 3))
            hubs = tuple((int(rng.integers(0, n)), int(rng.integers(1, n + 1))) for _ in range(nh))
            ei = random_graph(n, int(rng.integers(1, 8 * n)), seed=seed, hubs=hubs)
            gph = Graph(ei.to(dev), n, True, bool(rng.integers(0, 2)))
            kk = rng.choice([None, 1, 2, 16, 31])
            kk = None if kk is None else int(kk)
            thr_h = float(rng.choice([-1.5, 0.0, 0.3]))
            if ops.head_supported(gph, c, kk):
                hh = z.detach()
                bias_h = (torch.randn(c, generator=gen) * 0.3).to(dev) if rng.integers(0, 2) else None
                logits = ops.aggregate_forward(gph, hh, kk, thr_h)[0]
                if bias_h is not None:
                    logits = logits + bias_h
                m8 = mask.to(torch.uint8)
                nm = max(int(mask.sum()), 1)
                (l0, c0), g0 = ops.head_nll_with_grad(logits, y, m8, nm)
                met = torch.zeros(2, device=dev)
                gfused = ops.aggregate(hh, gph, kk, thr_h, None, None, bias_h, ops.HeadEpilogue(y, m8, met, nm, grad=True))
                assert torch.equal(gfused, g0), "head epilogue gradient"
                assert float(met[1]) == float(c0) and abs(float(met[0]) - float(l0)) <= 2e-6 * max(1.0, abs(float(l0))), \
                    "head epilogue metrics"
                sets = (torch.randint(0, 4, (n,), generator=gen)).to(torch.uint8).to(dev)
                na, nb = max(int((sets & 1).ne(0).sum()), 1), max(int((sets & 2).ne(0).sum()), 1)
                m4r = ops.head_nll2(logits, y, sets, na, nb)
                m4 = torch.zeros(4, device=dev)
                lg2 = ops.aggregate(hh, gph, kk, thr_h, None, None, bias_h, ops.HeadEpilogue(y, sets, m4, na, nb))
                assert torch.equal(lg2, logits), "head epilogue leaves the logits"
                assert float(m4[1]) == float(m4r[1]) and float(m4[3]) == float(m4r[3]), "head epilogue counts"
                for qq in (0, 2):
                    assert abs(float(m4[qq]) - float(m4r[qq])) <= 2e-6 * max(1.0, abs(float(m4r[qq]))), "head epilogue losses"
                head_cases += 1
        # blend
        o0 = torch.randn(n, c, generator=gen).to(dev).requires_grad_(True)
        o1 = torch.randn(n, c, generator=gen).to(dev).requires_grad_(True)
        beta = torch.tensor([float(rng.uniform(-0.5, 1.5))], device=dev, requires_grad=True)
        rb = beta * o0 + (1 - beta) * o1
        rb.backward(gout)
        want = (rb.detach().clone(), o0.grad.clone(), o1.grad.clone(), float(beta.grad))
        o0.grad = o1.grad = beta.grad = None
        ob = ops.blend(o0, o1, beta)
        ob.backward(gout)
        assert torch.equal(ob, want[0]) and torch.equal(o0.grad, want[1]) and torch.equal(o1.grad, want[2]), "blend"
        # d beta is a sum of n*c signed terms: judge it against the fp64 value and the terms' size
        terms = gout.double() * (o0.detach().double() - o1.detach().double())
        assert abs(float(beta.grad) - float(terms.sum())) <= 2e-6 * float(terms.abs().sum()) + 1e-6, "blend dbeta"
    except Exception as ex:      # noqa: BLE001
        print("FAIL", tag, "->", repr(ex)[:400], flush=True)
        sys.exit(1)
    cases += 1
    seed += 1
    if cases % 100 == 0:
        print(f"{cases} cases ok (last {tag})", flush=True)
print(f"done: {cases} random cases passed ({head_cases} with the head inside the aggregation), next seed {seed}")
