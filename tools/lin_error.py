"""Rounding error of self.lin's forward against float64: products on the bf16 matrix cores (exact
three-way split, default) beside the fp32 MFMAs, relative to sum |x w| (the scale an fp32 dot
product's error bound is stated on)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sngnn_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.load()
for n, f, c in [(169343, 128, 40), (50000, 64, 64), (50000, 32, 32)]:
    g = torch.Generator().manual_seed(f)
    x = torch.randn(n, f, generator=g)
    w = torch.randn(c, f, generator=g) / f ** 0.5
    b = torch.randn(c, generator=g) * 0.1
    ref = x.double() @ w.double().t() + b.double()
    mag = x.double().abs() @ w.double().abs().t() + b.double().abs()
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for mode, name in ((0, "bf16 split"), (1, "fp32 MFMA")):
        lib.sngnn_tuning_set(5, mode)
        h = torch.empty(n, c, device=dev)
        _lib.check(lib.sngnn_linear_forward(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), n, f, c, h.data_ptr(), st), "lin")
        e = (h.cpu().double() - ref).abs() / mag
        res[name] = (e.max().item(), e.mean().item())
    lib.sngnn_tuning_set(5, 0)
    t = (torch.nn.functional.linear(xd, wd, bd).cpu().double() - ref).abs() / mag
    res["torch (rocBLAS)"] = (t.max().item(), t.mean().item())
    print(f"{n} x {f} -> {c}: " + "; ".join(f"{k}: max {v[0] / 2 ** -24:.2f} mean {v[1] / 2 ** -24:.3f} ulp(2^-24)" for k, v in res.items()))
