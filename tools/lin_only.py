"""Tuning aid: 200 calls of self.lin's forward at arxiv size, plain and with the normalising
epilogue (run under tools/profile_cmd.sh for rocprofv3 kernel times)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sngnn_amd import ops  # noqa: E402
from sngnn_amd._lib import load  # noqa: E402

if os.environ.get("LIN_MODE"):      # 1 = fp32 matrix instructions instead of the bf16 split
    load().sngnn_tuning_set(5, int(os.environ["LIN_MODE"]))

dev = torch.device("cuda:0")
f, c = int(os.environ.get("F", 128)), int(os.environ.get("C", 40))
n = int(os.environ.get("N", 169343))
x = torch.randn(n, f, device=dev)
lin = torch.nn.Linear(f, c).to(dev)
with torch.no_grad():
    for _ in range(200):
        ops.linear(x, lin)
    for _ in range(200):
        ops._Linear.apply(x, lin.weight, lin.bias, None, None, ops.UnitRows(False), None)
torch.cuda.synchronize()
