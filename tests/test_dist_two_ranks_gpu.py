"""The multi-GPU design end to end on the real kernels: two ranks (node-range partition,
all-gather of h forward, summed partial grad_h backward, all-reduced parameter gradients)
reproduce the single-process result - outputs, loss and every parameter gradient.  The two
ranks share this box's one GPU and talk over gloo; the collectives' RCCL form is covered by
tests/test_dist_world1_gpu.py."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.dist_case import build_case, build_model

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("kind", ["SNGNN_Plus", "SNGNN_Plus_Plus", "SNGNN", "AGNN"])
def test_two_ranks_equal_one_process(cuda, kind, tmp_path):
    from sngnn_amd.synth import Data
    world = 2
    out_path = str(tmp_path / "ranks.npz")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", "29541",
           os.path.join(ROOT, "tests", "dist_worker_gpu.py"), kind, out_path]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    z = np.load(out_path)

    x, ei, y, mask, _ = build_case(world)
    model = build_model(kind, x.size(1), x.size(0)).to(cuda)
    model.train()
    out = model(Data(x=x.to(cuda), edge_index=ei.to(cuda)))
    loss = F.nll_loss(out[mask.to(cuda)], y.to(cuda)[mask.to(cuda)])
    loss.backward()
    assert np.allclose(z["out"], out.detach().cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert abs(float(z["loss"]) - float(loss)) <= 1e-5 * max(1.0, abs(float(loss)))
    for k, p in model.named_parameters():
        g, w = z["grad." + k], p.grad.detach().cpu().numpy()
        assert np.abs(g - w).max() <= 2e-5 * max(np.abs(w).max(), 1e-6), k
