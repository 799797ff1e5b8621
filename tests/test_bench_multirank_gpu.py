"""bench.py's N > 1 code path (node-range partition per rank, all-gather of h every step,
max-over-ranks timing, summed edges) rehearsed with two ranks on ONE GPU over gloo: the
driver runs the real thing (one rank per GPU, RCCL) on an 8-GPU node this box does not have."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_rehearsal(cuda):
    env = dict(os.environ, SNGNN_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "3", "--warmup", "1", "--scale", "0.05", "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]           # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    n, e = d["config"]["nodes_per_gpu"], d["config"]["edges_per_gpu"]
    # whole-job value: the edges of BOTH ranks over the slowest rank's time
    total_edges = d["value"] * d["ms_per_step"] * 1e-3
    assert 1.5 * e < total_edges < 2.5 * e and n > 0
    assert d["roofline"]["achieved"] > 0 and "partition" in d["config"]["parallelism"]


def test_two_rank_rehearsal_of_the_products_workload(cuda):
    """``bench.py --workload products --gpus 2``: the STRONG-scaling branch (one products-shaped graph cut in
    two node ranges, the ++ layer with ``w`` sharded by node range: the flipped-list halo plan, ``gather_sum``
    over the [own | halo] table of W^T rows, the blend) has run at least once before a driver launches it on
    eight GPUs.  Rehearsal: ranks share the GPU, collectives over gloo, numbers mean nothing."""
    env = dict(os.environ, SNGNN_BENCH_REHEARSAL="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29535", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--workload", "products", "--steps", "2", "--warmup", "1", "--scale", "0.01",
           "--no-cpu-baseline"]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["channels"] == 48
    assert "w sharded by node range" in d["config"]["parallelism"]
    assert d["exchange"]["form"] == "halo" and d["exchange"]["rows_received_per_rank"] > 0
    assert d["value"] > 0 and d["roofline_layer"]["algorithmic_bytes"] > d["roofline"]["algorithmic_bytes"]
