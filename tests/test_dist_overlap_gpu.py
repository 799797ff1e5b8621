"""The overlap's ordering (the halo all-to-all-v "must be overlapped
(start local-column rows first)"): two ranks on this box's one GPU over gloo run ``dist.halo_aggregate`` with
HIP events on the launch stream at the four points that define it.  What one GPU cannot show is the
transport's own time (RCCL over xGMI); what it can: that the interior rows' kernels are on the stream BEFORE
the rank waits for the exchange, the boundary rows' AFTER, and that the result is the unpartitioned one."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_interior_rows_are_enqueued_before_the_exchange_is_waited_for(cuda, tmp_path):
    out_path = str(tmp_path / "overlap.json")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29547",
           os.path.join(ROOT, "tests", "dist_overlap_worker_gpu.py"), out_path]
    res = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    recs = json.load(open(out_path))
    assert len(recs) == 2
    for r in recs:
        assert r["labels"] == ["exchange_issued", "interior_enqueued", "exchange_waited", "boundary_enqueued"], r
        # both classes of rows exist on this graph, so both launches did work
        assert 0 < r["n_boundary"] < r["n_local"] and r["n_halo"] > 0, r
        # device order = host order: every event completes no earlier than the one before it, and the
        # interior rows' kernels (normalise own rows + row-filtered aggregation) and the boundary rows'
        # (normalise halo rows + row-filtered aggregation) each occupy the stream for a measurable time
        assert all(g >= 0.0 for g in r["gaps_ms"]), r
        assert r["gaps_ms"][0] > 0.002 and r["gaps_ms"][2] > 0.002, r
        assert r["equal_to_whole"], r
