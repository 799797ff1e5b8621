"""GPU: the multi-GPU code path end to end with a one-rank RCCL group (the GPU box has
one card): partition graphs, the halo exchange's all_to_all_single / the all-gather +
reduce-scatter autograd seams, the flipped partition of SNGNN++ and the gradient all-reduce
must reproduce the single-GPU model."""
import os
import socket

import pytest
import torch
import torch.nn.functional as F

from sngnn_amd.synth import Data
from tests.helpers import random_graph

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_group(cuda):
    import torch.distributed as dist
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=cuda)
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["halo", "allgather"])
@pytest.mark.parametrize("kind,args", [
    ("SNGNN", lambda f, n: (f, 16, 5, 2)),
    ("SNGNN_Plus", lambda f, n: (f, 16, 5, n, 2, 4, 0.0, 1, 0.0)),
    ("SNGNN_Plus_Plus", lambda f, n: (f, 16, 5, n, 2, 4, 0.0, 0.4, 1, 0.0)),
])
def test_partition_path_with_one_rank_equals_single_gpu(cuda, one_rank_group, kind, args, exchange):
    import sngnn_amd
    from sngnn_amd import dist as sd
    n, f = 500, 24
    ei = random_graph(n, 5000, seed=2, hubs=((0, 499), (3, 150)))
    ei = torch.cat([ei, torch.tensor([[0], [1]])], 1)
    ei = torch.unique(ei, dim=1).to(cuda)
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(n, f, generator=gen).to(cuda)
    y = torch.randint(0, 5, (n,), generator=gen).to(cuda)
    torch.manual_seed(3)
    single = getattr(sngnn_amd, kind)(*args(f, n)).to(cuda)
    torch.manual_seed(3)
    multi = getattr(sngnn_amd, kind)(*args(f, n)).to(cuda)
    single.eval()
    multi.eval()
    data = Data(x=x, edge_index=ei)
    F.nll_loss(single(data), y).backward()
    part = sd.Partition(0, 1, n, exchange=exchange)
    sd.set_partition(part)
    try:
        out = multi(data)
        F.nll_loss(out, y).backward()
        sd.allreduce_grads(multi, part)
    finally:
        sd.set_partition(None)
    assert torch.allclose(out, single(data), rtol=1e-5, atol=1e-6)
    for (name, p), (_, q) in zip(multi.named_parameters(), single.named_parameters()):
        assert (p.grad - q.grad).abs().max() <= 1e-5 * q.grad.abs().max() + 1e-9, name
