"""CPU, world_size 2, gloo: the node-range partition bookkeeping and the
all-gather / reduce-scatter autograd seam.  The local aggregation is played by the
oracle (the HIP kernels need a GPU); what is checked is that
    all_gather(h) -> aggregate owned rows -> backward through reduce-scatter
equals the single-process result, forward and gradient."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import sngnn_oracle as O
from tests.helpers import random_graph


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_local, C, k, thr, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sngnn_amd import dist as sd
    part = sd.Partition(rank, world, n_local)
    assert (part.n_total, part.row_begin, part.row_end) == (world * n_local, rank * n_local,
                                                            (rank + 1) * n_local)
    n = part.n_total
    ei = random_graph(n, 40 * n // 4, seed=9, hubs=((1, n - 1),))
    gen = torch.Generator().manual_seed(0)
    h_all = torch.randn(n, C, generator=gen)
    gout = torch.randn(n, C, generator=gen)
    lin = torch.nn.Linear(C, C)
    torch.manual_seed(3)
    lin.reset_parameters()

    # this rank's shard
    x_loc = h_all[part.row_begin:part.row_end].clone()
    h_loc = lin(x_loc)
    h_full = sd.all_gather_rows(h_loc, part)
    assert h_full.shape == (n, C)
    ei_p = O.sn_edge_list(ei, n, True, True)
    mine = (ei_p[1] >= part.row_begin) & (ei_p[1] < part.row_end)
    norm = torch.nn.functional.normalize(h_full, dim=-1)
    out_all, *_ = O.propagate_mean(h_full, norm, ei_p[:, mine], k, thr)
    out_loc = out_all[part.row_begin:part.row_end]
    (out_loc * gout[part.row_begin:part.row_end]).sum().backward()
    sd.allreduce_grads(lin, part)

    # single-process reference
    lin2 = torch.nn.Linear(C, C)
    lin2.load_state_dict({k_: v.detach().clone() for k_, v in lin.state_dict().items()})
    h2 = lin2(h_all)
    out2, *_ = O.propagate_mean(h2, torch.nn.functional.normalize(h2, dim=-1), ei_p, k, thr)
    (out2 * gout).sum().backward()
    ok_fwd = torch.allclose(out_loc, out2[part.row_begin:part.row_end], rtol=1e-5, atol=1e-6)
    ok_bwd = all(torch.allclose(p.grad, p2.grad, rtol=1e-4, atol=1e-5)
                 for p, p2 in zip(lin.parameters(), lin2.parameters()))
    q.put((rank, bool(ok_fwd), bool(ok_bwd)))
    dist.barrier()
    dist.destroy_process_group()


def test_partitioned_aggregation_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 30, 6, 3, 0.0, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True, True), (1, True, True)]


def test_set_partition_switch():
    from sngnn_amd import dist as sd
    assert sd.current_partition() is None
    p = sd.Partition(1, 4, 10)
    sd.set_partition(p)
    assert sd.current_partition() is p and p.row_begin == 10 and p.n_total == 40
    sd.set_partition(None)
    assert sd.current_partition() is None
