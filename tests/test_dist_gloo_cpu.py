"""CPU, world_size 2, gloo: the node-range partition bookkeeping (uneven ranges), the halo
plan, and both exchanges' autograd seams.  The local aggregation is played by the oracle
(the HIP kernels need a GPU); what is checked is that
    exchange(h) -> aggregate owned rows -> backward through the exchange's transpose
equals the single-process result, forward and gradient - for the halo exchange (only the
referenced rows travel) and for the full all-gather, on a node count the ranks do not divide."""
import os
import socket

import pytest
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


def _local_aggregate(table, ei_local, n_local, k, thr, remove_loops):
    """The rank's conv on its local table: loops appended for the OWNED rows only (what
    sngnn_graph_create_partition does), reference order kept."""
    loops = torch.arange(n_local)
    ei = torch.cat([ei_local, torch.stack([loops, loops])], dim=1)
    if remove_loops:
        ei = ei[:, ei[0] != ei[1]]
    norm = torch.nn.functional.normalize(table, dim=-1)
    out_all, *_ = O.propagate_mean(table, norm, ei, k, thr)
    return out_all[:n_local]


def _worker(rank, world, port, bounds, C, k, thr, exchange, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sngnn_amd import dist as sd
    part = sd.Partition(rank, world, bounds=bounds, exchange=exchange)
    n = part.n_total
    assert part.n_local == bounds[rank + 1] - bounds[rank] and part.row_begin == bounds[rank]
    ei = random_graph(n, 40 * n // 4, seed=9, hubs=((1, n - 1), (n - 2, n // 2)))
    gen = torch.Generator().manual_seed(0)
    h_all = torch.randn(n, C, generator=gen)
    gout = torch.randn(n, C, generator=gen)
    lin = torch.nn.Linear(C, C)
    torch.manual_seed(3)
    lin.reset_parameters()
    r0, r1 = part.row_begin, part.row_end

    # this rank's shard
    h_loc = lin(h_all[r0:r1].clone())
    info = {}
    if exchange == "halo":
        plan = sd.HaloPlan(ei, part)
        table = sd.halo_exchange(h_loc, plan)
        assert table.shape == (part.n_local + plan.n_halo, C)
        # the halo holds exactly the remote sources of my in-edges, grouped by owner, ascending
        mine = (ei[1] >= r0) & (ei[1] < r1)
        remote = torch.unique(ei[0][mine & ((ei[0] < r0) | (ei[0] >= r1))])
        assert torch.equal(plan.halo_ids, remote)
        assert sum(plan.recv_counts) == plan.n_halo and plan.recv_counts[rank] == 0
        info["halo_rows"] = plan.n_halo
        out_loc = _local_aggregate(table, plan.edge_index, part.n_local, k, thr, True)
    else:
        h_full = sd.all_gather_rows(h_loc, part)
        assert h_full.shape == (n, C)
        ei_p = O.sn_edge_list(ei, n, True, True)
        mine = (ei_p[1] >= r0) & (ei_p[1] < r1)
        norm = torch.nn.functional.normalize(h_full, dim=-1)
        out_all, *_ = O.propagate_mean(h_full, norm, ei_p[:, mine], k, thr)
        out_loc = out_all[r0:r1]
    (out_loc * gout[r0:r1]).sum().backward()
    sd.allreduce_grads(lin, part)

    # single-process reference
    lin2 = torch.nn.Linear(C, C)
    lin2.load_state_dict({k_: v.detach().clone() for k_, v in lin.state_dict().items()})
    h2 = lin2(h_all)
    ei_p = O.sn_edge_list(ei, n, True, True)
    out2, *_ = O.propagate_mean(h2, torch.nn.functional.normalize(h2, dim=-1), ei_p, k, thr)
    (out2 * gout).sum().backward()
    ok_fwd = torch.allclose(out_loc, out2[r0:r1], rtol=1e-5, atol=1e-6)
    ok_bwd = all(torch.allclose(p.grad, p2.grad, rtol=1e-4, atol=1e-5)
                 for p, p2 in zip(lin.parameters(), lin2.parameters()))
    q.put((rank, bool(ok_fwd), bool(ok_bwd), info))
    dist.barrier()
    dist.destroy_process_group()


# uneven ranges (61 nodes: nothing divides; the padded all-gather / reduce-scatter), even ranges (the
# plain all_gather_into_tensor / reduce_scatter_tensor), three ranks of which one owns NO node
# (zero-row blocks in every collective)
@pytest.mark.parametrize("bounds", [(0, 37, 61), (0, 30, 60), (0, 25, 25, 61)], ids=["uneven", "even", "empty_rank"])
@pytest.mark.parametrize("exchange", ["halo", "allgather"])
def test_partitioned_aggregation_equals_single_process(exchange, bounds):
    """The collectives are the calls the GPUs make (all_to_all_single with split sizes,
    all_gather_into_tensor, reduce_scatter_tensor): sngnn_amd/dist.py has one branch for every backend."""
    world, port = len(bounds) - 1, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bounds, 6, 3, 0.0, exchange, q))
             for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[:3] for r in res] == [(r, True, True) for r in range(world)]


def _bn_worker(rank, world, port, bounds, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from sngnn_amd import dist as sd
    part = sd.Partition(rank, world, bounds=bounds)
    gen = torch.Generator().manual_seed(1)
    x_all = torch.randn(bounds[-1], 5, generator=gen) * 3 + 1
    g_all = torch.randn(bounds[-1], 5, generator=gen)
    torch.manual_seed(0)
    bn, bn2 = torch.nn.BatchNorm1d(5), torch.nn.BatchNorm1d(5)
    with torch.no_grad():
        for m in (bn, bn2):
            m.weight.copy_(torch.tensor([1., 2., .5, -1., 3.]))
            m.bias.copy_(torch.tensor([0., 1., -1., .5, 2.]))
    r0, r1 = part.row_begin, part.row_end
    x = x_all[r0:r1].clone().requires_grad_(True)
    y = sd.sync_batch_norm(bn, x, part)
    (y * g_all[r0:r1]).sum().backward()
    sd.allreduce_grads(bn, part)
    x2 = x_all.clone().requires_grad_(True)
    y2 = bn2(x2)
    (y2 * g_all).sum().backward()
    ok = (torch.allclose(y, y2[r0:r1], rtol=1e-5, atol=1e-5)
          and torch.allclose(x.grad, x2.grad[r0:r1], rtol=1e-4, atol=1e-5)
          and torch.allclose(bn.weight.grad, bn2.weight.grad, rtol=1e-4, atol=1e-5)
          and torch.allclose(bn.bias.grad, bn2.bias.grad, rtol=1e-4, atol=1e-5)
          and torch.allclose(bn.running_mean, bn2.running_mean, rtol=1e-5, atol=1e-6)
          and torch.allclose(bn.running_var, bn2.running_var, rtol=1e-5, atol=1e-6)
          and int(bn.num_batches_tracked) == int(bn2.num_batches_tracked))
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sync_batch_norm_equals_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bn_worker, args=(r, world, port, (0, 23, 40), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_partition_bounds():
    from sngnn_amd import dist as sd
    # BASELINE config 5: ogbn-products over the 8 GPUs of a node - 2 449 029 is not divisible
    b = sd.Partition.even_bounds(2449029, 8)
    assert b[0] == 0 and b[-1] == 2449029 and len(b) == 9
    sizes = [y - x for x, y in zip(b, b[1:])]
    assert max(sizes) - min(sizes) == 1 and sum(sizes) == 2449029
    p = sd.Partition.even(5, 8, 2449029)
    assert (p.row_begin, p.row_end, p.n_local, p.n_total) == (b[5], b[6], sizes[5], 2449029)
    # edge-balanced: a hub pulls its range's end forward
    deg = torch.tensor([5, 0, 0, 100, 1, 1, 1, 1, 50, 3])
    eb = sd.Partition.edge_balanced_bounds(deg, 3)
    assert eb[0] == 0 and eb[-1] == 10 and list(eb) == sorted(eb)
    cost = [int((deg[x:y] + 1).sum()) for x, y in zip(eb, eb[1:])]
    assert max(cost) <= int((deg + 1).sum()) * 0.7
    with pytest.raises(ValueError):
        sd.Partition(0, 2, bounds=(0, 5))
    with pytest.raises(ValueError):
        sd.Partition(0, 2, bounds=(1, 5, 9))


def test_set_partition_switch():
    from sngnn_amd import dist as sd
    assert sd.current_partition() is None
    p = sd.Partition(1, 4, 10)
    sd.set_partition(p)
    assert sd.current_partition() is p and p.row_begin == 10 and p.n_total == 40
    sd.set_partition(None)
    assert sd.current_partition() is None


def _ckpt_worker(rank, world, port, bounds, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sngnn_amd
    from sngnn_amd import dist as sd
    n = bounds[-1]
    args = (6, 8, 3, n, 2, 2, 0.0, 0.5, 1, 0.0)
    torch.manual_seed(5)
    single = sngnn_amd.SNGNN_Plus_Plus(*args)                 # the reference-shaped model
    part = sd.Partition(rank, world, bounds=bounds)
    sd.set_partition(part)
    torch.manual_seed(5)
    sharded = sngnn_amd.SNGNN_Plus_Plus(*args)                # w holds this rank's columns only
    ok = sharded.lins[0].w.weight.shape == (8, part.n_local)
    full = sd.full_state_dict(sharded, part)                  # collective: gathers the column shards
    ref = single.state_dict()
    ok = ok and list(full) == list(ref) and all(torch.equal(full[k], ref[k]) for k in ref)
    # ... and the reference-format checkpoint loads back into a model sharded differently
    other = sd.Partition(rank, world, bounds=(0, 9, n))
    sd.set_partition(other)
    torch.manual_seed(99)
    again = sngnn_amd.SNGNN_Plus_Plus(*args)
    again.load_state_dict(full)
    ok = ok and all(torch.equal(again.lins[l].w.weight, ref[f"lins.{l}.w.weight"][:, other.row_begin:other.row_end])
                    for l in (0, 1))
    sd.set_partition(None)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_w_checkpoint_round_trip():
    """per-rank shards of ``w.weight`` share one state_dict key - ``full_state_dict``
    gathers them into the reference's [C, N] layout, and such a checkpoint loads into a model
    sharded over any partition (or into a single-process model)."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ckpt_worker, args=(r, world, port, (0, 23, 40), q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]
