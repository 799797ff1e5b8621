"""Worker of tests/test_dist_two_ranks_gpu.py (one process per rank, launched with
torch.distributed.run): one training-mode forward + backward of a model under a node-range
partition (uneven ranges), HIP kernels on the GPU, collectives over gloo (the ranks share the
one GPU of the test box; on a real node the same code runs over RCCL).

    dist_worker_gpu.py KIND OUT.npz EXCHANGE SHARDED
EXCHANGE: halo | allgather.  SHARDED = 1: the model is BUILT under the partition, so
SNGNN++'s ``w`` holds only the rank's own columns."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sngnn_amd  # noqa: E402
from sngnn_amd import dist as sd  # noqa: E402
from sngnn_amd.synth import Data  # noqa: E402
from tests.dist_case import BOUNDS, build_case, build_model  # noqa: E402


def main():
    kind, out_path, exchange, sharded = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    # SNGNN_DIST_BACKEND=nccl: one rank per GPU over RCCL (needs >= world GPUs); default: the ranks
    # share GPU 0 and talk over gloo
    backend = os.environ.get("SNGNN_DIST_BACKEND", "gloo")
    dev_id = int(os.environ.get("LOCAL_RANK", "0")) if backend == "nccl" else 0
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group("gloo")
    x, ei, y, mask, _ = build_case(world)
    part = sd.Partition(rank, world, bounds=BOUNDS[world], exchange=exchange)
    if sharded:
        sd.set_partition(part)
    model = build_model(kind, x.size(1), x.size(0)).to(dev)
    sd.set_partition(part)
    r0, r1 = part.row_begin, part.row_end
    model.train()
    out = model(Data(x=x[r0:r1].to(dev), edge_index=ei.to(dev)))        # full edge list: the
    m = mask[r0:r1].to(dev)                                              # partition filters it
    loss = F.nll_loss(out[m], y[r0:r1].to(dev)[m], reduction="sum") / float(mask.sum())
    loss.backward()
    sd.allreduce_grads(model, part)

    def gather_rows(t):          # [n_local, ...] shards of uneven length -> [n_total, ...]
        mx = max(part.sizes)
        pad = t.new_zeros((mx,) + tuple(t.shape[1:]))
        pad[:t.size(0)] = t
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad.contiguous())
        return torch.cat([p[:n] for p, n in zip(parts, part.sizes)])

    outs = gather_rows(out.detach())
    tot = loss.detach().clone()
    dist.all_reduce(tot)
    grads = {}
    for k, p in model.named_parameters():
        g = p.grad.detach()
        if getattr(p, "_sngnn_sharded", False):      # [C, n_local] column shards -> [C, N]
            g = gather_rows(g.t().contiguous()).t()
        grads["grad." + k] = g.cpu().numpy()
    bufs = {"buf." + k: b.detach().cpu().numpy() for k, b in model.named_buffers()}
    if rank == 0:
        np.savez(out_path, out=outs.cpu().numpy(), loss=tot.cpu().numpy(), **grads, **bufs)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
