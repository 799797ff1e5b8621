"""Worker of tests/test_dist_two_ranks_gpu.py (one process per rank, launched with
torch.distributed.run): one training-mode forward + backward of a model under a node-range
partition, HIP kernels on the GPU, collectives over gloo (the ranks share the one GPU of
the test box; on a real node the same code runs over RCCL)."""
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
from tests.dist_case import build_case, build_model  # noqa: E402


def main():
    kind, out_path = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    x, ei, y, mask, n_local = build_case(world)
    model = build_model(kind, x.size(1), x.size(0)).to(dev)
    part = sd.Partition(rank, world, n_local)
    sd.set_partition(part)
    r0, r1 = part.row_begin, part.row_end
    model.train()
    out = model(Data(x=x[r0:r1].to(dev), edge_index=ei.to(dev)))        # full edge list: the
    m = mask[r0:r1].to(dev)                                              # partition filters it
    loss = F.nll_loss(out[m], y[r0:r1].to(dev)[m], reduction="sum") / float(mask.sum())
    loss.backward()
    sd.allreduce_grads(model, part)
    outs = [torch.empty_like(out) for _ in range(world)]
    dist.all_gather(outs, out.detach().contiguous())
    tot = loss.detach().clone()
    dist.all_reduce(tot)
    if rank == 0:
        np.savez(out_path, out=torch.cat(outs).cpu().numpy(), loss=tot.cpu().numpy(),
                 **{"grad." + k: p.grad.detach().cpu().numpy() for k, p in model.named_parameters()})
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
