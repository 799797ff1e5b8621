"""Seeded inputs shared by tests/test_dist_two_ranks_gpu.py and its worker processes."""
import torch

from tests.helpers import random_graph


BOUNDS = {2: (0, 520, 900)}          # uneven node ranges


def build_case(world: int):
    n_local = 450
    n = n_local * world
    ei = random_graph(n, 9000, seed=21, hubs=((3, 500), (n - 2, 300), (n_local + 1, 150)))
    loops = torch.arange(0, n, 5)
    ei = torch.unique(torch.cat([ei, torch.stack([loops, loops])], 1), dim=1)
    gen = torch.Generator().manual_seed(9)
    x = torch.randn(n, 24, generator=gen)
    y = torch.randint(0, 6, (n,), generator=gen)
    mask = torch.rand(n, generator=gen) < 0.6
    return x, ei, y, mask, n_local


def build_model(kind: str, f: int, n: int):
    import sngnn_amd
    torch.manual_seed(77)
    if kind == "SNGNN_Plus":
        return sngnn_amd.SNGNN_Plus(f, 16, 6, n, 2, 4, 0.05, 1, 0.0)
    if kind == "SNGNN_Plus_bn":          # batch norm between the layers (and conv bias, models.py:177-178)
        return sngnn_amd.SNGNN_Plus(f, 16, 6, n, 2, 4, 0.05, 1, 0.0, True)
    if kind == "SNGNN_Plus_Plus":
        return sngnn_amd.SNGNN_Plus_Plus(f, 16, 6, n, 2, 4, 0.05, 0.4, 1, 0.0)
    if kind == "SNGNN":
        return sngnn_amd.SNGNN(f, 16, 6, 1)
    if kind == "AGNN":
        return sngnn_amd.AGNN(f, 16, 6, 1)
    raise ValueError(kind)
