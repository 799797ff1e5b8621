"""Seeded synthetic stand-ins for the datasets of BASELINE.json's configs.

The reference loads its graphs through PyG dataset classes that need a network
(datasets/datasets.py:16-304); there is none here, so graphs of the published
sizes are generated as SURVEY.md section 8(d) prescribes: power-law in-degree
(alpha ~ 2.1) clipped to the dataset's maximum, uniform sources, de-duplicated
and sorted by (src, dst) like PyG's ``coalesce`` (datasets/datasets.py:170,221,
284), ~0.3 % self-loops, >= 1 % zero-in-degree nodes, 60/20/20 masks.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

# name -> (N, E, F, classes, max in-degree, feature kind, density)
SHAPES = {
    "cora":      (2708, 10556, 1433, 7, 168, "binary", 0.013),
    "chameleon": (2277, 36101, 2325, 5, 732, "binary", 0.010),
    "actor":     (7600, 30019, 932, 5, 1296, "binary", 0.006),
    "arxiv":     (169343, 1166243, 128, 40, 13000, "normal", 1.0),
    "products":  (2449029, 123718280, 100, 47, 17000, "normal", 1.0),
}


@dataclass
class Data:
    """Minimal stand-in for ``torch_geometric.data.Data`` as the reference's models
    and trainer use it (models.py:77,202,294; train.py:81-83,399-409)."""
    x: torch.Tensor
    edge_index: torch.Tensor
    y: Optional[torch.Tensor] = None
    train_mask: Optional[torch.Tensor] = None
    val_mask: Optional[torch.Tensor] = None
    test_mask: Optional[torch.Tensor] = None

    @property
    def num_nodes(self) -> int:
        return self.x.size(0)

    def to(self, device):
        kw = {}
        for k, v in self.__dict__.items():
            kw[k] = v.to(device) if isinstance(v, torch.Tensor) else v
        return Data(**kw)


def powerlaw_degrees(rng: np.random.Generator, n: int, e: int, max_deg: int,
                     alpha: float = 2.1, zero_frac: float = 0.02) -> np.ndarray:
    """In-degree sequence summing to exactly ``e``: Pareto-like tail clipped at
    ``max_deg``, ``zero_frac`` of the nodes forced to zero, then rescaled."""
    u = rng.random(n)
    raw = (1.0 - u) ** (-1.0 / (alpha - 1.0))          # Pareto(alpha-1), >= 1
    raw[rng.random(n) < zero_frac] = 0.0
    raw = np.minimum(raw, max_deg)
    top = int(np.argmax(raw))
    deg = None
    scale = e / max(raw.sum(), 1.0)
    for _ in range(40):                                  # fixed-point on the clip
        deg = np.minimum(np.floor(raw * scale), max_deg)
        deg[(raw > 0) & (deg < 1)] = 1                   # only the forced zeros are isolated
        deg[top] = max_deg                               # pin the published maximum
        tot = deg.sum()
        if abs(tot - e) <= max(1, e // 100000):
            break
        scale *= e / max(tot, 1.0)
    deg = deg.astype(np.int64)
    diff = int(e - deg.sum())
    nz = np.flatnonzero((deg > 0) & (deg < max_deg))
    while diff != 0:
        step = min(abs(diff), nz.size)
        pick = rng.choice(nz, size=step, replace=False)
        if diff > 0:
            deg[pick] += 1
        else:
            pick = pick[deg[pick] > 1]
            deg[pick] -= 1
            step = pick.size
        diff = int(e - deg.sum())
        nz = np.flatnonzero((deg > 0) & (deg < max_deg))
    return deg


def make_edges(rng: np.random.Generator, n: int, e: int, max_deg: int,
               loop_frac: float = 0.003, n_src: Optional[int] = None,
               dst_offset: int = 0, uniform: bool = False) -> np.ndarray:
    """Directed edge list [2, ~e] int64 sorted by (src, dst), de-duplicated.
    ``n_src``/``dst_offset`` let a rank generate only the edges that point into its
    own node range of a larger graph (sources are global ids)."""
    n_src = n if n_src is None else n_src
    if uniform:
        deg = np.full(n, e // n, dtype=np.int64)
    else:
        deg = powerlaw_degrees(rng, n, e, min(max_deg, n_src - 1))
    dst = np.repeat(np.arange(n, dtype=np.int64), deg) + dst_offset
    src = rng.integers(0, n_src, size=dst.size, dtype=np.int64)
    n_loop = int(np.ceil(loop_frac * dst.size))
    if n_loop:
        where = rng.choice(dst.size, size=n_loop, replace=False)
        src[where] = dst[where]
    key = np.unique(src * (n_src + 1) + dst)             # coalesce: sort + dedup
    for _ in range(8):                                   # top up what dedup removed
        miss = dst.size - key.size
        if miss <= 0:
            break
        d2 = rng.choice(dst, size=miss)                  # degree-proportional targets
        s2 = rng.integers(0, n_src, size=miss, dtype=np.int64)
        key = np.unique(np.concatenate([key, s2 * (n_src + 1) + d2]))
    return np.stack([key // (n_src + 1), key % (n_src + 1)])


def make_features(rng: np.random.Generator, n: int, f: int, kind: str,
                  density: float) -> np.ndarray:
    if kind == "normal":
        return rng.standard_normal((n, f), dtype=np.float32)
    x = (rng.random((n, f), dtype=np.float32) < density).astype(np.float32)
    dup = rng.choice(n, size=max(2, n // 50), replace=False)   # exact duplicate rows
    x[dup[1::2][: dup[0::2].size]] = x[dup[0::2][: dup[1::2].size]]
    return x


def make_dataset(name: str, seed: int = 1234, *, uniform: bool = False,
                 scale: float = 1.0, with_features: bool = True) -> Data:
    n, e, f, c, max_deg, kind, dens = SHAPES[name]
    if scale != 1.0:
        n, e = max(16, int(n * scale)), max(16, int(e * scale))
        max_deg = max(4, min(max_deg, n // 2))
    rng = np.random.default_rng(seed)
    ei = make_edges(rng, n, e, max_deg, uniform=uniform)
    x = make_features(rng, n, f, kind, dens) if with_features else np.zeros((n, 1), np.float32)
    y = rng.integers(0, c, size=n, dtype=np.int64)
    r = rng.random(n)
    tr, va = r < 0.6, (r >= 0.6) & (r < 0.8)
    return Data(x=torch.from_numpy(x), edge_index=torch.from_numpy(ei),
                y=torch.from_numpy(y), train_mask=torch.from_numpy(tr),
                val_mask=torch.from_numpy(va), test_mask=torch.from_numpy(~(tr | va)))


def num_classes(name: str) -> int:
    return SHAPES[name][3]
