#!/usr/bin/env python3
"""Golden vectors of the cosine-attention mode (AGNNConv, models.py:377-405) from the
CPU oracle's restatement (NOT the reference itself, see oracle/sngnn_oracle.py).
Run from the repo root:  python tests/golden/make_golden_attn.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import sngnn_oracle as O  # noqa: E402
from tests.helpers import random_graph  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (n, e, C, hubs)
    "c40": (600, 6000, 40, ((0, 599), (3, 180), (9, 60))),
    "c7": (300, 1500, 7, ((5, 200),)),
}

for name, (n, e, C, hubs) in CASES.items():
    ei = random_graph(n, e, seed=len(name) * 77 + n, hubs=hubs)
    loops = torch.arange(1, n, 4)
    ei = torch.unique(torch.cat([ei, torch.stack([loops, loops])], 1), dim=1)
    gen = torch.Generator().manual_seed(n + C)
    h = torch.randn(n, C, generator=gen)
    h[5] = h[6]
    h[8] = 0.0
    gout = torch.randn(n, C, generator=gen)
    hr = h.clone().requires_grad_(True)
    ref = O.attention_reference(hr, ei)
    (ref["out"] * gout).sum().backward()
    np.savez_compressed(os.path.join(OUT, f"attn_{name}.npz"), h=h.numpy(), edge_index=ei.numpy(),
                        gout=gout.numpy(), out=ref["out"].detach().numpy(), grad_h=hr.grad.numpy(),
                        s=ref["s"].detach().numpy(), alpha=ref["alpha"].detach().numpy(),
                        ei_prime=ref["ei"].numpy())
    print(name, "E' =", ref["ei"].size(1))


# Harness fixture for the attention model (as make_golden.py's make_trajectory): the
# reference trainer's loop (train.py:73-160) on the oracle's AGNN, CPU, seeded.  One layer:
# the wrapper's dropout (p = 0.5) never runs, so the trajectory is RNG-free.
def make_trajectory(epochs=5):
    import torch.nn.functional as F
    from sngnn_amd import synth
    data = synth.make_dataset("cora", seed=7, scale=0.25)
    torch.manual_seed(1234)
    model = O.AGNN(data.x.size(1), 16, 7, 1)
    init = {k: v.clone().numpy() for k, v in model.state_dict().items()}
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    traj = []
    for _ in range(epochs):
        model.train()
        opt.zero_grad()
        out = model(data)
        loss = F.nll_loss(out[data.train_mask], data.y[data.train_mask])
        loss.backward()
        opt.step()
        model.eval()
        with torch.no_grad():
            out = model(data)
            rec = [float(loss)]
            for m in (data.val_mask, data.test_mask):
                rec += [float(F.nll_loss(out[m], data.y[m])),
                        float((out[m].max(1)[1] == data.y[m]).float().mean())]
        traj.append(rec)
    final = {"final." + k: v.clone().numpy() for k, v in model.state_dict().items()}
    np.savez_compressed(os.path.join(OUT, "traj_agnn_1layer.npz"), traj=np.array(traj, np.float64),
                        **{"init." + k: v for k, v in init.items()}, **final)
    print("agnn trajectory", [round(t[0], 4) for t in traj])


make_trajectory()
