"""Minimal harness reproducing the reference trainer's loop (train.py:73-160,376):
Adam over all parameters; per epoch one train step (nll_loss on the train mask,
backward, step) followed by eval-mode validation and test passes; early stopping
on the validation loss (strict <) with a patience counter; the test accuracy at
the best validation loss is reported; mean epoch wall time as ``Time(s)``.

The reference's own train.py imports PyG at module level (train.py:15-16), so it
cannot run where PyG is absent; the model classes it constructs (train.py:305-315)
are the drop-in surface and are used here with the same positional arguments.
"""
from __future__ import annotations

import time
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .synth import Data


def train_step(model, data, optimizer):
    """train.py:73-89."""
    model.train()
    optimizer.zero_grad()
    output = model(data)
    train_loss = F.nll_loss(output[data.train_mask], data.y[data.train_mask])
    _, pred = output.max(dim=1)
    correct = int(pred[data.train_mask].eq(data.y[data.train_mask]).sum().item())
    train_acc = correct / int(data.train_mask.sum())
    train_loss.backward()
    optimizer.step()
    return train_loss, train_acc


@torch.no_grad()
def eval_step(model, data, mask):
    """train.py:92-103 (validation) and :106-117 (test)."""
    model.eval()
    output = model(data)
    _, pred = output.max(dim=1)
    correct = int(pred[mask].eq(data.y[mask]).sum().item())
    loss = F.nll_loss(output[mask], data.y[mask])
    return loss, correct / int(mask.sum())


def train(model, data, optimizer, epochs: int, patience: int, log=None) -> Dict:
    """train.py:120-160."""
    dur = []
    final_test_acc = 0.0
    bad_counter = 0
    smallest_val_loss = float("inf")
    history = []
    for epoch in range(epochs):
        t0 = time.time()
        train_loss, train_acc = train_step(model, data, optimizer)
        val_loss, val_acc = eval_step(model, data, data.val_mask)
        test_loss, test_acc = eval_step(model, data, data.test_mask)
        dur.append(time.time() - t0)
        rec = dict(epoch=epoch, train_loss=float(train_loss), train_acc=train_acc,
                   val_loss=float(val_loss), val_acc=val_acc, test_loss=float(test_loss),
                   test_acc=test_acc, time_s=sum(dur) / len(dur))
        history.append(rec)
        if log is not None:
            log(rec)
        if float(val_loss) < smallest_val_loss:
            smallest_val_loss = float(val_loss)
            final_test_acc = test_acc
            bad_counter = 0
        else:
            bad_counter += 1
        if bad_counter == patience:
            break
    return dict(final_test_acc=final_test_acc, history=history,
                mean_epoch_s=sum(dur) / max(len(dur), 1))


def epoch_time_ms(workload: str, x: torch.Tensor, edge_index: torch.Tensor, n: int,
                  classes: int, top_k: int, thr: float, *, seed: int = 1234, lr: float = 0.01,
                  weight_decay: float = 5e-4, epochs: int = 30, warmup: int = 5) -> float:
    """Mean wall time of train + validation + test steps (3 forwards, 1 backward,
    Adam) of a 1-layer SNGNN_Plus, device-synchronised - the quantity train.py:135-143
    logs as ``Time(s)``."""
    from .models import SNGNN_Plus
    gen = torch.Generator().manual_seed(seed)
    y = torch.randint(0, classes, (n,), generator=gen).to(x.device)
    r = torch.rand(n, generator=gen)
    data = Data(x=x, edge_index=edge_index, y=y, train_mask=(r < 0.6).to(x.device),
                val_mask=((r >= 0.6) & (r < 0.8)).to(x.device),
                test_mask=(r >= 0.8).to(x.device))
    torch.manual_seed(seed)
    model = SNGNN_Plus(x.size(1), 32, classes, n, 1, top_k, thr, 1, 0.0).to(x.device)
    opt = torch.optim.Adam(model.parameters(), lr=lr, weight_decay=weight_decay)

    def one_epoch():
        train_step(model, data, opt)
        eval_step(model, data, data.val_mask)
        eval_step(model, data, data.test_mask)

    for _ in range(warmup):
        one_epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        one_epoch()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / epochs * 1e3
