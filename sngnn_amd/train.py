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

import os
import time
from typing import Dict, Optional

import torch
import torch.nn.functional as F

from .synth import Data


def _split_metrics(log_probs, labels, mask):
    """(mean NLL, accuracy) of the rows a boolean mask selects - what every step of the reference's
    loop computes on its own split (train.py:81-84, 98-102, 112-116)."""
    rows, want = log_probs[mask], labels[mask]
    hits = int((rows.argmax(dim=1) == want).sum())
    return F.nll_loss(rows, want), hits / max(int(want.numel()), 1)


def train_step(model, data, optimizer):
    """One optimisation step on the training split (the contract of train.py:73-89: returns the
    loss tensor and the training accuracy of the forward the step was taken from)."""
    model.train()
    optimizer.zero_grad()
    loss, acc = _split_metrics(model(data), data.y, data.train_mask)
    loss.backward()
    optimizer.step()
    return loss, acc


@torch.no_grad()
def eval_step(model, data, mask):
    """Evaluation-mode loss and accuracy on ``mask`` (train.py:92-103 validation, :106-117 test)."""
    model.eval()
    return _split_metrics(model(data), data.y, mask)


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


class GraphedEpoch:
    """One full epoch (train step + validation + test passes) captured in a HIP graph.

    The reference loop (train.py:134-143) spends most of an epoch on launch latency
    and on host synchronisations: boolean-mask indexing (a ``nonzero`` each),
    ``.item()`` after every step (train.py:83,100,114).  Here the masks become index
    tensors once, the six metrics stay on the device, Adam runs ``capturable`` and the
    whole epoch - the training forward and backward, the optimizer step, the eval-mode forward
    (one for validation and test: see ``share_eval_forward``) - replays as one graph;
    the host reads one 6-float tensor per epoch (needed for early stopping on the
    validation loss, train.py:150-158).  Same arithmetic as :func:`train_step` /
    :func:`eval_step`.
    """

    def __init__(self, model, data, optimizer, warmup: int = 3, share_eval_forward: bool = True):
        self.model, self.data, self.opt = model, data, optimizer
        # validate_step and test_step (train.py:92-117) run the SAME eval-mode forward - same
        # parameters, same data, no dropout, batch-norm on its running statistics - and differ only
        # in the mask their metrics read; these kernels are deterministic, so one forward serves
        # both, bit for bit.  False replays the reference's two passes.
        self.share_eval_forward = share_eval_forward
        dev = data.x.device
        from . import ops
        self._ops = ops
        self.mask = {k: getattr(data, k + "_mask").to(torch.uint8).contiguous()
                     for k in ("train", "val", "test")}
        self.count = {k: max(int(m.sum()), 1) for k, m in self.mask.items()}
        self._eval_sets = (self.mask["val"] | (self.mask["test"] << 1)).contiguous()   # bit 0: val, bit 1: test
        self.fused = hasattr(model, "forward_logits")
        # the head inside the last layer's launches (models._Stack.forward_head) where the model offers it
        self.fused_head = (self.fused and hasattr(model, "forward_head")
                           and os.environ.get("SNGNN_FUSE_HEAD", "1") == "1")
        self.metrics = torch.zeros(6, dtype=torch.float32, device=dev)
        for g in optimizer.param_groups:
            g["capturable"] = True
            # one multi-tensor kernel per step instead of ~10 small ones per parameter (same
            # update rule); only when the caller did not choose an implementation and every
            # parameter is dense (SNGNN++'s column-major w.weight is: its gradient and the
            # Adam state share its strides, so the flat multi-tensor walk lines up)
            def dense(p):
                return p.is_contiguous() or (p.dim() == 2 and p.t().is_contiguous())
            if (isinstance(optimizer, (torch.optim.Adam, torch.optim.AdamW)) and g.get("fused") is None
                    and g.get("foreach") is None and os.environ.get("SNGNN_FUSED_ADAM", "1") == "1"
                    and all(dense(p) and p.is_cuda for p in g["params"])):
                g["fused"] = True
        self._materialise_adam_state()
        if hasattr(model, "prepare_capture"):
            # host-side lazy state of a training forward (the in-kernel dropout's seed counters) must
            # exist before the capture: with warmup=0 the first training forward IS the captured one
            model.prepare_capture(dev)
        from .graph import GLOBAL_CACHE
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            # library handles and the per-(width, STREAM) workspaces exist before the capture: the
            # priming forward, the warm-up and the capture all run on this one stream
            with torch.no_grad():
                model.eval()
                model(data)
            for _ in range(warmup):          # allocator / lazy-init warm-up outside the capture
                self._epoch()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        # The captured launches hold raw device pointers into the structure arrays and the
        # workspaces of the graphs the model runs on.  The cache that owns them evicts (FIFO):
        # keep exactly those alive for as long as this capture can be replayed.
        with GLOBAL_CACHE.record() as used:
            with torch.cuda.graph(self.graph, stream=side):
                self._epoch()
        self._held_graphs = list(used)

    def _materialise_adam_state(self):
        """Adam creates its state lazily inside the first ``step()``; inside a capture
        those zero-fills would be replayed every epoch.  Create it up front (what
        torch.optim.Adam._init_group does for capturable groups)."""
        if not isinstance(self.opt, (torch.optim.Adam, torch.optim.AdamW)):
            return
        for group in self.opt.param_groups:
            for p in group["params"]:
                st = self.opt.state[p]
                if len(st) == 0:
                    st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    if group.get("amsgrad", False):
                        st["max_exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)

    def _forward(self):
        return self.model.forward_logits(self.data) if self.fused else self.model(self.data)

    def _loss(self, which, out=None):
        """(mean NLL, correct count) on one mask: the fused head kernel on the model's
        logits, or the reference expressions for a model without ``forward_logits``.
        ``out``: the forward's result when the caller already has it."""
        if out is None:
            out = self._forward()
        if self.fused:
            # the kernel writes (loss, correct) straight into this split's slots of the metrics
            slot = {"train": 0, "val": 2, "test": 4}[which]
            return self._ops.head_nll(out, self.data.y,
                                      self.mask[which], self.count[which], out=self.metrics[slot:slot + 2])
        m = self.mask[which].bool()
        return (F.nll_loss(out[m], self.data.y[m]),
                (out[m].max(dim=1)[1] == self.data.y[m]).sum().float())

    def _epoch(self):
        self.model.train()
        # grads set to None: autograd then ASSIGNS the fresh gradient tensors (allocated in
        # the graph's private pool, same addresses on every replay) instead of zero-filling
        # and accumulating - two small kernels per parameter less
        self.opt.zero_grad(set_to_none=True)
        if self.fused_head:
            # the LAST layer's launches run the head (ops.HeadEpilogue): its output IS d loss / d logits
            head = self._ops.HeadEpilogue(self.data.y, self.mask["train"], self.metrics[0:2], self.count["train"],
                                          grad=True)
            g = self.model.forward_head(self.data, head)
            if head.applied:
                g.backward(g.detach())
            else:
                (loss, correct), grad = self._ops.head_nll_with_grad(
                    g, self.data.y, self.mask["train"], self.count["train"], out=self.metrics[0:2])
                g.backward(grad)
        elif self.fused:
            # the head kernel hands back d loss / d logits directly: backward starts at the logits
            logits = self.model.forward_logits(self.data)
            (loss, correct), grad = self._ops.head_nll_with_grad(
                logits, self.data.y, self.mask["train"], self.count["train"], out=self.metrics[0:2])
            logits.backward(grad)
        else:
            loss, correct = self._loss("train")
            loss.backward()
        self.opt.step()
        with torch.no_grad():
            self.model.eval()
            if self.fused_head and self.share_eval_forward:
                head = self._ops.HeadEpilogue(self.data.y, self._eval_sets, self.metrics[2:6], self.count["val"],
                                              self.count["test"])
                shared = self.model.forward_head(self.data, head)
                if head.applied:
                    return
            elif self.fused_head:
                for which, slot in (("val", 2), ("test", 4)):
                    head = self._ops.HeadEpilogue(self.data.y, self.mask[which], self.metrics[slot:slot + 2],
                                                  self.count[which])
                    z = self.model.forward_head(self.data, head)
                    if not head.applied:
                        self._loss(which, z)
                return
            else:
                shared = self._forward() if self.share_eval_forward else None
            if shared is not None and self.fused and shared.size(1) <= 64:
                # both splits' metrics in one pass over the logits, straight into metrics[2:6]
                self._ops.head_nll2(shared, self.data.y, self._eval_sets, self.count["val"],
                                    self.count["test"], out=self.metrics[2:6])
                return
            vl, vc = self._loss("val", shared)
            tl, tc = self._loss("test", shared)
            if not self.fused:
                torch.stack([loss.detach(), correct, vl, vc, tl, tc], out=self.metrics)   # one kernel

    def run(self) -> Dict[str, float]:
        """Replay one epoch; returns the metrics (one host read)."""
        self.graph.replay()
        m = self.metrics.tolist()
        return dict(train_loss=m[0], train_acc=m[1] / self.count["train"], val_loss=m[2],
                    val_acc=m[3] / self.count["val"], test_loss=m[4],
                    test_acc=m[5] / self.count["test"])


def train_graphed(model, data, optimizer, epochs: int, patience: int) -> Dict:
    """:func:`train` with the epoch replayed from a HIP graph (same early stopping).
    The capture warm-up runs real optimizer steps, so pass a freshly initialised
    model when the trajectory must match :func:`train` epoch for epoch."""
    ge = GraphedEpoch(model, data, optimizer, warmup=0)
    dur, history = [], []
    final_test_acc, bad_counter, smallest_val_loss = 0.0, 0, float("inf")
    for epoch in range(epochs):
        t0 = time.time()
        rec = ge.run()
        dur.append(time.time() - t0)
        rec.update(epoch=epoch, time_s=sum(dur) / len(dur))
        history.append(rec)
        if rec["val_loss"] < smallest_val_loss:
            smallest_val_loss, final_test_acc, bad_counter = rec["val_loss"], rec["test_acc"], 0
        else:
            bad_counter += 1
        if bad_counter == patience:
            break
    return dict(final_test_acc=final_test_acc, history=history,
                mean_epoch_s=sum(dur) / max(len(dur), 1))


def epoch_time_ms(workload: str, x: torch.Tensor, edge_index: torch.Tensor, n: int,
                  classes: int, top_k: int, thr: float, *, seed: int = 1234, lr: float = 0.01,
                  weight_decay: float = 5e-4, epochs: int = 30, warmup: int = 5,
                  graphed: bool = False, share_eval_forward: bool = True) -> float:
    """Wall time per epoch (median of five batches' means) of train + validation + test steps of a 1-layer SNGNN_Plus,
    device-synchronised - the quantity train.py:135-143 logs as ``Time(s)``.  Eager: the
    reference's loop (3 forwards, 1 backward, Adam).  ``graphed`` replays the epoch from a HIP
    graph (``share_eval_forward``: one eval forward for the validation and test metrics)."""
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

    if graphed:
        ge = GraphedEpoch(model, data, opt, share_eval_forward=share_eval_forward)
        one_epoch = ge.run
    else:
        def one_epoch():
            train_step(model, data, opt)
            eval_step(model, data, data.val_mask)
            eval_step(model, data, data.test_mask)

    for _ in range(warmup):
        one_epoch()
    torch.cuda.synchronize()
    # the median of five batches' means: one stall of the machine inside a single loop of 30
    # epochs (seen once: 0.37 -> 0.98 ms) would otherwise be the figure
    per = max(epochs // 5, 1)
    batches = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(per):
            one_epoch()
        torch.cuda.synchronize()
        batches.append((time.perf_counter() - t0) / per * 1e3)
    batches.sort()
    return batches[len(batches) // 2]
