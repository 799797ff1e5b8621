"""SNGNN / SNGNN_Plus / SNGNN_Plus_Plus: the reference's model wrappers
(models/models.py:265-303, 161-211, 35-86) with identical positional constructor
signatures, ``state_dict`` keys and ``forward(data)`` contract, so the model
construction at train.py:305-315 works unchanged."""
from __future__ import annotations

import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dist as sn_dist
from . import ops
from .conv import AGNNConv, SNConv, SNConv_plus, SNConv_plus_plus


# A/B switch of the fused hidden-layer epilogue (_Stack.forward_logits; tests flip it, SNGNN_FUSE_HIDDEN=0
# starts a process with it off)
FUSE_HIDDEN = os.environ.get("SNGNN_FUSE_HIDDEN", "1") != "0"


class _Stack(nn.Module):
    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        if self.bn:
            for bn in self.bns:
                bn.reset_parameters()

    def forward_head(self, data, head):
        """:meth:`forward_logits` with the classification head (``ops.HeadEpilogue``: log_softmax, the NLL
        of a split, its accuracy count and, training, d loss / d logits) run inside the LAST layer's
        own launches where that layer can (SNConv / SNConv_plus: in the aggregation's finalize launch;
        SNConv_plus_plus: in the blend's pass - on one GPU, at most 64 classes in rows of 16-byte vectors):
        ``head.applied`` says whether it did - the result is then what the head
        was asked to leave (the gradient, or the logits) and ``head.metrics`` are written; otherwise
        the result is the logits and the caller runs the head itself."""
        return self.forward_logits(data, head)

    def forward_logits(self, data, head=None):
        """Everything before the final ``log_softmax`` (lets a trainer fuse the
        classification head, sngnn_amd/train.py:GraphedEpoch)."""
        x, edge_index = data.x, data.edge_index
        if head is not None:
            head.applied = False
        takes_head = head is not None and isinstance(self.lins[-1], (SNConv, SNConv_plus, SNConv_plus_plus))
        # relu + dropout between two conv layers as the store epilogue of the aggregation that
        # produces their operand, and backward as the store epilogue of the next ``lin``'s input
        # gradient (ops.HiddenEpilogue) - without batch norm in between, on one GPU, for the layers
        # that take it (SNConv, SNConv_plus); everything else runs the reference's op sequence
        fusable = (FUSE_HIDDEN and not self.bn and sn_dist.current_partition() is None and x.is_cuda
                   and isinstance(self.lins[0], (SNConv, SNConv_plus, SNConv_plus_plus)))
        # EVALUATION with batch norm (models.py:207-208): the running statistics are constants, so the norm is a
        # per-channel scale and shift - folded into the NEXT conv's ``lin`` (conv.LinFold) - and the conv's bias
        # + relu in front of it go into the aggregation's store epilogue: no elementwise pass is left between two
        # conv layers.  (Training-mode batch norm needs the batch's statistics: the op sequence below.)
        if (FUSE_HIDDEN and self.bn and not self.training and sn_dist.current_partition() is None and x.is_cuda
                and isinstance(self.lins[0], (SNConv, SNConv_plus))):
            return self._forward_logits_bn_eval(x, edge_index, head)
        act = None
        seeds = self._dropout_seeds(x.device) if (fusable and len(self.lins) > 1 and self.training
                                                   and self.dropout.p > 0.0) else None
        for i, lin in enumerate(self.lins[:-1]):
            if fusable:
                epi = ops.HiddenEpilogue(True, self.dropout.p, self.training, None if seeds is None else seeds[i:i + 1])
                x = lin(x, edge_index, epi, act)
                if epi.applied:
                    act = epi
                    continue
                act = None
            else:
                x = lin(x, edge_index)
            x = F.relu(x, inplace=True)
            if self.bn:
                part = sn_dist.current_partition()
                # batch statistics over every rank's rows, as the single-process batch has them
                x = self.bns[i](x) if part is None else sn_dist.sync_batch_norm(self.bns[i], x, part)
            x = self.dropout(x)
        if takes_head:
            return self.lins[-1](x, edge_index, None, act if fusable else None, head)
        if fusable and act is not None:
            return self.lins[-1](x, edge_index, None, act)
        return self.lins[-1](x, edge_index)

    def _forward_logits_bn_eval(self, x, edge_index, head):
        from .conv import LinFold
        fold = None
        for i, lin in enumerate(self.lins[:-1]):
            epi = ops.HiddenEpilogue(True, 0.0, False)                   # conv bias + relu in the stores
            y = lin(x, edge_index, epi, None, None, fold)
            if epi.applied:
                x, fold = y, LinFold.of_batch_norm(self.bns[i])          # the norm rides in the next lin
            else:                                                        # (a shape the epilogue does not take)
                x, fold = self.bns[i](F.relu(y, inplace=True)), None
        takes_head = head is not None and isinstance(self.lins[-1], (SNConv, SNConv_plus))
        return self.lins[-1](x, edge_index, None, None, head if takes_head else None, fold)

    def forward(self, data):
        return F.log_softmax(self.forward_logits(data), dim=1)

    def _dropout_seeds(self, device):
        """One counter per hidden layer for the in-kernel dropout draw (ops.HiddenEpilogue): started
        from torch's CPU generator (so ``torch.manual_seed`` makes a run reproducible) and advanced on
        the device once per training forward - inside a captured epoch too, where every replay must
        drop differently.  The counters are CREATED outside any capture (a host-to-device copy of
        pageable memory inside one would either fail or be replayed: the same mask every epoch):
        :meth:`prepare_capture` does it for a trainer, the first eager training forward otherwise."""
        s = getattr(self, "_drop_seed", None)
        if s is None or s.device != device or s.numel() != len(self.lins) - 1:
            if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("the dropout seeds must exist before a graph capture: call "
                                   "model.prepare_capture(device) (GraphedEpoch does) or run one eager "
                                   "training forward first")
            s = torch.randint(0, 2 ** 62, (len(self.lins) - 1,), dtype=torch.int64).to(device)
            self._drop_seed = s
        else:
            s.add_(1)
        return s

    def prepare_capture(self, device) -> None:
        """Everything a training forward creates lazily on the host, created now - call before
        capturing the model in a HIP graph (sngnn_amd.train.GraphedEpoch does)."""
        device = torch.device(device)
        if len(self.lins) > 1 and getattr(self.dropout, "p", 0.0) > 0.0:
            s = getattr(self, "_drop_seed", None)
            if s is None or s.device != device or s.numel() != len(self.lins) - 1:
                self._dropout_seeds(device)

    def _build(self, conv, in_channels, hidden_channels, out_channels, num_layers):
        self.lins = nn.ModuleList()
        if self.bn:
            self.bns = nn.ModuleList()
        if num_layers == 1:
            self.lins.append(conv(in_channels, out_channels))
        else:
            self.lins.append(conv(in_channels, hidden_channels))
            if self.bn:
                self.bns.append(nn.BatchNorm1d(hidden_channels))
            for _ in range(num_layers - 2):
                self.lins.append(conv(hidden_channels, hidden_channels))
                if self.bn:
                    self.bns.append(nn.BatchNorm1d(hidden_channels))
            self.lins.append(conv(hidden_channels, out_channels))


class AGNN(_Stack):
    """models.py:336-374: the cosine-attention baseline on the same kernel skeleton."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, bn=False):
        super().__init__()
        self.bn = bn
        self._build(lambda i, o: AGNNConv(i, o), in_channels, hidden_channels, out_channels,
                    num_layers)
        self.dropout = nn.Dropout(p=0.5)
        self.reset_parameters()


class SNGNN(_Stack):
    """models.py:265-303 (dropout is fixed at 0.5, :283)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_layers, bn=False):
        super().__init__()
        self.bn = bn
        self._build(lambda i, o: SNConv(i, o), in_channels, hidden_channels, out_channels,
                    num_layers)
        self.dropout = torch.nn.Dropout(p=0.5)
        self.reset_parameters()


class SNGNN_Plus(_Stack):
    """models.py:161-211.  ``bn`` is passed positionally into the conv's ``bias``
    slot exactly as the reference does (:177-178)."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_nodes, num_layers,
                 top_k=2, thr=0.0, is_remove_self_loops=1, droput_rate=0.5, bn=False):
        super().__init__()
        self.top_k = top_k
        self.thr = thr
        self.bn = bn
        self.num_nodes = num_nodes
        self.is_remove_self_loops = (is_remove_self_loops == 1)
        self._build(lambda i, o: SNConv_plus(i, o, self.num_nodes, self.top_k, self.thr,
                                             self.is_remove_self_loops, self.bn),
                    in_channels, hidden_channels, out_channels, num_layers)
        self.dropout = torch.nn.Dropout(p=droput_rate)
        self.reset_parameters()


class SNGNN_Plus_Plus(_Stack):
    """models.py:35-86."""

    def __init__(self, in_channels, hidden_channels, out_channels, num_nodes, num_layers,
                 top_k=2, thr=0.0, init_beta=0.5, is_remove_self_loops=1, droput_rate=0.5,
                 bn=False):
        super().__init__()
        self.top_k = top_k
        self.thr = thr
        self.bn = bn
        self.init_beta = init_beta
        self.num_nodes = num_nodes
        self.is_remove_self_loops = (is_remove_self_loops == 1)
        self._build(lambda i, o: SNConv_plus_plus(i, o, self.num_nodes, self.top_k, self.thr,
                                                  self.init_beta, self.is_remove_self_loops,
                                                  self.bn),
                    in_channels, hidden_channels, out_channels, num_layers)
        self.dropout = torch.nn.Dropout(p=droput_rate)
        self.reset_parameters()
