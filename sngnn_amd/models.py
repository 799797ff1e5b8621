"""SNGNN / SNGNN_Plus / SNGNN_Plus_Plus: the reference's model wrappers
(models/models.py:265-303, 161-211, 35-86) with identical positional constructor
signatures, ``state_dict`` keys and ``forward(data)`` contract, so the model
construction at train.py:305-315 works unchanged."""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dist as sn_dist
from .conv import AGNNConv, SNConv, SNConv_plus, SNConv_plus_plus


class _Stack(nn.Module):
    def reset_parameters(self):
        for lin in self.lins:
            lin.reset_parameters()
        if self.bn:
            for bn in self.bns:
                bn.reset_parameters()

    def forward_logits(self, data):
        """Everything before the final ``log_softmax`` (lets a trainer fuse the
        classification head, sngnn_amd/train.py:GraphedEpoch)."""
        x, edge_index = data.x, data.edge_index
        for i, lin in enumerate(self.lins[:-1]):
            x = lin(x, edge_index)
            x = F.relu(x, inplace=True)
            if self.bn:
                part = sn_dist.current_partition()
                # batch statistics over every rank's rows, as the single-process batch has them
                x = self.bns[i](x) if part is None else sn_dist.sync_batch_norm(self.bns[i], x, part)
            x = self.dropout(x)
        return self.lins[-1](x, edge_index)

    def forward(self, data):
        return F.log_softmax(self.forward_logits(data), dim=1)

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
