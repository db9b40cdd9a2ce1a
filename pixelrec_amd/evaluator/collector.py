"""Collector -- gathers what the metrics need from each evaluated batch (reference REC/evaluator/collector.py).

`eval_batch_collect(scores, positive_u, positive_i)` keeps the reference signature (top-k over a full [B, N] score
matrix, collector.py:131-139).  `eval_topk_collect(topk_idx, positive_i)` is the entry used by the fused
scoring kernel, which never materialises the [B, N] scores nor the [B, N] int positive matrix: with exactly one
positive per user (SeqEvalDataset, evalset.py:25-34) rec.topk = [hit flags of the top-K | 1]."""
from __future__ import annotations

import copy

import torch


class DataStruct:
    def __init__(self):
        self._data_dict = {}

    def __getitem__(self, name):
        return self._data_dict[name]

    def __setitem__(self, name, value):
        self._data_dict[name] = value

    def __delitem__(self, name):
        self._data_dict.pop(name)

    def __contains__(self, key):
        return key in self._data_dict

    def get(self, name):
        if name not in self._data_dict:
            raise IndexError("Can not load the data without registration !")
        return self[name]

    def set(self, name, value):
        self._data_dict[name] = value

    def update_tensor(self, name, value):
        value = value.detach().cpu()
        if name not in self._data_dict:
            self._data_dict[name] = value.clone()
        else:
            self._data_dict[name] = torch.cat((self._data_dict[name], value), dim=0)


class Collector:
    def __init__(self, config):
        self.config = config
        self.data_struct = DataStruct()
        self.topk = config["topk"]
        self._pending = []       # rec.topk blocks of eval_topk_collect, still on the device (one D2H copy per evaluation)

    def eval_batch_collect(self, scores_tensor, positive_u, positive_i, interaction=None):
        _, topk_idx = torch.topk(scores_tensor, max(self.topk), dim=-1)
        pos_matrix = torch.zeros_like(scores_tensor, dtype=torch.int)
        pos_matrix[positive_u, positive_i] = 1
        pos_len_list = pos_matrix.sum(dim=1, keepdim=True)
        pos_idx = torch.gather(pos_matrix, dim=1, index=topk_idx)
        self.data_struct.update_tensor("rec.topk", torch.cat((pos_idx, pos_len_list), dim=1))

    def eval_topk_collect(self, topk_idx, positive_i):
        """topk_idx int64 [B, K] (already masked top-K item ids), positive_i int64 [B]."""
        pos_idx = (topk_idx == positive_i.to(topk_idx.device).view(-1, 1)).to(torch.int)
        pos_len = torch.ones(pos_idx.shape[0], 1, dtype=torch.int, device=pos_idx.device)
        # kept on the device: a .cpu() per batch (DataStruct.update_tensor) is a host synchronisation per batch -- with the scores
        # never leaving the GPU it was most of a 200 000-user evaluation's wall time (tools/diag/trainer_throughput.py)
        self._pending.append(torch.cat((pos_idx, pos_len), dim=1))

    def get_data_struct(self):
        if self._pending:
            self.data_struct.update_tensor("rec.topk", torch.cat(self._pending, dim=0))
            self._pending = []
        returned = copy.deepcopy(self.data_struct)
        for key in ["rec.topk"]:
            if key in self.data_struct:
                del self.data_struct[key]
        return returned
