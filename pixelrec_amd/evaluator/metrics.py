"""Recall@k and NDCG@k with the reference's exact conventions (REC/evaluator/metrics.py:115-178,
base_metric.py:30-67): inputs are the [n_users, K] hit flags of the top-K list and the number of positives per
user; `topk_result` returns per-rank SUMS over users (the trainer divides after the all_gather)."""
from __future__ import annotations

import numpy as np
import torch


class TopkMetric:
    metric_need = ["rec.topk"]

    def __init__(self, config):
        self.topk = config["topk"]
        self.decimal_place = config["metric_decimal_place"] or 4

    def used_info(self, dataobject):
        rec_mat = dataobject.get("rec.topk")
        topk_idx, pos_len_list = torch.split(rec_mat, [max(self.topk), 1], dim=1)
        return topk_idx.to(torch.bool).numpy(), pos_len_list.squeeze(-1).numpy()

    def topk_result(self, metric, value):
        avg_result = value.sum(axis=0)
        return {"{}@{}".format(metric, k): avg_result[k - 1] for k in self.topk}


class Recall(TopkMetric):
    def calculate_metric(self, dataobject):
        pos_index, pos_len = self.used_info(dataobject)
        return self.topk_result("recall", self.metric_info(pos_index, pos_len))

    def metric_info(self, pos_index, pos_len):
        return np.cumsum(pos_index, axis=1) / pos_len.reshape(-1, 1)


class NDCG(TopkMetric):
    def calculate_metric(self, dataobject):
        pos_index, pos_len = self.used_info(dataobject)
        return self.topk_result("ndcg", self.metric_info(pos_index, pos_len))

    def metric_info(self, pos_index, pos_len):
        K = pos_index.shape[1]
        len_rank = np.full_like(pos_len, K)
        idcg_len = np.where(pos_len > len_rank, len_rank, pos_len)
        ranks = np.zeros_like(pos_index, dtype=np.float64)
        ranks[:, :] = np.arange(1, K + 1)
        idcg = np.cumsum(1.0 / np.log2(ranks + 1), axis=1)
        for row, idx in enumerate(idcg_len):
            idcg[row, idx:] = idcg[row, idx - 1]
        dcg = 1.0 / np.log2(ranks + 1)
        dcg = np.cumsum(np.where(pos_index, dcg, 0), axis=1)
        return dcg / idcg


metrics_dict = {"recall": Recall, "ndcg": NDCG}
