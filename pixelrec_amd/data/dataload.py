"""Data -- interaction CSV -> per-user sequences -> training windows, with the reference's exact rules
(code/REC/data/dataload.py) but vectorised (the reference walks Python dict/list loops, dataload.py:94-150).

Wire format: `<data_path>/<dataset>.csv`, header row, columns item_id,user_id,timestamp (dataload.py:30-37).
Ids are factorised in first-appearance (file) order, +1, 0 = '[PAD]' (:46-54).  build(): sort by timestamp
(pandas sort_values, :68), group per user keeping time order (:71-77), train = all but the last two
interactions of each user (:80-83), then windows of MAX_ITEM_LIST_LENGTH+1: a longer history drops its OLDEST
`len % (L+1)` items and is cut into consecutive full windows, a shorter one is a single window (:103-150).
"""
from __future__ import annotations

import os
from logging import getLogger

import numpy as np
import pandas as pd

from ..utils.enum_type import InputType


class Data:
    def __init__(self, config):
        self.config = config
        self.dataset_path = config["data_path"]
        self.dataset_name = config["dataset"]
        self.logger = getLogger()
        self._load_inter_feat(self.dataset_name, self.dataset_path)
        self._data_processing()

    def _load_inter_feat(self, token, dataset_path):
        path = os.path.join(dataset_path, f"{token}.csv")
        if not os.path.isfile(path):
            raise ValueError(f"File {path} not exist.")
        self.inter_feat = pd.read_csv(path, delimiter=",", dtype={"item_id": str, "user_id": str, "timestamp": int},
                                      header=0, names=["item_id", "user_id", "timestamp"])

    def _data_processing(self):
        self.id2token, self.token2id = {}, {}
        for feature in ["user_id", "item_id"]:
            new_ids, mp = pd.factorize(self.inter_feat[feature])
            mp = np.array(["[PAD]"] + list(mp))
            self.id2token[feature] = mp
            self.token2id[feature] = {t: i for i, t in enumerate(mp)}
            self.inter_feat[feature] = new_ids + 1
        self.user_num = len(self.id2token["user_id"])
        self.item_num = len(self.id2token["item_id"])
        self.inter_num = len(self.inter_feat)
        self.uid_field, self.iid_field = "user_id", "item_id"
        self.user_seq = None
        self.train_feat = None

    def build(self):
        self.inter_feat.sort_values(by="timestamp", ascending=True, inplace=True)
        users = self.inter_feat["user_id"].values
        items = self.inter_feat["item_id"].values
        # group by user in first-appearance order, keeping time order inside each group
        uniq, first_pos, inv = np.unique(users, return_index=True, return_inverse=True)
        order_of_first = np.argsort(first_pos, kind="stable")          # users ranked by first appearance
        rank_of_user = np.empty_like(order_of_first)
        rank_of_user[order_of_first] = np.arange(len(uniq))
        grp = rank_of_user[inv]                                         # group id per interaction
        perm = np.argsort(grp, kind="stable")                           # stable => time order inside a group
        counts = np.bincount(grp, minlength=len(uniq))
        starts = np.concatenate([[0], np.cumsum(counts)[:-1]])
        sorted_items = items[perm]
        uids = uniq[order_of_first]
        self.user_seq = {int(uids[g]): sorted_items[starts[g]:starts[g] + counts[g]] for g in range(len(uniq))}
        self._uids, self._starts, self._counts, self._sorted_items = uids, starts, counts, sorted_items

        if self.config["MODEL_INPUT_TYPE"] not in (InputType.SEQ, None):
            raise NotImplementedError("only InputType.SEQ (SASRec family) is built on this path")
        self.train_feat = self._build_seq()

    def _build_seq(self):
        W = self.config["MAX_ITEM_LIST_LENGTH"] + 1
        uid_list, seqs = [], []
        for g in range(len(self._uids)):
            n = int(self._counts[g]) - 2          # leave-last-two-out
            if n <= 0:
                continue
            s = int(self._starts[g])
            hist = self._sorted_items[s:s + n]
            if n > W:
                off = n % W
                for c in range((n - off) // W):
                    uid_list.append(self._uids[g])
                    seqs.append(hist[off + c * W: off + (c + 1) * W])
            else:
                uid_list.append(self._uids[g])
                seqs.append(hist)
        return {"user_id": np.array(uid_list), "item_seq": seqs}

    # ---- statistics used in the log line -----------------------------------------------------------------
    @property
    def sparsity(self):
        return 1 - self.inter_num / self.user_num / self.item_num

    def __str__(self):
        return "\n".join([str(self.dataset_name), f"The number of users: {self.user_num}",
                          f"The number of items: {self.item_num}", f"The number of inters: {self.inter_num}",
                          f"The sparsity of the dataset: {self.sparsity * 100}%"])

    __repr__ = __str__
