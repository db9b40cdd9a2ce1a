"""load_data / bulid_dataloader with the reference's names (REC/data/utils.py:15-114) for the SASRec family.

Train side: `SeqTrainBatcher` (vectorised whole batches; order = torch DistributedSampler's).  Eval side: the
reference's NonConsecutiveSequentialDistributedSampler sharding -- rank r takes users r, r+W, r+2W, ... with
no padding (utils.py:126-156) -- over SeqEvalDataset + seq_eval_collate.
"""
from __future__ import annotations

import math
from logging import getLogger

import torch
from torch.utils.data import DataLoader

from ..parallel import world_info
from .dataload import Data
from .dataset import SeqEvalBatcher, SeqEvalDataset, SeqTrainBatcher, seq_eval_collate

SUPPORTED = {"SASRec": "SEQ", "MOSASRec": "SEQ", "FSASRec": "SEQ", "GRU4Rec": "SEQ", "NextItNet": "SEQ",
             "MOGRU4Rec": "SEQ", "MONextItNet": "SEQ"}      # REC/data/utils.py:24-31


def load_data(config):
    return Data(config)


class NonConsecutiveSequentialDistributedSampler(torch.utils.data.sampler.Sampler):
    def __init__(self, dataset, rank=None, num_replicas=None):
        r, w = world_info()
        self.dataset = dataset
        self.num_replicas = w if num_replicas is None else num_replicas
        self.rank = r if rank is None else rank
        self.total_size = len(dataset)
        self.num_samples = math.ceil((self.total_size - self.rank) / self.num_replicas)

    def __iter__(self):
        return iter(list(range(self.total_size))[self.rank:self.total_size:self.num_replicas])

    def __len__(self):
        return self.num_samples


class _TrainLoader:
    """DataLoader-shaped facade over SeqTrainBatcher: `.sampler.set_epoch`, `len()`, iteration."""

    def __init__(self, batcher):
        self.batcher = batcher
        self.sampler = batcher
        self.dataset = batcher
        self.item_num = batcher.item_num

    def __len__(self):
        return len(self.batcher)

    def __iter__(self):
        return iter(self.batcher)


def bulid_dataloader(config, dataload):
    """-> (train_loader, valid_loader, test_loader).  (The misspelt name is the reference's, utils.py:20.)"""
    model_name = config["model"]
    if model_name not in SUPPORTED:
        raise NotImplementedError(f"data pipeline for model {model_name!r} is outside this build's scope")
    dataload.build()
    rank, world = world_info()
    logger = getLogger()
    logger.info(f"[Training]: train_batch_size = [{config['train_batch_size']}]")
    logger.info(f"[Evaluation]: eval_batch_size = [{config['eval_batch_size']}]")
    train_loader = _TrainLoader(SeqTrainBatcher(config, dataload, rank=rank, world=world))
    loaders = []
    workers = int(config["eval_num_workers"] or 0)
    for phase in ("valid", "test"):
        if config["eval_vectorized"] is None or bool(config["eval_vectorized"]):
            loaders.append(SeqEvalBatcher(config, dataload, phase=phase, rank=rank, world=world))
            continue
        # literal form of the reference's loaders (data/utils.py:95-110); `eval_vectorized: False` selects it
        ds = SeqEvalDataset(config, dataload, phase=phase)
        sampler = NonConsecutiveSequentialDistributedSampler(ds, rank=rank, num_replicas=world)
        loaders.append(DataLoader(ds, batch_size=config["eval_batch_size"], num_workers=workers, pin_memory=False,
                                  sampler=sampler, collate_fn=seq_eval_collate))
    return train_loader, loaders[0], loaders[1]
