"""Small helpers with the reference's names and behaviour (code/REC/utils/utils.py)."""
from __future__ import annotations

import datetime
import importlib
import os
import random

import numpy as np
import torch


def get_local_time():
    """Time stamp used in log / checkpoint names; like the reference (utils.py:11-21) ranks meet at a barrier
    first so that they agree on the second."""
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.barrier()
    return datetime.datetime.now().strftime("%b-%d-%Y_%H-%M-%S")


def ensure_dir(dir_path):
    os.makedirs(dir_path, exist_ok=True)


def get_model(model_name):
    """Registry by file name, like the reference (utils.py:35-61): class `Name` lives in module `name`.
    Search order mirrors IDNet -> PixelNet (only the SASRec family is implemented in this build)."""
    file_name = model_name.lower()
    for pkg in ("pixelrec_amd.model",):
        path = f"{pkg}.{file_name}"
        if importlib.util.find_spec(path) is not None:
            module = importlib.import_module(path)
            if hasattr(module, model_name):
                return getattr(module, model_name)
    raise ValueError("`model_name` [{}] is not the name of an existing model.".format(model_name))


def early_stopping(value, best, cur_step, max_step, bigger=True):
    """Validation-based early stopping (utils.py:65-106) -> (best, cur_step, stop_flag, update_flag)."""
    stop_flag = False
    update_flag = False
    better = value >= best if bigger else value <= best
    if better:
        cur_step = 0
        best = value
        update_flag = True
    else:
        cur_step += 1
        if cur_step > max_step:
            stop_flag = True
    return best, cur_step, stop_flag, update_flag


def calculate_valid_score(valid_result, valid_metric=None):
    return valid_result[valid_metric] if valid_metric else valid_result["Recall@10"]


def dict2str(result_dict):
    return "    ".join(str(m) + " : " + str(v) for m, v in result_dict.items())


def init_seed(seed, reproducibility):
    """utils.py:138-156 (the cudnn switches have no ROCm counterpart on this path: every kernel here is
    deterministic by construction)."""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def set_color(log, color=None, highlight=True):
    return log  # plain logging in this build (colour glue is out of scope)
