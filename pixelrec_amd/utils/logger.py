"""Plain `logging` set-up with the reference's file layout: ./log/<model>/<time>.log + stdout, rank 0 at the
configured level and other ranks at WARNING (code/REC/utils/logger.py:41-101, minus the colour glue)."""
from __future__ import annotations

import logging
import os

import torch

from .utils import ensure_dir, get_local_time


def init_logger(config):
    rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
    log_root = config["log_path"] or "./log/"
    dir_name = os.path.join(log_root, str(config["model"]))
    ensure_dir(dir_name)
    logfilepath = os.path.join(dir_name, "{}.log".format(get_local_time()))
    fmt = logging.Formatter("%(asctime)-15s %(levelname)s  %(message)s", "%a %d %b %Y %H:%M:%S")
    state = (config["state"] or "info").lower()
    level = {"info": logging.INFO, "debug": logging.DEBUG, "error": logging.ERROR, "warning": logging.WARNING,
             "critical": logging.CRITICAL}.get(state, logging.INFO)
    if rank != 0:
        level = logging.WARNING
    handlers = []
    fh = logging.FileHandler(logfilepath)
    fh.setLevel(level)
    fh.setFormatter(fmt)
    sh = logging.StreamHandler()
    sh.setLevel(level)
    sh.setFormatter(fmt)
    handlers = [sh, fh]
    logging.basicConfig(level=level, handlers=handlers, force=True)
    return logfilepath
