"""One rank of a training job (what `main.py` launches through torch.distributed.run; the counterpart of the
reference's code/run.py).  Stages, each a function so that they can be driven separately from tests / notebooks:

    join_world()   RCCL process group when WORLD_SIZE > 1 (backend "nccl" IS RCCL on ROCm), one process per GPU
    build()        YAML -> Config -> seeds/logger -> Data + loaders -> model -> DataParallel wrapper
    run_loop()     build + Trainer.fit + test evaluation  (name kept: it is the reference's programmatic entry)

The wrapper exchanges a 17 MB flat gradient plus the batch's sparse table rows instead of DDP's dense 836 MB
all-reduce (pixelrec_amd/parallel.py).
"""
import argparse
import os
from logging import getLogger

import torch
import torch.distributed as dist


def join_world():
    """-> (local_rank, world_size).  Binds this process to its GPU and, for multi-rank jobs, joins the RCCL group."""
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    return local_rank, world


def build(local_rank, config_file=None, config_dict=None):
    """-> (config, dataload, (train, valid, test) loaders, wrapped model) on cuda:<local_rank>."""
    from pixelrec_amd.config import Config
    from pixelrec_amd.data import bulid_dataloader, load_data
    from pixelrec_amd.parallel import DataParallel
    from pixelrec_amd.utils import get_model, init_logger, init_seed

    config = Config(config_file_list=config_file, config_dict=config_dict)
    config["device"] = torch.device("cuda", local_rank)
    init_seed(config["seed"], config["reproducibility"])
    init_logger(config)
    dataload = load_data(config)
    loaders = bulid_dataloader(config, dataload)
    if str(config["table_sharding"] or "").lower() == "row":
        # row-sharded item table (BASELINE configs[3]): owner(id) = id % world, see pixelrec_amd/model/sharded.py
        from pixelrec_amd.model import ShardedDataParallel, ShardedSASRec

        if config["model"] != "SASRec":
            raise NotImplementedError("table_sharding: row is built for the ID model (SASRec)")
        net = ShardedSASRec(config, dataload).to(config["device"])
        return config, dataload, loaders, ShardedDataParallel(net)
    net = get_model(config["model"])(config, dataload).to(config["device"])
    return config, dataload, loaders, DataParallel(net, exchange_rows=exchange_rows(config, loaders[0]))


def exchange_rows(config, train_loader):
    """`dp_exchange_rows` of the YAML: the row capacity of the data-parallel exchange of the sparse table gradient
    (pixelrec_amd/parallel.GradSync).  Absent / 0: the worst case B*(2L+1) in ONE collective; an integer: that many rows per
    rank (ids + count and rows travel in two collectives); "auto": a bound measured on this run's own batches
    (SeqTrainBatcher.estimate_exchange_rows), the maximum over the ranks.  A batch that exceeds the capacity raises."""
    xr = config["dp_exchange_rows"]
    if not xr:
        return None
    if isinstance(xr, str):
        if xr.lower() != "auto":
            raise ValueError(f"dp_exchange_rows must be an integer or 'auto', got {xr!r}")
        batcher = getattr(train_loader, "batcher", None)
        if batcher is None or not hasattr(batcher, "estimate_exchange_rows"):
            return None
        xr = batcher.estimate_exchange_rows()
        if dist.is_initialized() and dist.get_world_size() > 1:
            t = torch.tensor([xr], dtype=torch.int64, device=config["device"] if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            xr = int(t.item())
    return int(xr)


def _report(log, config, dataload, model):
    world = dist.get_world_size() if dist.is_initialized() else 1
    log.info(f"\nWorld_Size = {world} \n")
    for thing in (config, dataload, model.module):
        log.info(thing)


def run_loop(local_rank, config_file=None, saved=True, config_dict=None):
    from pixelrec_amd.trainer import Trainer

    config, dataload, (train_loader, valid_loader, test_loader), model = build(local_rank, config_file, config_dict)
    log = getLogger()
    _report(log, config, dataload, model)
    trainer = Trainer(config, model)
    progress = config["show_progress"]
    best_score, best_result = trainer.fit(train_loader, valid_loader, saved=saved, show_progress=progress)
    test_result = trainer.evaluate(test_loader, load_best_model=saved, show_progress=progress)
    log.info(f"best valid : {best_result}")
    log.info(f"test result: {test_result}")
    return {"best_valid_score": best_score, "valid_score_bigger": config["valid_metric_bigger"],
            "best_valid_result": best_result, "test_result": test_result}


def main(argv=None):
    cli = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    cli.add_argument("--config_file", nargs="+", type=str, help="model YAML followed by the overall YAML")
    args = cli.parse_args(argv)
    local_rank, _ = join_world()
    try:
        run_loop(local_rank=local_rank, config_file=args.config_file)
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
