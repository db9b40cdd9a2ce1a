"""Per-rank driver -- counterpart of the reference's code/run.py:18-82 on the MI355X-native path:
init RCCL process group (backend "nccl" is RCCL on ROCm) -> Config -> seed -> logger -> data -> model ->
DataParallel (flat all-reduce + sparse row exchange instead of DDP's dense 836 MB all-reduce) -> Trainer.fit ->
evaluate(test)."""
import argparse
import os
from logging import getLogger

import torch
import torch.distributed as dist

from pixelrec_amd.config import Config
from pixelrec_amd.data import bulid_dataloader, load_data
from pixelrec_amd.parallel import DataParallel
from pixelrec_amd.trainer import Trainer
from pixelrec_amd.utils import get_model, init_logger, init_seed


def run_loop(local_rank, config_file=None, saved=True, config_dict=None):
    config = Config(config_file_list=config_file, config_dict=config_dict)
    device = torch.device("cuda", local_rank)
    config["device"] = device
    init_seed(config["seed"], config["reproducibility"])
    init_logger(config)
    logger = getLogger()

    dataload = load_data(config)
    train_loader, valid_loader, test_loader = bulid_dataloader(config, dataload)
    model = get_model(config["model"])(config, dataload).to(device)
    model = DataParallel(model)

    world_size = dist.get_world_size() if dist.is_initialized() else 1
    logger.info(f"\nWorld_Size = {world_size} \n")
    logger.info(config)
    logger.info(dataload)
    logger.info(model.module)

    trainer = Trainer(config, model)
    best_valid_score, best_valid_result = trainer.fit(train_loader, valid_loader, saved=saved,
                                                      show_progress=config["show_progress"])
    test_result = trainer.evaluate(test_loader, load_best_model=saved, show_progress=config["show_progress"])
    logger.info(f"best valid : {best_valid_result}")
    logger.info(f"test result: {test_result}")
    return {"best_valid_score": best_valid_score, "valid_score_bigger": config["valid_metric_bigger"],
            "best_valid_result": best_valid_result, "test_result": test_result}


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--config_file", nargs="+", type=str)
    args = parser.parse_args()
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    run_loop(local_rank=local_rank, config_file=args.config_file)
    if dist.is_initialized():
        dist.destroy_process_group()
