"""Config -- YAML-compatible with the reference's `REC.config.Config` (code/REC/config/configurator.py:13-180).

Same surface: `Config(config_file_list)`; YAML files merged in order, later files win (:73-79); the custom float
resolver so that `1e-12` parses as a float (:32-46); dict-style access returning None for missing keys (:148-152);
derived keys MODEL_INPUT_TYPE, eval_type, valid_metric_bigger, and topk validation (:98-137).
"""
from __future__ import annotations

import re

import yaml

from ..evaluator.register import metric_types, smaller_metrics
from ..utils.utils import get_model

general_arguments = ["seed", "reproducibility", "state", "data_path", "checkpoint_dir", "show_progress", "config_file",
                     "log_wandb", "use_modality"]
training_arguments = ["epochs", "train_batch_size", "optim_args", "eval_step", "stopping_step", "clip_grad_norm",
                      "loss_decimal_place"]
evaluation_arguments = ["eval_type", "repeatable", "metrics", "topk", "valid_metric", "valid_metric_bigger",
                        "eval_batch_size", "metric_decimal_place"]
dataset_arguments = ["MAX_TEXT_LENGTH", "MAX_ITEM_LIST_LENGTH", "MAX_ITEM_LIST_LENGTH_TEST", "text_path", "text_keys"]


def _yaml_loader():
    # a private subclass so the implicit resolver does not leak into yaml.FullLoader globally
    class Loader(yaml.FullLoader):
        pass

    Loader.add_implicit_resolver(
        "tag:yaml.org,2002:float",
        re.compile(
            """^(?:
             [-+]?(?:[0-9][0-9_]*)\\.[0-9_]*(?:[eE][-+]?[0-9]+)?
            |[-+]?(?:[0-9][0-9_]*)(?:[eE][-+]?[0-9]+)
            |\\.[0-9_]+(?:[eE][-+][0-9]+)?
            |[-+]?[0-9][0-9_]*(?::[0-5]?[0-9])+\\.[0-9_]*
            |[-+]?\\.(?:inf|Inf|INF)
            |\\.(?:nan|NaN|NAN))$""", re.X),
        list("-+0123456789."))
    return Loader


class Config:
    def __init__(self, config_file_list=None, config_dict=None):
        self.parameters = {"General": general_arguments, "Training": training_arguments,
                           "Evaluation": evaluation_arguments, "Dataset": dataset_arguments}
        self.yaml_loader = _yaml_loader()
        self.final_config_dict = self._load_config_files(config_file_list)
        if config_dict:
            self.final_config_dict.update(config_dict)
        self.model_class = get_model(self.final_config_dict["model"])
        self._set_default_parameters()

    def _load_config_files(self, file_list):
        cfg = {}
        for file in file_list or []:
            with open(file, "r", encoding="utf-8") as f:
                loaded = yaml.load(f.read(), Loader=self.yaml_loader)
                if loaded:
                    cfg.update(loaded)
        return cfg

    def _set_default_parameters(self):
        d = self.final_config_dict
        if hasattr(self.model_class, "input_type"):
            d["MODEL_INPUT_TYPE"] = self.model_class.input_type
        metrics = d.get("metrics", ["Recall", "NDCG"])
        if isinstance(metrics, str):
            metrics = [metrics]
        d["metrics"] = metrics
        eval_type = set()
        for metric in metrics:
            if metric.lower() in metric_types:
                eval_type.add(metric_types[metric.lower()])
            else:
                raise NotImplementedError(f"There is no metric named '{metric}'")
        if len(eval_type) > 1:
            raise RuntimeError("Ranking metrics and value metrics can not be used at the same time.")
        d["eval_type"] = eval_type.pop()
        valid_metric = d.get("valid_metric", "NDCG@10").split("@")[0]
        d.setdefault("valid_metric", "NDCG@10")
        d["valid_metric_bigger"] = valid_metric.lower() not in smaller_metrics
        topk = d.get("topk", [10])
        if isinstance(topk, int):
            topk = [topk]
        if not isinstance(topk, list):
            raise TypeError(f"The topk [{topk}] must be a integer, list")
        for k in topk:
            if k <= 0:
                raise ValueError(f"topk must be a positive integer or a list of positive integers, but get `{k}`")
        d["topk"] = topk

    def __setitem__(self, key, value):
        if not isinstance(key, str):
            raise TypeError("index must be a str.")
        self.final_config_dict[key] = value

    def __getattr__(self, item):
        if "final_config_dict" not in self.__dict__:
            raise AttributeError("'Config' object has no attribute 'final_config_dict'")
        if item in self.final_config_dict:
            return self.final_config_dict[item]
        raise AttributeError(f"'Config' object has no attribute '{item}'")

    def __getitem__(self, item):
        return self.final_config_dict.get(item, None)

    def __contains__(self, key):
        if not isinstance(key, str):
            raise TypeError("index must be a str.")
        return key in self.final_config_dict

    def __str__(self):
        lines = []
        for category, names in self.parameters.items():
            lines.append(f"{category} Hyper Parameters:")
            lines += [f"{k} = {v}" for k, v in self.final_config_dict.items() if k in names]
            lines.append("")
        known = {n for names in self.parameters.values() for n in names} | {"model", "dataset", "config_files"}
        lines.append("Other Hyper Parameters:")
        lines += [f"{k} = {v}" for k, v in self.final_config_dict.items() if k not in known]
        return "\n" + "\n".join(lines) + "\n"

    __repr__ = __str__
