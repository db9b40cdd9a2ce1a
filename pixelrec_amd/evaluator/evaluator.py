"""Evaluator -- runs the configured metrics over a collected DataStruct (reference REC/evaluator/evaluator.py)."""
from collections import OrderedDict

from .metrics import metrics_dict


class Evaluator:
    def __init__(self, config):
        self.config = config
        self.metrics = [m.lower() for m in config["metrics"]]
        self.metric_class = {m: metrics_dict[m](config) for m in self.metrics}

    def evaluate(self, dataobject):
        result = OrderedDict()
        for m in self.metrics:
            result.update(self.metric_class[m].calculate_metric(dataobject))
        return result
