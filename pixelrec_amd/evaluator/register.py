"""Metric registry (reference REC/evaluator/register.py:5-49 + metrics.py).  Only the metrics the shipped configs
use on this path are implemented (Recall, NDCG: overall/ID.yaml:28-30); the names of the reference's other
ranking metrics are known so that configs naming them fail with a clear message."""
metric_types = {"recall": "ranking", "ndcg": "ranking"}
smaller_metrics = ["rmse", "mae", "logloss", "averagepopularity", "giniindex"]
metric_information = {"recall": ["rec.topk"], "ndcg": ["rec.topk"]}
