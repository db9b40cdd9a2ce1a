from .collector import Collector, DataStruct  # noqa: F401
from .evaluator import Evaluator  # noqa: F401
from .register import metric_types, smaller_metrics  # noqa: F401
