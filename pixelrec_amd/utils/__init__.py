from .enum_type import InputType  # noqa: F401
from .utils import (calculate_valid_score, dict2str, early_stopping, ensure_dir, get_local_time, get_model,  # noqa: F401
                    init_seed, set_color)
from .logger import init_logger  # noqa: F401
