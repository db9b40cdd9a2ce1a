from .dataload import Data  # noqa: F401
from .dataset import SEQTrainDataset, SeqEvalBatcher, SeqEvalDataset, SeqTrainBatcher, seq_eval_collate  # noqa: F401
from .utils import NonConsecutiveSequentialDistributedSampler, bulid_dataloader, load_data  # noqa: F401
