from .basemodel import BaseModel  # noqa: F401
from .sasrec import SASRec  # noqa: F401
from .mosasrec import MOSASRec  # noqa: F401
from .fsasrec import FSASRec  # noqa: F401
from .gru4rec import GRU4Rec  # noqa: F401
from .nextitnet import NextItNet  # noqa: F401
from .mogru4rec import MOGRU4Rec  # noqa: F401
from .monextitnet import MONextItNet  # noqa: F401
from .sharded import ShardedDataParallel, ShardedSASRec  # noqa: F401
