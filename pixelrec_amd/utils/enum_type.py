"""Input-type tags read by the config layer (mirror of the reference's REC/utils/enum_type.py InputType)."""
from enum import Enum


class InputType(Enum):
    SEQ = 1
    PAIR = 2
    AUGSEQ = 3
