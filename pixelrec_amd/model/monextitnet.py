"""MONextItNet (PixelNet) -- drop-in for `REC.model.PixelNet.monextitnet.MONextItNet`
(code/REC/model/PixelNet/monextitnet.py:12-128): NextItNet's residual blocks over item vectors produced END-TO-END by the
visual encoder.  Composition of MOSASRec's shell (`model/mosasrec.py`) and the block of `model/nextitnet.py`
(`NextItBlock`); parameter names as the reference registers them: `visual_encoder.*`, `residual_blocks.{i}.*`,
`final_layer.*`.
"""
from __future__ import annotations

from ..utils.enum_type import InputType
from .mosasrec import MOSASRec
from .nextitnet import NextItBlock
from .seqcore import SeqRecCore
from .visual import load_model


class MONextItNet(NextItBlock, MOSASRec):
    input_type = InputType.SEQ

    def __init__(self, config, dataload):
        SeqRecCore.__init__(self)
        self.visual_encoder = load_model(config=config)                 # monextitnet.py:25
        self._build_blocks(config, dataload)
        self.residual_blocks.apply(self._init_weights)                  # monextitnet.py:40-41
        self.final_layer.apply(self._init_weights)
        self._init_runtime_state(config)
        self._idx_cache = {}
