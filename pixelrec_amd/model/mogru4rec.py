"""MOGRU4Rec (PixelNet) -- drop-in for `REC.model.PixelNet.mogru4rec.MOGRU4Rec` (code/REC/model/PixelNet/mogru4rec.py:9-82):
the GRU4Rec recurrent block over item vectors produced END-TO-END by the visual encoder.  Composition of two things this
build already has: MOSASRec's shell (`model/mosasrec.py`: encoder output rows as the block's "table", the interleaved
pos | neg row ids, `pxr_mosasrec_emb_grad_f32` for the gradient w.r.t. the encoder output, `predict` / `compute_item`) and
the recurrent block of `model/gru4rec.py` (`GRUBlock`).  Parameter names as the reference registers them: `visual_encoder.*`
(the modal parameter group of trainer.py:74-98), `gru_layers.weight_{ih,hh}_l{k}`, `dense.{weight,bias}`.
"""
from __future__ import annotations

import torch.nn as nn

from ..utils.enum_type import InputType
from .gru4rec import GRUBlock
from .mosasrec import MOSASRec
from .seqcore import SeqRecCore
from .visual import load_model


class MOGRU4Rec(GRUBlock, MOSASRec):
    input_type = InputType.SEQ

    def __init__(self, config, dataload):
        SeqRecCore.__init__(self)
        self.initializer_range = config["initializer_range"]
        self.visual_encoder = load_model(config=config)                 # mogru4rec.py:27
        self._build_gru(config, dataload)
        nn.init.xavier_uniform_(self.gru_layers.weight_hh_l0)           # mogru4rec.py:39-41
        nn.init.xavier_uniform_(self.gru_layers.weight_ih_l0)
        nn.init.xavier_normal_(self.dense.weight)
        self._init_runtime_state(config)
        self._idx_cache = {}
