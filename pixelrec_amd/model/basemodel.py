"""BaseModel: the small common base of the reference's model classes (REC/model/basemodel.py:10-32):
`load_weights(path)` (non-strict, with the rec_fc -> visual_encoder.item_encoder.fc key remap) and a `__str__`
that reports the trainable parameter count."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn


class BaseModel(nn.Module):
    def __init__(self):
        super().__init__()

    def load_weights(self, path):
        checkpoint = torch.load(path, map_location="cpu")
        pretrained = checkpoint["state_dict"]
        state = {k.replace("item_embedding.rec_fc", "visual_encoder.item_encoder.fc"): v for k, v in pretrained.items()}
        return self.load_state_dict(state, strict=False)

    def __str__(self):
        params = sum(int(np.prod(p.size())) for p in self.parameters() if p.requires_grad)
        return super().__str__() + f"\nTrainable parameters: {params}"
