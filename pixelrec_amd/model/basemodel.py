"""Common base of the model classes (the reference keeps one too, REC/model/basemodel.py): checkpoint warm-start via
`load_weights(path)` -- non-strict, with the historical `item_embedding.rec_fc` -> `visual_encoder.item_encoder.fc`
key rename the reference applies -- and a printable summary ending in the trainable-parameter count that the reference
logs at start-up."""
from __future__ import annotations

import torch
from torch import nn

_RENAMES = (("item_embedding.rec_fc", "visual_encoder.item_encoder.fc"),)


def _renamed(key: str) -> str:
    for old, new in _RENAMES:
        key = key.replace(old, new)
    return key


class BaseModel(nn.Module):
    def load_weights(self, path):
        """Warm-start from a `.pth` written by `Trainer._save_checkpoint` (or by the reference): keys that do not
        exist here are ignored, as are missing ones (strict=False); returns torch's missing/unexpected report."""
        stored = torch.load(path, map_location="cpu", weights_only=False)["state_dict"]
        return self.load_state_dict({_renamed(k): v for k, v in stored.items()}, strict=False)

    def trainable_parameter_count(self) -> int:
        return sum(p.numel() for p in self.parameters() if p.requires_grad)

    def __str__(self):
        return f"{super().__str__()}\nTrainable parameters: {self.trainable_parameter_count()}"
