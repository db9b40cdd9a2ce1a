"""PxrAdamW -- torch.optim.AdamW semantics (reference trainer.py:66-103,125) on the fused HIP kernels.

  * every non-table parameter is updated by ONE launch over the model's flat buffer (pxr_adamw_flat_f32);
  * the item-embedding table keeps DENSE AdamW semantics (weight decay and stale moments move every row each
    step, overall/ID.yaml:20-23) but consumes the step's gradient in sparse form (pxr_adamw_table_f32), so the
    819 MB dense gradient of the reference never exists.
Defaults follow torch.optim.AdamW: betas (0.9, 0.999), eps 1e-8.
"""
from __future__ import annotations

import torch

from . import ops


class PxrAdamW:
    def __init__(self, model, lr=1e-4, weight_decay=0.1, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        self.lr, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.step_count = 0
        self._m = self._v = self._tm = self._tv = self._slot = None
        self.table_events = None  # bench.py: list collecting (start, end) HIP events around the table sweep
        self.param_groups = [{"lr": self.lr, "weight_decay": self.weight_decay, "betas": self.betas, "eps": self.eps}]

    def _ensure_state(self):
        flat, _ = self.model.flat_parameters()
        table = self.model.item_embedding.weight.data
        if self._m is None or self._m.device != flat.device or self._m.numel() != flat.numel():
            self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
        if self._tm is None or self._tm.device != table.device or self._tm.shape != table.shape:
            self._tm, self._tv = torch.zeros_like(table), torch.zeros_like(table)
            self._slot = torch.empty(table.shape[0], dtype=torch.int32, device=table.device)
            ops.slot_fill(self._slot, -1)
        return flat, table

    def zero_grad(self, set_to_none: bool = False):
        """No-op: every backward OVERWRITES the flat gradient buffer and the sparse table gradient."""
        return None

    @torch.no_grad()
    def step(self):
        flat, table = self._ensure_state()
        _, gflat = self.model.flat_parameters()
        self.step_count += 1
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        ops.adamw_flat(flat, gflat, self._m, self._v, g["lr"], b1, b2, g["eps"], g["weight_decay"], self.step_count)
        ev = None
        if self.table_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        ops.adamw_table(table, self._tm, self._tv, self._slot, self.model.sparse_table_grad, g["lr"], b1, b2,
                        g["eps"], g["weight_decay"], self.step_count)
        if ev is not None:
            ev[1].record()
            self.table_events.append(ev)

    def state_dict(self):
        self._ensure_state()
        return {"step": self.step_count, "param_groups": self.param_groups, "m": self._m, "v": self._v,
                "table_m": self._tm, "table_v": self._tv}

    def load_state_dict(self, sd):
        self._ensure_state()
        self.step_count = int(sd["step"])
        self.param_groups = sd["param_groups"]
        for dst, key in ((self._m, "m"), (self._v, "v"), (self._tm, "table_m"), (self._tv, "table_v")):
            dst.copy_(sd[key])
