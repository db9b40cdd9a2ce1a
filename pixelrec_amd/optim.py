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
    def __init__(self, model, lr=1e-4, weight_decay=0.1, betas=(0.9, 0.999), eps=1e-8, table_update="lazy"):
        """table_update: "lazy" (default) = exact catch-up replay of untouched rows, no O(N*D) sweep per step;
        "dense" = sweep the whole table every step.  Both implement the SAME dense-AdamW semantics."""
        if table_update not in ("lazy", "dense"):
            raise ValueError("table_update must be 'lazy' or 'dense'")
        self.table_update = table_update
        self._last = self._hyper = self._cumlog = None
        self._dirty = False   # lazy mode: some rows lag behind step_count (set by step, cleared by flush)
        self.model = model
        if table_update == "lazy":
            model.register_table_hooks(self)
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
            self._last = torch.full((table.shape[0],), self.step_count, dtype=torch.int32, device=table.device)
            self._grow_hyper(table.device, max(1 << 16, 2 * self.step_count + 2))
        return flat, table

    def _grow_hyper(self, device, cap):
        hyper = torch.zeros(cap, 4, dtype=torch.float32, device=device)
        cumlog = torch.zeros(cap, dtype=torch.float64, device=device)
        if self._hyper is not None:
            n = self._hyper.shape[0]
            hyper[:n].copy_(self._hyper)
            cumlog[:n].copy_(self._cumlog)
        self._hyper, self._cumlog = hyper, cumlog

    # ---- hooks called by the model (lazy mode) ----------------------------------------------------------------
    def catch_up_rows(self, idx, n_dev, cap):
        """Bring the rows a forward pass is about to read up to date (through the last completed step)."""
        if self.table_update != "lazy" or self.step_count == 0 or self._last is None:
            return
        g = self.param_groups[0]
        ops.adamw_rows(self.model.item_embedding.weight.data, self._tm, self._tv, self._last, self._hyper, self._cumlog,
                       self.step_count, 0, g["betas"][0], g["betas"][1], g["eps"], rows=idx, n_rows=n_dev, max_rows=cap)

    def flush(self):
        """Bring EVERY row up to date (before evaluation, checkpointing, or reading the table)."""
        if self.table_update != "lazy" or not self._dirty or self._last is None:
            return
        g = self.param_groups[0]
        ops.adamw_rows(self.model.item_embedding.weight.data, self._tm, self._tv, self._last, self._hyper, self._cumlog,
                       self.step_count, 0, g["betas"][0], g["betas"][1], g["eps"])
        self._dirty = False

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
        sp = self.model.sparse_table_grad
        if self.table_update == "dense":
            ops.adamw_table(table, self._tm, self._tv, self._slot, sp, g["lr"], b1, b2, g["eps"], g["weight_decay"],
                            self.step_count)
        else:
            if self.step_count + 1 >= self._cumlog.numel():
                self._grow_hyper(table.device, 2 * self._cumlog.numel())
            ops.adamw_hyper_append(self._hyper, self._cumlog, self.step_count, g["lr"], b1, b2, g["eps"],
                                   g["weight_decay"])
            if sp is not None:
                ops.adamw_rows(table, self._tm, self._tv, self._last, self._hyper, self._cumlog, self.step_count - 1,
                               self.step_count, b1, b2, g["eps"], rows=sp.idx, n_rows=sp.n, max_rows=sp.cap,
                               grows=sp.rows)
            self._dirty = True
        if ev is not None:
            ev[1].record()
            self.table_events.append(ev)

    def state_dict(self):
        self._ensure_state()
        self.flush()
        return {"step": self.step_count, "param_groups": self.param_groups, "m": self._m, "v": self._v,
                "table_m": self._tm, "table_v": self._tv}

    def load_state_dict(self, sd):
        self._ensure_state()
        self.step_count = int(sd["step"])
        self.param_groups = sd["param_groups"]
        for dst, key in ((self._m, "m"), (self._v, "v"), (self._tm, "table_m"), (self._tv, "table_v")):
            dst.copy_(sd[key])
        # a checkpoint is always flushed: every row is up to date through `step`; the per-step scalars of earlier
        # steps are never needed again, only the cumulative-log origin must be consistent (restart it at 0)
        self._last.fill_(self.step_count)
        if self.step_count + 2 >= self._cumlog.numel():
            self._grow_hyper(self._tm.device, 2 * (self.step_count + 2))
        self._cumlog.zero_()
