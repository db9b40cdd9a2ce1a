"""PxrAdamW -- torch.optim.AdamW semantics (reference trainer.py:66-103,125) on the fused HIP kernels.

  * every non-table parameter is updated by ONE launch over the model's flat buffer;
  * the item-embedding table keeps DENSE AdamW semantics (weight decay and stale moments move every row each
    step, overall/ID.yaml:20-23) while consuming the step's gradient in sparse form, so the 819 MB dense gradient of
    the reference never exists.  Two equivalent schedules:
      table_update="lazy"  (default): untouched rows are not swept; each row remembers the step it is current
                           through and its missed zero-gradient steps are replayed exactly when the row is next read
                           (before the forward) or updated; `flush()` brings every row up to date before
                           evaluation / checkpointing (called automatically through the model's table hooks);
      table_update="dense": sweep p, m, v of the whole table every step (HBM-bound, 4.9 GB/step at N=400K, D=512).
  * the step number lives ON THE DEVICE (`_step_dev`): bias corrections are computed by a one-thread kernel into a
    per-step scalar table, so a whole training step can be captured in a hipGraph and replayed.
Defaults follow torch.optim.AdamW: betas (0.9, 0.999), eps 1e-8.
"""
from __future__ import annotations

import os

import torch

from . import ops

def _layer_names(i):
    """Parameter names of encoder layer i in the order the reference registers them (layers.py:569-578, 634-638)."""
    a, f = f"trm_encoder.layer.{i}.multi_head_attention.", f"trm_encoder.layer.{i}.feed_forward."
    out = []
    for mod in ("query", "key", "value", "dense", "LayerNorm"):
        out += [a + mod + ".weight", a + mod + ".bias"]
    for mod in ("dense_1", "dense_2", "LayerNorm"):
        out += [f + mod + ".weight", f + mod + ".bias"]
    return out


def reference_rec_parameter_names(model):
    """Names of the 'rec' parameters (everything outside `visual_encoder`) in the order the REFERENCE model yields them
    from .parameters() -- the order torch.optim.AdamW.state_dict() numbers its per-parameter state in:
    SASRec   (sasrec.py:31-45):   item_embedding, position_embedding, trm_encoder layers, LayerNorm;
    MOSASRec (mosasrec.py:30-47): [visual_encoder first], position_embedding, LayerNorm, trm_encoder layers."""
    if hasattr(model, "rec_parameter_names"):          # a backbone that is not the Transformer block (GRU4Rec)
        return list(model.rec_parameter_names())
    layers = [n for i in range(model.n_layers) for n in _layer_names(i)]
    if has_item_table(model):
        return ["item_embedding.weight", "position_embedding.weight"] + layers + ["LayerNorm.weight", "LayerNorm.bias"]
    if hasattr(model, "encoder_parameter_names"):
        # FSASRec (fsasrec.py:34-49): the item encoder's parameters, position_embedding, trm_encoder layers, LayerNorm
        return (list(model.encoder_parameter_names()) + ["position_embedding.weight"] + layers
                + ["LayerNorm.weight", "LayerNorm.bias"])
    return ["position_embedding.weight", "LayerNorm.weight", "LayerNorm.bias"] + layers


def has_item_table(model) -> bool:
    """True for the ID model: `item_embedding` is the nn.Embedding that is updated sparsely / lazily and exchanged as rows.
    (FSASRec also has an `item_embedding`, but it is an encoder module whose parameters live in the flat buffer.)"""
    return isinstance(getattr(model, "item_embedding", None), torch.nn.Embedding)


def is_torch_adamw_state(sd) -> bool:
    return isinstance(sd, dict) and "state" in sd and "param_groups" in sd and "m" not in sd


_TORCH_GROUP_DEFAULTS = {"amsgrad": False, "maximize": False, "foreach": None, "capturable": False,
                         "differentiable": False, "fused": None}


def native_to_torch_state(sd, model, first_index=0):
    """PxrAdamW's native state ({step, m, v, table_m, table_v, param_groups}) -> the layout of
    torch.optim.AdamW.state_dict() over the reference's parameter order (what the reference Trainer stores under
    'optimizer', trainer.py:153, and resumes from, trainer.py:186).  Table moments must be reference-shaped [N, D]."""
    names = reference_rec_parameter_names(model)
    g = dict(sd["param_groups"][0])
    step = int(sd["step"])
    state = {}
    if step > 0:     # torch creates a parameter's state at its first step
        for j, name in enumerate(names):
            if name == "item_embedding.weight":
                m, v = sd["table_m"], sd["table_v"]
            else:
                off, n, shape = model._views[_short_name(name, model)]
                m, v = sd["m"][off:off + n].view(shape), sd["v"][off:off + n].view(shape)
            state[first_index + j] = {"step": torch.tensor(float(step)), "exp_avg": m.detach().clone(),
                                      "exp_avg_sq": v.detach().clone()}
    group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"],
             **_TORCH_GROUP_DEFAULTS, "params": list(range(first_index, first_index + len(names)))}
    return {"state": state, "param_groups": [group]}


def torch_to_native_state(sd, model, group_index=-1):
    """Inverse of native_to_torch_state for the rec group (the LAST param group of a reference checkpoint)."""
    names = reference_rec_parameter_names(model)
    g = sd["param_groups"][group_index]
    ids = list(g["params"])
    if len(ids) != len(names):
        raise ValueError(f"optimizer state has {len(ids)} parameters in its rec group, this model has {len(names)}")
    flat, _ = model.flat_parameters()
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    out = {"param_groups": [{"lr": g["lr"], "weight_decay": g["weight_decay"], "betas": tuple(g["betas"]), "eps": g["eps"]}]}
    steps = set()
    for pid, name in zip(ids, names):
        st = sd["state"].get(pid)
        if st is None:
            continue
        steps.add(int(float(st["step"])))
        if name == "item_embedding.weight":
            out["table_m"], out["table_v"] = st["exp_avg"], st["exp_avg_sq"]
            continue
        off, n, shape = model._views[_short_name(name, model)]
        if tuple(st["exp_avg"].shape) != tuple(shape):
            raise ValueError(f"optimizer state of {name}: shape {tuple(st['exp_avg'].shape)} != {tuple(shape)}")
        m[off:off + n].copy_(st["exp_avg"].reshape(-1))
        v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
    if len(steps) > 1:
        raise ValueError(f"per-parameter step counts differ ({sorted(steps)}): not a state this optimizer can resume")
    out["step"] = steps.pop() if steps else 0
    out["m"], out["v"] = m, v
    if has_item_table(model) and "table_m" not in out:
        t = model.item_embedding.weight
        out["table_m"], out["table_v"] = torch.zeros_like(t), torch.zeros_like(t)
    return out


def _short_name(name, model=None):
    """reference parameter name -> key of SeqRecCore._views (the flat-buffer layout)."""
    extra = model.encoder_parameter_names() if (model is not None and hasattr(model, "encoder_parameter_names")) else {}
    if name in extra:
        return extra[name]
    if model is not None and hasattr(model, "rec_parameter_names"):
        return model.rec_parameter_names()[name]
    if name == "position_embedding.weight":
        return "pos"
    if name.startswith("LayerNorm."):
        return "ln0.w" if name.endswith("weight") else "ln0.b"
    parts = name.split(".")
    i, blk, mod, kind = parts[2], parts[3], parts[4], parts[5][0]
    key = {("multi_head_attention", "query"): "q", ("multi_head_attention", "key"): "k",
           ("multi_head_attention", "value"): "v", ("multi_head_attention", "dense"): "o",
           ("multi_head_attention", "LayerNorm"): "ln1", ("feed_forward", "dense_1"): "f1",
           ("feed_forward", "dense_2"): "f2", ("feed_forward", "LayerNorm"): "ln2"}[(blk, mod)]
    return f"{i}.{key}.{kind}"


HYPER_CAPACITY = 1 << 22   # steps; 96 MB of per-step scalars, sized once so graph replays never see a reallocation


class PxrAdamW:
    def __init__(self, model, lr=1e-4, weight_decay=0.1, betas=(0.9, 0.999), eps=1e-8, table_update="lazy"):
        if table_update not in ("lazy", "dense"):
            raise ValueError("table_update must be 'lazy' or 'dense'")
        self.table_update = table_update
        self.model = model
        self.lr, self.weight_decay, self.betas, self.eps = float(lr), float(weight_decay), tuple(betas), float(eps)
        self.step_count = 0          # host mirror of the device counter
        self._m = self._v = self._tm = self._tv = self._slot = None
        self._last = self._hyper = self._cumlog = self._step_dev = None
        self._cur_hyper, self._cur_for = None, None      # scalars of the step in flight, parked by catch_up_input_ids
        self._dirty = False          # lazy mode: some rows lag behind step_count (set by step, cleared by flush)
        self._seeded_cfg = None      # hyper-parameters the table entry of step_count+1 was written with (None: not yet)
        self.param_groups = [{"lr": self.lr, "weight_decay": self.weight_decay, "betas": self.betas, "eps": self.eps}]
        self.has_table = has_item_table(model)
        if table_update == "lazy" and self.has_table:
            model.register_table_hooks(self)

    def _ensure_state(self):
        flat, _ = self.model.flat_parameters()
        dev = flat.device
        if self._m is None or self._m.device != flat.device or self._m.numel() != flat.numel():
            self._m, self._v = torch.zeros_like(flat), torch.zeros_like(flat)
        if not self.has_table:
            if self._hyper is None or self._hyper.device != dev:
                self._hyper = torch.zeros(HYPER_CAPACITY, 4, dtype=torch.float32, device=dev)
                self._cumlog = torch.zeros(HYPER_CAPACITY, dtype=torch.float64, device=dev)
                self._step_dev = torch.full((1,), self.step_count, dtype=torch.int64, device=dev)
                self._seeded_cfg = None
            return flat, None
        table = self.model.item_embedding.weight.data
        if self._tm is None or self._tm.device != dev or self._tm.shape != table.shape:
            self._tm, self._tv = torch.zeros_like(table), torch.zeros_like(table)
            self._slot = torch.empty(table.shape[0], dtype=torch.int32, device=dev)
            ops.slot_fill(self._slot, -1)
            self._last = torch.full((table.shape[0],), self.step_count, dtype=torch.int32, device=dev)
            self._hyper = torch.zeros(HYPER_CAPACITY, 4, dtype=torch.float32, device=dev)
            self._cumlog = torch.zeros(HYPER_CAPACITY, dtype=torch.float64, device=dev)
            self._step_dev = torch.full((1,), self.step_count, dtype=torch.int64, device=dev)
            self._seeded_cfg = None
        return flat, table

    def zero_grad(self, set_to_none: bool = False):
        """No-op: every backward OVERWRITES the flat gradient buffer and the sparse table gradient."""
        return None

    # ---- hooks called by the model (lazy mode) ----------------------------------------------------------------
    def catch_up_rows(self, idx, n_dev, cap, max_blocks=0):
        """Bring the rows a forward pass is about to read up to date (through the last completed step).  max_blocks: a
        thin grid for the look-ahead call that runs beside the step's GEMMs (SASRec._start_prefetch)."""
        if self.table_update != "lazy" or self._last is None or not self.has_table:
            return
        b1, b2 = self.param_groups[0]["betas"]
        ops.adamw_rows(self.model.item_embedding.weight.data, self._tm, self._tv, self._last, self._hyper, self._cumlog,
                       self.step_count, 0, b1, b2, self.param_groups[0]["eps"], rows=idx, n_rows=n_dev, max_rows=cap,
                       step_dev=self._step_dev, max_blocks=max_blocks)

    def catch_up_ids(self, ids):
        """catch_up_rows on the batch's raw id tensor (duplicates and padding zeros allowed): needs no sorted unique list, so the
        model can run the sort of the ids beside its forward pass.  False when there is nothing lazy to do."""
        if self.table_update != "lazy" or self._last is None or not self.has_table:
            return False
        b1, b2 = self.param_groups[0]["betas"]
        ops.adamw_rows_ids(self.model.item_embedding.weight.data, self._tm, self._tv, self._last, self._hyper, self._cumlog,
                           self.step_count, b1, b2, self.param_groups[0]["eps"], ids, step_dev=self._step_dev)
        return True

    def catch_up_input_ids(self, items):
        """catch_up_ids on the INPUT ids of a batch only (items [B, 2, L+1] -> the window items[:, 0, 0:L]): the rows the forward
        pass gathers first.  False when there is nothing lazy to do."""
        if self.table_update != "lazy" or self._last is None or not self.has_table:
            return False
        b1, b2 = self.param_groups[0]["betas"]
        B, _, W = items.shape
        if self._cur_hyper is None or self._cur_hyper.device != items.device:
            self._cur_hyper = torch.zeros(4, dtype=torch.float32, device=items.device)
        ops.adamw_rows_ids2d(self.model.item_embedding.weight.data, self._tm, self._tv, self._last, self._hyper, self._cumlog,
                             self.step_count, b1, b2, self.param_groups[0]["eps"], items, B, W - 1, 2 * W, step_dev=self._step_dev,
                             cur_hyper_out=self._cur_hyper)
        # the launch left the scalars of step step_count + 1 in _cur_hyper: step() may read them there and close the step in
        # its flat launch (valid only for that step, and only if the entry existed already -- see step())
        self._cur_for = self.step_count + 1 if self._seeded_cfg is not None else None
        return True

    def flush(self):
        """Bring EVERY row up to date (before evaluation, checkpointing, or reading the table as a whole)."""
        if self.table_update != "lazy" or not self._dirty or self._last is None:
            return
        b1, b2 = self.param_groups[0]["betas"]
        ops.adamw_rows(self.model.item_embedding.weight.data, self._tm, self._tv, self._last, self._hyper, self._cumlog,
                       self.step_count, 0, b1, b2, self.param_groups[0]["eps"], step_dev=self._step_dev)
        self._dirty = False

    @torch.no_grad()
    def step(self):
        flat, table = self._ensure_state()
        _, gflat = self.model.flat_parameters()
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        if self.step_count + 2 >= HYPER_CAPACITY:
            raise RuntimeError("PxrAdamW: per-step scalar table exhausted (raise optim.HYPER_CAPACITY)")
        sd = self._step_dev
        cfg = (g["lr"], b1, b2, g["eps"], g["weight_decay"])
        reseeded = self._seeded_cfg != cfg
        if self._seeded_cfg != cfg:
            # first step, after load_state_dict, or the hyper-parameters changed since the entry of this step was
            # prepared (at the end of the previous step): (re)write it now
            ops.adamw_hyper_append(self._hyper, self._cumlog, self.step_count + 1, *cfg, step_dev=sd)
            self._seeded_cfg = cfg
        # table rows first, flat buffer last: the two updates are independent, and under data parallelism the flat
        # gradient's all-reduce may still be in flight (GradSync.sync(defer_flat=True)) -- it then runs under the
        # row update instead of in front of it
        sp = self.model.sparse_table_grad if self.has_table else None
        pre = getattr(self.model, "_prefetched", None) if self.has_table else None
        if pre is not None:
            self.model.join_prefetch()      # the look-ahead catch-up touches table rows: order this step's update behind it
            self.model._prefetched = None
        if not self.has_table:
            pass
        elif self.table_update == "dense":
            # host-computed scalars: this schedule is for eager execution / A-B measurements, not for graph capture
            ops.adamw_table(table, self._tm, self._tv, self._slot, sp, g["lr"], b1, b2, g["eps"], g["weight_decay"],
                            self.step_count + 1)
        elif sp is not None:
            ops.adamw_rows(table, self._tm, self._tv, self._last, self._hyper, self._cumlog, self.step_count,
                           self.step_count + 1, b1, b2, g["eps"], rows=sp.idx, n_rows=sp.n, max_rows=sp.cap,
                           grows=sp.rows, step_dev=sd)
            self._dirty = True
            if pre is not None:
                # rows of the NEXT batch (caught up through the previous step beside this step's GEMMs) advance through
                # THIS step too -- zero gradient unless the update above already did it -- so the next forward finds them
                # current and its own catch-up has nothing to replay
                ops.adamw_rows(table, self._tm, self._tv, self._last, self._hyper, self._cumlog, self.step_count + 1, 0,
                               b1, b2, g["eps"], rows=pre.idx, n_rows=pre.n, max_rows=pre.cap, step_dev=sd,
                               step_dev_bias=1)
        wait = getattr(self.model, "wait_flat_grads", None)
        if wait is not None:
            wait()
        # planes mode: the updated weight matrices leave this launch already split (the next forward skips its split launch)
        segs = self.model.weight_plane_segments() if hasattr(self.model, "weight_plane_segments") else None
        # this step's scalars were parked by the catch-up launch at the head of the step (catch_up_input_ids) and have not been
        # re-seeded since: the flat launch reads them there and closes the step itself (one launch less)
        fold = (self._cur_for == self.step_count + 1 and not reseeded and os.environ.get("PXR_FOLD_CLOSE", "1") != "0")
        self._cur_for = None
        exps = self.model.weight_plane_exps() if (segs and hasattr(self.model, "weight_plane_exps")) else None
        ops.adamw_flat_tab(flat, gflat, self._m, self._v, self._hyper, self.step_count + 1, b1, b2, g["eps"], step_dev=sd,
                           plane_segments=segs, planes_exps=exps,
                           close=(self._cumlog, self._cur_hyper, g["lr"], g["weight_decay"]) if fold else None)
        if segs:
            self.model.mark_weight_planes_fresh()
        if hasattr(self.model, "planes_housekeeping") and flat.is_cuda and not torch.cuda.is_current_stream_capturing():
            self.model.planes_housekeeping()     # eager loops; a captured step is looked after by GraphedTrainStep.__call__
        if fold:
            self.step_count += 1
            return
        # close the step: count it on the device and prepare the next step's scalars -- one 1-thread launch.  (Folding it into
        # the flat update as "the last workgroup to arrive closes the step" was measured: 4096 arrivals on one atomic cost
        # +29 us on the launch to save this 5 us one, and a fenced variant for folded reductions 0.6 ms -- DESIGN.md dead ends.)
        ops.adamw_hyper_append(self._hyper, self._cumlog, self.step_count + 2, *cfg, step_dev=sd, advance=True)
        self.step_count += 1

    def state_dict(self, layout: str = "native"):
        """layout="native": {step, param_groups, m, v, table_m, table_v} (views of the live buffers);
        layout="torch": the dict torch.optim.AdamW.state_dict() would hold for the reference model (per-parameter
        step / exp_avg / exp_avg_sq in the reference's parameter order), so checkpoints interchange with the
        reference Trainer (trainer.py:153,186).  load_state_dict accepts either."""
        self._ensure_state()
        self.flush()
        sd = {"step": self.step_count, "param_groups": self.param_groups, "m": self._m, "v": self._v}
        if self.has_table:
            sd.update(table_m=self._tm, table_v=self._tv)
        if layout == "torch":
            out = native_to_torch_state(sd, self.model)
            lead = getattr(self, "empty_leading_group", None)
            if lead is not None:     # 4-key optim_args without trainable visual parameters: an empty modal group comes first
                g0 = {"lr": lead["lr"], "betas": tuple(self.param_groups[0]["betas"]), "eps": self.param_groups[0]["eps"],
                      "weight_decay": lead["weight_decay"], **_TORCH_GROUP_DEFAULTS, "params": []}
                out["param_groups"] = [g0] + out["param_groups"]
            return out
        if layout != "native":
            raise ValueError("layout must be 'native' or 'torch'")
        return sd

    def load_state_dict(self, sd):
        self._ensure_state()
        self.flush()                     # updates still owed under the OLD state are applied with the old state
        if is_torch_adamw_state(sd):     # a reference checkpoint (or layout="torch"): per-parameter state -> flat buffers
            sd = torch_to_native_state(sd, self.model)
        self.step_count = int(sd["step"])
        self.param_groups = sd["param_groups"]
        pairs = [(self._m, "m"), (self._v, "v")] + ([(self._tm, "table_m"), (self._tv, "table_v")] if self.has_table else [])
        for dst, key in pairs:
            dst.copy_(sd[key])
        self._step_dev.fill_(self.step_count)
        self._cumlog.zero_()
        self._dirty = False
        self._seeded_cfg = None          # the entry of step_count+1 is rewritten by the next step()
        if not self.has_table:
            return
        # a checkpoint is always flushed: every row is current through `step`; earlier per-step scalars are never
        # needed again, only the cumulative-log origin must be consistent (restart it at 0)
        self._last.fill_(self.step_count)
        self._step_dev.fill_(self.step_count)
        self._cumlog.zero_()
        self._dirty = False


def clip_grad_norm_(model, max_norm, norm_type=2.0, **_ignored):
    """torch.nn.utils.clip_grad_norm_ for this build's gradient layout (reference trainer.py:123-124, K18): the flat
    gradient buffer + the sparse table rows (absent rows are zeros of the dense gradient the reference clips) + the
    visual-encoder gradients.  Everything stays on the device (no .item()), so it is hipGraph-capturable.  Call it
    after `sync_gradients()`.  Returns the total norm as a 0-dim device tensor."""
    if float(norm_type) != 2.0:
        raise NotImplementedError("clip_grad_norm_: only the 2-norm is built (torch's default)")
    model = model.module if hasattr(model, "module") else model
    if hasattr(model, "wait_flat_grads"):
        model.wait_flat_grads()
    _, gflat = model.flat_parameters()
    sq = gflat.pow(2).sum()
    sp = getattr(model, "sparse_table_grad", None) if has_item_table(model) else None
    live = None
    if sp is not None:
        ar = torch.arange(sp.rows.shape[0], device=sp.rows.device, dtype=torch.int32)
        # slots beyond n / empty slots of a merged list are not rows.  A row-sharded model re-indexes sp.idx to the rows
        # THIS rank owns (0 elsewhere): the norm is over the merged global list (sp.gidx), identical on every rank
        ids = getattr(sp, "gidx", sp.idx)
        live = ((ar < sp.n) & (ids > 0)).unsqueeze(1)
        sq = sq + torch.where(live, sp.rows, 0.0).pow(2).sum()
    extra = [p.grad for n, p in model.named_parameters() if "visual_encoder" in n and p.grad is not None]
    for g in extra:
        sq = sq + g.pow(2).sum()
    total = sq.sqrt()
    coef = torch.clamp(float(max_norm) / (total + 1e-6), max=1.0)
    gflat.mul_(coef)
    if sp is not None:
        sp.rows.mul_(coef)
    for g in extra:
        g.mul_(coef)
    return total


class VisualAdamW:
    """torch.optim.AdamW semantics for the 'visual_encoder' parameter group (reference trainer.py:74-96: lr = modal_lr,
    weight_decay = modal_decay over every trainable visual_encoder parameter) on the fused flat kernel: the encoder's
    parameters live in one flat buffer (model/vit_native.NativeTower); each contiguous trainable segment is one
    pxr_adamw_flat_f32 launch (one segment for a tune_scale at a block boundary: blocks >= first trainable + rec_fc).
    A parameter that never receives a gradient (`post_layernorm`, replaced by Identity in the reference) is skipped,
    as torch skips `p.grad is None`.  `state_dict()` is torch.optim.AdamW's layout over the trainable parameters in
    registration order (what the reference stores for this group)."""

    VISUAL_HYPER_CAPACITY = 1 << 20      # optimizer steps of per-step scalars (16 MB + 8 MB), sized once: graph replays see no reallocation

    def __init__(self, encoder, lr=1e-4, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8):
        self.encoder = encoder
        self.tower = encoder._native
        self.param_groups = [{"lr": float(lr), "weight_decay": float(weight_decay), "betas": tuple(betas), "eps": float(eps)}]
        self.step_count = 0
        self._m = self._v = None
        # the step number lives ON THE DEVICE, as PxrAdamW's does: bias corrections come from a table of per-step scalars a
        # one-thread launch appends (ops.adamw_hyper_append), the flat launches read the entry of *_step_dev + 1 -- nothing of the
        # step depends on a host scalar, so a training step that includes this optimizer can be captured in a hipGraph
        self._hyper = self._cumlog = self._step_dev = None
        self._seeded_cfg = None

    def _state(self):
        t = self.tower
        t.ensure_packed()
        lo = min((a for a, _ in t.segments), default=0)
        dev = t.flat.device
        if self._m is None or self._m.device != dev or self._m.numel() != t.flat.numel() - lo:
            self._base = lo
            self._m = torch.zeros(t.flat.numel() - lo, dtype=torch.float32, device=dev)
            self._v = torch.zeros_like(self._m)
        if self._hyper is None or self._hyper.device != dev:
            self._hyper = torch.zeros(self.VISUAL_HYPER_CAPACITY, 4, dtype=torch.float32, device=dev)
            self._cumlog = torch.zeros(self.VISUAL_HYPER_CAPACITY, dtype=torch.float64, device=dev)
            self._step_dev = torch.full((1,), self.step_count, dtype=torch.int64, device=dev)
            self._seeded_cfg = None
        return t

    def zero_grad(self, set_to_none: bool = False):
        return None          # every backward overwrites the flat gradient buffer

    @torch.no_grad()
    def step(self):
        t = self._state()
        g = self.param_groups[0]
        b1, b2 = g["betas"]
        if self.step_count + 2 >= self.VISUAL_HYPER_CAPACITY:
            raise RuntimeError("VisualAdamW: per-step scalar table exhausted (raise VisualAdamW.VISUAL_HYPER_CAPACITY)")
        cfg = (g["lr"], b1, b2, g["eps"], g["weight_decay"])
        sd = self._step_dev
        if self._seeded_cfg != cfg:      # first step, after load_state_dict, or the hyper-parameters changed: (re)write this step's entry
            ops.adamw_hyper_append(self._hyper, self._cumlog, self.step_count + 1, *cfg, step_dev=sd)
            self._seeded_cfg = cfg
        for lo, hi in t.segments:
            ops.adamw_flat_tab(t.flat[lo:hi], t.gflat[lo:hi], self._m[lo - self._base:hi - self._base],
                               self._v[lo - self._base:hi - self._base], self._hyper, self.step_count + 1, b1, b2, g["eps"],
                               step_dev=sd)
        # close the step: count it on the device and prepare the next step's scalars (one 1-thread launch)
        ops.adamw_hyper_append(self._hyper, self._cumlog, self.step_count + 2, *cfg, step_dev=sd, advance=True)
        self.step_count += 1
        if hasattr(t, "drop_weight_planes"):
            t.drop_weight_planes(trainable_only=True)

    def _trainable(self):
        """[(name, parameter)] of the group in the reference's order: registration order, requires_grad only."""
        return [(n, p) for n, p in self.encoder.named_parameters() if p.requires_grad]

    def state_dict(self, layout: str = "torch"):
        t = self._state()
        names = self._trainable()
        state = {}
        if self.step_count > 0:
            for i, (n, p) in enumerate(names):
                if n not in t.views:      # no gradient ever reaches it (post_layernorm under 'mean' / 'cls', where the reference
                    continue              # has replaced it by Identity, load.py:112,116); under 'pool' it trains and is saved
                off, cnt, shape = t.views[n]
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self._m[off - self._base:off - self._base + cnt].view(shape).clone(),
                            "exp_avg_sq": self._v[off - self._base:off - self._base + cnt].view(shape).clone()}
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"],
                 **_TORCH_GROUP_DEFAULTS, "params": list(range(len(names)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        t = self._state()
        names = self._trainable()
        g = sd["param_groups"][0]
        ids = list(g["params"])
        if len(ids) != len(names):
            raise ValueError(f"visual optimizer state has {len(ids)} parameters, this encoder trains {len(names)}")
        self.param_groups = [{"lr": g["lr"], "weight_decay": g["weight_decay"], "betas": tuple(g["betas"]), "eps": g["eps"]}]
        self._m.zero_(); self._v.zero_()
        steps = set()
        for pid, (n, p) in zip(ids, names):
            st = sd["state"].get(pid)
            if st is None or n not in t.views:
                continue
            off, cnt, shape = t.views[n]
            steps.add(int(float(st["step"])))
            self._m[off - self._base:off - self._base + cnt].copy_(st["exp_avg"].reshape(-1))
            self._v[off - self._base:off - self._base + cnt].copy_(st["exp_avg_sq"].reshape(-1))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)})")
        self.step_count = steps.pop() if steps else 0
        self._step_dev.fill_(self.step_count)
        self._seeded_cfg = None          # the entry of step_count + 1 is rewritten by the next step()


class OptimizerGroup:
    """Several optimizers stepped together (reference trainer.py:86-96 builds ONE torch AdamW with a 'visual_encoder'
    group and a rec group; here the rec group is PxrAdamW and the visual-encoder group a torch AdamW)."""

    def __init__(self, *opts):
        self.opts = [o for o in opts if o is not None]

    def zero_grad(self, set_to_none: bool = True):
        for o in self.opts:
            o.zero_grad(set_to_none=set_to_none) if not isinstance(o, PxrAdamW) else o.zero_grad()

    def step(self):
        for o in self.opts:
            o.step()

    def flush(self):
        for o in self.opts:
            if hasattr(o, "flush"):
                o.flush()

    def _split(self):
        """(torch optimizers of the visual-encoder group, the PxrAdamW of the rec group)."""
        rec = [o for o in self.opts if isinstance(o, PxrAdamW)]
        vis = [o for o in self.opts if not isinstance(o, PxrAdamW)]
        if len(rec) != 1 or len(vis) > 1:
            raise ValueError("OptimizerGroup: expected [torch AdamW (visual_encoder group)], PxrAdamW (rec group)")
        return (vis[0] if vis else None), rec[0]

    def state_dict(self, layout: str = "native"):
        """layout="torch": ONE torch.optim.AdamW-shaped dict with the reference's two param groups (trainer.py:86-96:
        group 0 = trainable visual_encoder parameters, group 1 = rec parameters; state indices run through both)."""
        if layout == "native":
            return {"group": [(o.state_dict() if not isinstance(o, PxrAdamW) else o.state_dict(layout="native"))
                              for o in self.opts]}
        vis, rec = self._split()
        v = vis.state_dict() if vis is not None else {"state": {}, "param_groups": []}
        n0 = sum(len(g["params"]) for g in v["param_groups"])
        r = rec.state_dict(layout="torch") if n0 == 0 else native_to_torch_state(rec.state_dict(), rec.model, first_index=n0)
        return {"state": {**v["state"], **r["state"]}, "param_groups": list(v["param_groups"]) + r["param_groups"]}

    def load_state_dict(self, sd):
        if is_torch_adamw_state(sd):
            vis, rec = self._split()
            if vis is not None:
                g0 = sd["param_groups"][0]
                vis.load_state_dict({"state": {i: sd["state"][i] for i in g0["params"] if i in sd["state"]},
                                     "param_groups": [g0]})
            rec.load_state_dict(torch_to_native_state(sd, rec.model, group_index=-1))
            return
        for o, s in zip(self.opts, sd["group"]):
            o.load_state_dict(s)


class FragmentAdamW:
    """The reference's `decay_check_name` optimizer (code/REC/trainer/trainer.py:73-91): ONE torch.optim.AdamW whose two param
    groups are split by a NAME FRAGMENT instead of by 'visual_encoder' -- group 0 = every trainable parameter whose name (as the
    DDP-wrapped model yields it: 'module.' + the state_dict key) contains the fragment, updated with (modal_lr, modal_decay); group 1
    = all the others with (rec_lr, rec_decay).  No shipped YAML sets it; it exists so that such a config runs instead of raising.

    The fragment can cut through the flat parameter buffer anywhere ('LayerNorm', 'bias', 'item_embedding', ...), so this optimizer
    works per parameter tensor: one `pxr_adamw_flat_f32` launch per tensor with its group's scalars computed on the host (torch's
    single-tensor formulae: weight decay, moments, bias corrections -- the same kernel the flat launch of PxrAdamW runs), the item
    table through the dense sweep `pxr_adamw_table_f32` (dense AdamW semantics on the sparse row gradient).  Eager only (host
    scalars: not capturable) and nothing else keeps operand planes current: the model re-splits its weights every forward."""

    def __init__(self, dp_model, fragment, in_lr, in_decay, out_lr, out_decay, betas=(0.9, 0.999), eps=1e-8):
        self.dp = dp_model
        self.model = dp_model.module if hasattr(dp_model, "module") else dp_model
        self.fragment = str(fragment)
        self.betas, self.eps = tuple(betas), float(eps)
        self.param_groups = [{"lr": float(in_lr), "weight_decay": float(in_decay), "betas": self.betas, "eps": self.eps},
                             {"lr": float(out_lr), "weight_decay": float(out_decay), "betas": self.betas, "eps": self.eps}]
        self.step_count = 0
        self._state = {}          # name -> (m, v)
        self._slot = None
        if hasattr(self.model, "trust_optimizer_planes"):
            self.model.trust_optimizer_planes = False
        self.has_table = has_item_table(self.model)

    def _named(self):
        """[(name as the reference's DDP-wrapped model yields it, parameter)] of the trainable parameters, named_parameters() order."""
        pre = "module." if hasattr(self.dp, "module") else ""
        return [(pre + n, p) for n, p in self.model.named_parameters() if p.requires_grad]

    def groups(self):
        """-> (names of group 0, names of group 1), each in named_parameters() order (trainer.py:73-85)."""
        named = self._named()
        return [n for n, _ in named if self.fragment in n], [n for n, _ in named if self.fragment not in n]

    def zero_grad(self, set_to_none: bool = False):
        return None               # every backward overwrites the gradient buffers

    def flush(self):
        return None               # dense table semantics: nothing is ever owed

    @torch.no_grad()
    def step(self):
        if hasattr(self.model, "wait_flat_grads"):
            self.model.wait_flat_grads()
        self.step_count += 1
        b1, b2 = self.betas
        for name, p in self._named():
            g = self.param_groups[0 if self.fragment in name else 1]
            st = self._state.get(name)
            if st is None:
                st = self._state[name] = (torch.zeros_like(p.data), torch.zeros_like(p.data))
            if self.has_table and p is self.model.item_embedding.weight:
                if self._slot is None or self._slot.numel() != p.shape[0] or self._slot.device != p.device:
                    self._slot = torch.empty(p.shape[0], dtype=torch.int32, device=p.device)
                    ops.slot_fill(self._slot, -1)
                ops.adamw_table(p.data, st[0], st[1], self._slot, self.model.sparse_table_grad, g["lr"], b1, b2, self.eps,
                                g["weight_decay"], self.step_count)
                continue
            if p.grad is None:
                continue          # torch skips parameters without a gradient
            if p.numel() % 4 != 0 or not p.data.is_contiguous() or not p.grad.is_contiguous():
                raise NotImplementedError(f"FragmentAdamW: parameter {name} ({tuple(p.shape)}) is not a contiguous multiple of 4 floats")
            ops.adamw_flat(p.data, p.grad, st[0], st[1], g["lr"], b1, b2, self.eps, g["weight_decay"], self.step_count)
        enc = getattr(getattr(self.model, "visual_encoder", None), "_native", None)
        if enc is not None and hasattr(enc, "drop_weight_planes"):
            enc.drop_weight_planes(trainable_only=True)

    def _order(self):
        g0, g1 = self.groups()
        return g0 + g1            # torch numbers the per-parameter state through the groups in order

    def state_dict(self, layout: str = "torch"):
        """The dict torch.optim.AdamW.state_dict() holds for the reference's two fragment groups (both layouts return it)."""
        g0, g1 = self.groups()
        state = {}
        for i, n in enumerate(g0 + g1):
            if n in self._state:
                state[i] = {"step": torch.tensor(float(self.step_count)), "exp_avg": self._state[n][0].clone(),
                            "exp_avg_sq": self._state[n][1].clone()}
        groups = []
        for gi, (names, first) in enumerate(((g0, 0), (g1, len(g0)))):
            g = self.param_groups[gi]
            groups.append({"lr": g["lr"], "betas": self.betas, "eps": self.eps, "weight_decay": g["weight_decay"],
                           **_TORCH_GROUP_DEFAULTS, "params": list(range(first, first + len(names)))})
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd):
        order = self._order()
        ids = [i for g in sd["param_groups"] for i in g["params"]]
        if len(ids) != len(order):
            raise ValueError(f"optimizer state has {len(ids)} parameters, this model trains {len(order)}")
        named = dict(self._named())
        steps = set()
        self._state = {}
        for pid, n in zip(ids, order):
            st = sd["state"].get(pid)
            if st is None:
                continue
            steps.add(int(float(st["step"])))
            dev = named[n].device
            self._state[n] = (st["exp_avg"].to(dev, torch.float32).clone().view_as(named[n]),
                              st["exp_avg_sq"].to(dev, torch.float32).clone().view_as(named[n]))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ ({sorted(steps)})")
        self.step_count = steps.pop() if steps else 0
        for gi, g in enumerate(sd["param_groups"][:2]):
            self.param_groups[gi].update(lr=g["lr"], weight_decay=g["weight_decay"])
