"""hipGraph capture of the whole training step (forward + backward + gradient exchange + optimizer).

At the reference batch size (64 sequences) a step is ~120 small kernels; issuing them from Python costs more host
time (~1.7 ms) than the GPU needs to run them (~1.3 ms).  Everything a step needs that changes from step to step --
the batch, the dropout seed offset, the optimizer step number -- lives in device memory, so the step can be captured
once and replayed: the host cost per step drops to one `hipGraphLaunch`.

    gstep = GraphedTrainStep(dp_model, optimizer, items_example, mask_example)
    loss = gstep(items, mask)          # device scalar; same semantics as the eager sequence
"""
from __future__ import annotations

import torch


class GraphedTrainStep:
    def __init__(self, dp_model, optimizer, items, masked_index, warmup=3, clip_grad_norm=None, loss_sum=None,
                 lookahead=False, h2_stale_scales=None):
        self.dp, self.opt = dp_model, optimizer
        self.loss_sum = loss_sum          # optional 0-dim device tensor: every replay adds its loss (epoch totals
                                          # without an eager add per step)
        self.clip = clip_grad_norm        # dict(max_norm=..., norm_type=2) as in the reference YAML, or None
        self.model = dp_model.module if hasattr(dp_model, "module") else dp_model
        # every step goes through optimizer.step(): the join with the weight-gradient side stream can wait until the
        # flat gradient is consumed (seqcore.SeqRecCore.wait_flat_grads) -- inside the capture, so the graph is closed
        self.model.defer_weight_grad_join = True
        if hasattr(self.model, "split_catch_up") and getattr(self.model, "_split_env", None) is None:
            self.model.split_catch_up = True             # sasrec.SASRec._forward_train: fork / join are graph edges here
        if hasattr(self.model, "trust_optimizer_planes"):
            self.model.trust_optimizer_planes = True     # the captured forward has no split launch (seqcore._weight_planes)
        if hasattr(self.model, "h2_stale_scales"):
            # ... and its backward no gradient split launches (seqcore, ops.H2Sites).  An owner that manages the switch itself (the
            # Trainer: exact scales for a while after an overflow) says which it wants; None: on
            self.model.h2_stale_scales = True if h2_stale_scales is None else bool(h2_stale_scales)
        # the batch lives in ONE buffer (ids | mask): a caller that hands over two views of one packed tensor (bench.py,
        # the trainer's batcher) pays one copy per step instead of two
        n_i = items.numel()
        if items.dtype == torch.int64 and masked_index.dtype == torch.int64:
            self._packed = torch.empty(n_i + masked_index.numel(), dtype=torch.int64, device=items.device)
            self.items = self._packed[:n_i].view(items.shape)
            self.mask = self._packed[n_i:].view(masked_index.shape)
        else:
            # a batch whose first member is not an id tensor (the pixel model: images [B, 2 (L+1), 3, H, W] fp32): two buffers
            self._packed = None
            self.items, self.mask = torch.empty_like(items), torch.empty_like(masked_index)
        self.items.copy_(items)
        self.mask.copy_(masked_index)
        # look-ahead: the NEXT batch's ids (SASRec.set_next_batch) -- its table rows are caught up beside this step's
        # GEMMs.  Without a next batch the buffer holds this batch again (nothing left to replay: two light launches).
        self.items_next = items.clone() if (lookahead and hasattr(self.model, "set_next_batch")) else None
        self._one = torch.ones((), dtype=torch.float32, device=items.device)   # d(loss)/d(loss): no fill kernel per step
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):          # eager warm-up: sizes every persistent buffer / workspace
            for _ in range(warmup):
                self._eager()
            if warmup == 0:
                # size the buffers WITHOUT consuming a training step: run one eager step on a snapshot of all state and
                # restore it (a trainer that captures in the middle of an epoch must not train on a batch twice)
                self._dry_run()
        cur.wait_stream(s)
        torch.cuda.synchronize()
        host_state = self._host_counters()
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: another host thread (the trainer's batch prefetcher pins host memory) may call into the
        # runtime while this thread captures; only THIS thread's unsafe calls should invalidate the capture
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):
            self.loss = self._eager().detach()
        # stream capture RECORDS the kernels without running them: the device counters did not advance, so the host
        # mirrors that the Python code bumped during capture are rolled back
        self._set_host_counters(host_state)

    def _opts(self):
        """The optimizers behind self.opt (an OptimizerGroup steps several: the pixel model's visual-encoder and rec groups)."""
        return list(getattr(self.opt, "opts", [self.opt]))

    def _host_counters(self):
        """Host mirrors of counters that live on the device (optimizer step numbers, the dropout step, the lazy table's dirty flag)."""
        return ([(o.step_count, getattr(o, "_dirty", None)) for o in self._opts()], self.model._step_counter)

    def _set_host_counters(self, st):
        for o, (sc, dirty) in zip(self._opts(), st[0]):
            o.step_count = sc
            if dirty is not None:
                o._dirty = dirty
        self.model._step_counter = st[1]

    def _dry_run(self):
        m, o = self.model, self.opt
        if hasattr(o, "opts"):
            raise NotImplementedError("GraphedTrainStep(warmup=0) snapshots ONE PxrAdamW; capture an optimizer group with warmup >= 1")
        flat, _ = m.flat_parameters()
        o._ensure_state()
        keep = [t.clone() for t in (flat, o._m, o._v)]
        table_state = None
        if getattr(o, "has_table", False):
            o.flush()
            table_state = [t.clone() for t in (m.item_embedding.weight.data, o._tm, o._tv, o._last)]
        counters = (o.step_count, m._step_counter, o._dirty, o._step_dev.clone(), m._drop_dev.clone())
        acc = self.loss_sum.clone() if self.loss_sum is not None else None
        self._eager()
        if acc is not None:
            self.loss_sum.copy_(acc)
        for dst, src in zip((flat, o._m, o._v), keep):
            dst.copy_(src)
        if table_state is not None:
            for dst, src in zip((m.item_embedding.weight.data, o._tm, o._tv, o._last), table_state):
                dst.copy_(src)
        o.step_count, m._step_counter, o._dirty = counters[:3]
        o._step_dev.copy_(counters[3])
        m._drop_dev.copy_(counters[4])
        if hasattr(m, "refresh_weight_planes"):
            m.refresh_weight_planes()     # the weights were restored behind the optimizer's back: its planes are stale

    def _eager(self):
        self.opt.zero_grad()
        if self.items_next is not None:
            self.model.set_next_batch(self.items_next)
        loss = self.dp((self.items, self.mask))
        loss.backward(self._one)
        if self.loss_sum is not None:
            self.loss_sum.add_(loss.detach().view(()))
        if hasattr(self.dp, "sync_gradients"):
            self.dp.sync_gradients(defer_flat=True)
        if self.clip:
            from .optim import clip_grad_norm_

            clip_grad_norm_(self.model, **self.clip)
        self.opt.step()
        return loss

    def matches(self, items, masked_index) -> bool:
        return items.shape == self.items.shape and masked_index.shape == self.mask.shape

    @staticmethod
    def pack(items, masked_index):
        """(items, masked_index) as two views of one packed int64 tensor (what __call__ copies in one go)."""
        pk = torch.cat([items.reshape(-1), masked_index.reshape(-1).to(torch.int64)])
        n_i = items.numel()
        return pk[:n_i].view(items.shape), pk[n_i:].view(masked_index.shape)

    def __call__(self, items, masked_index, next_items=None):
        if (self._packed is not None and items.dtype == torch.int64 and masked_index.dtype == torch.int64 and items.is_contiguous()
                and masked_index.is_contiguous() and items.device == masked_index.device
                and masked_index.data_ptr() == items.data_ptr() + 8 * items.numel()
                and items.untyped_storage().data_ptr() == masked_index.untyped_storage().data_ptr()):
            src = torch.as_strided(items, (self._packed.numel(),), (1,))      # the packed range, one copy
            self._packed.copy_(src, non_blocking=True)
        else:
            self.items.copy_(items, non_blocking=True)
            self.mask.copy_(masked_index, non_blocking=True)
        if self.items_next is not None:
            nxt = next_items if (next_items is not None and next_items.shape == self.items.shape) else items
            self.items_next.copy_(nxt, non_blocking=True)
        self.graph.replay()
        for o in self._opts():
            o.step_count += 1
            if hasattr(o, "_dirty"):
                o._dirty = True
        self.model._step_counter += 1
        if hasattr(self.model, "planes_housekeeping"):
            self.model.planes_housekeeping()     # every PXR_H2_REFRESH_STEPS replays: exponents / statistics of the weight planes
                                                 # re-derived from the values, status word polled (no stream stall)
        return self.loss
