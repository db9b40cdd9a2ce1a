"""Trainer -- the reference's epoch loop (code/REC/trainer/trainer.py:19-409) around the MI355X-native step.

Same surface: `Trainer(config, model)` with `model` a DDP-shaped wrapper exposing `.module`; `fit(train, valid,
saved, show_progress)`; `evaluate(loader, load_best_model, model_file)`; `resume_checkpoint(file)`; same
checkpoint dict keys (`config, epoch, cur_step, best_valid_score, state_dict, optimizer, rng_state,
cuda_rng_state`, trainer.py:146-155) with `state_dict` under the reference's parameter names and `optimizer` in
torch.optim.AdamW.state_dict()'s layout over the reference's parameter order (optim.native_to_torch_state), so `.pth`
files interchange in both directions (what does NOT transfer: the dropout stream -- `dropout_step` here, torch's
generator state there); early stopping / eval_step / valid_metric logic
as trainer.py:256-325; metric averaging = per-rank SUM -> all_gather -> / #users -> round (trainer.py:360-364,
399-406).

What changed for the hardware (SURVEY.md §7 hard part 6):
  * the loss stays on the device: the per-step `.item()` + isnan host sync (trainer.py:120-121) becomes one sync
    per epoch (NaN still raises ValueError('Training loss is nan'));
  * zero_grad is a no-op (gradients are overwritten), backward leaves the table gradient sparse, gradients are
    exchanged by parallel.GradSync, the optimizer is optim.PxrAdamW;
  * full-sort evaluation uses the fused scoring+mask+top-K kernel; `eval_fused_topk: False` falls back to the
    reference's literal sequence (predict -> mask -> torch.topk) for cross-checking.
"""
from __future__ import annotations

import os
import queue
import threading
from logging import getLogger
from time import time

import numpy as np
import torch

from .. import ops
from ..evaluator import Collector, Evaluator
from ..optim import OptimizerGroup, PxrAdamW, clip_grad_norm_
from ..parallel import world_info
from ..utils import calculate_valid_score, dict2str, early_stopping, ensure_dir, get_local_time


def host_threads_for(config) -> int:
    """Intra-op CPU threads the Trainer sets for the host side of its loop (0: leave torch's setting alone): `host_threads` of the
    YAML if given; otherwise 1 unless the caller exported OMP_NUM_THREADS (see Trainer.__init__)."""
    ht = config["host_threads"]
    if ht is None:
        ht = 0 if os.environ.get("OMP_NUM_THREADS") else 1
    return int(ht)


class _Prefetcher:
    """Builds host batches on a background thread (numpy releases the GIL) and stages them on the device."""

    def __init__(self, loader, device, depth=4):
        self.loader, self.device, self.q = loader, device, queue.Queue(maxsize=depth)
        self.t = threading.Thread(target=self._run, daemon=True)
        self.t.start()

    def _run(self):
        try:
            pin = self.device.type == "cuda"
            for batch in self.loader:   # pinned staging (in this thread) makes the consumer's H2D copies asynchronous
                if (len(batch) == 2 and all(isinstance(x, torch.Tensor) and x.dtype == torch.int64 for x in batch)):
                    # (items, masked_index): ONE packed host tensor -> one H2D copy, and the graphed step copies the two
                    # device views into its static buffer in one go (graph.GraphedTrainStep.__call__)
                    pk = torch.cat([batch[0].reshape(-1), batch[1].reshape(-1)])
                    batch = ("packed", pk.pin_memory() if pin else pk, batch[0].shape, batch[1].shape)
                elif pin:
                    batch = tuple(x.pin_memory() if isinstance(x, torch.Tensor) else x for x in batch)
                self.q.put(batch)
        except BaseException as e:  # surfaced in the consumer
            self.q.put(e)
        self.q.put(None)

    def __iter__(self):
        while True:
            b = self.q.get()
            if b is None:
                return
            if isinstance(b, BaseException):
                raise b
            if isinstance(b[0], str):        # packed (items | masked_index)
                d = b[1].to(self.device, non_blocking=True)
                n_i = int(torch.Size(b[2]).numel())
                yield d[:n_i].view(b[2]), d[n_i:].view(b[3])
                continue
            if isinstance(b[1], tuple):      # device sampler: (positive windows, (seed, batch counter))
                pos = b[0].to(self.device, non_blocking=True)
                yield ops.sample_negatives(pos, self.loader.item_num, *b[1])
                continue
            yield tuple(x.to(self.device, non_blocking=True) for x in b)


class Trainer:
    def __init__(self, config, model):
        self.config = config
        self.model = model
        self.logger = getLogger()
        self.optim_args = config["optim_args"]
        self.epochs = config["epochs"]
        self.eval_step = min(config["eval_step"], self.epochs)
        self.stopping_step = config["stopping_step"]
        self.clip_grad_norm = config["clip_grad_norm"]
        self.valid_metric = config["valid_metric"].lower()
        self.valid_metric_bigger = config["valid_metric_bigger"]
        self.test_batch_size = config["eval_batch_size"]
        self.device = config["device"]
        self.rank, self.world = world_info()
        self.checkpoint_dir = config["checkpoint_dir"] or "saved"
        if self.rank == 0:
            ensure_dir(self.checkpoint_dir)
        self.saved_model_file = os.path.join(self.checkpoint_dir, "{}-{}.pth".format(config["model"], get_local_time()))
        self.use_modality = config["use_modality"]
        self.start_epoch = 0
        self.cur_step = 0
        self.best_valid_score = -np.inf if self.valid_metric_bigger else np.inf
        self.best_valid_result = None
        self.train_loss_dict = {}
        self.optimizer = self._build_optimizer()
        self.eval_collector = Collector(config)
        self.evaluator = Evaluator(config)
        self.item_feature = None
        self.tot_item_num = None
        self.fused_topk = config["eval_fused_topk"] is None or bool(config["eval_fused_topk"])
        # The host side of the loop is index arithmetic on small arrays (batcher, packing, pinned staging).  With the intra-op pool
        # torch sizes by default -- one thread per core, 256 on the GPU hosts -- every torch.cat / copy above ATen's grain size wakes
        # that whole team: measured on the Pixel200K-shaped run, B = 512 batches took 32 ms each to build instead of 3 (18 K instead
        # of 141 K sequences/s; B = 64 stays below the grain size and never showed it).  The reference's launcher pins
        # OMP_NUM_THREADS=1 for the same loop (main.py:5), and so does this build's main.py; a Trainer driven from a notebook or
        # run.build() gets the same here unless the caller chose a thread count (OMP_NUM_THREADS set, or `host_threads` in the YAML;
        # host_threads: 0 leaves torch's setting alone).
        ht = host_threads_for(config)
        if int(ht) > 0 and torch.get_num_threads() != int(ht):
            self.logger.info("host threads of torch's intra-op pool: %d -> %d (host_threads / OMP_NUM_THREADS override)",
                             torch.get_num_threads(), int(ht))
            torch.set_num_threads(int(ht))
        # whole-step hipGraph replay for full-size batches (ID model): the host issues one graph launch per step instead of
        # ~50 kernel launches.  Data parallel over RCCL too (round 6): the captured step CONTAINS its collectives (row all-gather,
        # flat all-reduce) -- an eager multi-rank step is bound by the host (1.47 ms against 0.85 ms replayed, 1-rank RCCL group on
        # one MI355X, profiles/r06) and the reference's DDP hides its exchange under the backward pass (run.py:40).  Not over gloo
        # (CPU staging cannot be captured).  `use_hip_graph: False` keeps the eager sequence.
        g = config["use_hip_graph"]
        sharded = getattr(model.module, "_sharded", False)      # collectives inside forward: issued eagerly
        import torch.distributed as _dist
        rccl = self.world > 1 and _dist.is_initialized() and _dist.get_backend() == "nccl"
        self.use_graph = (g is None or bool(g)) and (self.world == 1 or rccl) and not self.use_modality and not sharded
        self._gstep = None
        # every training step ends in optimizer.step(), which joins the weight-gradient side stream itself
        if hasattr(model.module, "defer_weight_grad_join"):
            model.module.defer_weight_grad_join = True
            if hasattr(model.module, "trust_optimizer_planes"):
                model.module.trust_optimizer_planes = True   # only the optimizer / load_state_dict touch the weights here
            if hasattr(model.module, "h2_stale_scales"):
                model.module.h2_stale_scales = True          # consecutive training steps: gradient planes under the previous step's scales
        if type(self.optimizer).__name__ == "FragmentAdamW":     # decay_check_name: host scalars per tensor, no plane upkeep
            self.use_graph = False
            if hasattr(model.module, "trust_optimizer_planes"):
                model.module.trust_optimizer_planes = False

    # ---------------------------------------------------------------------------------------------- optimizer
    def _build_optimizer(self):
        """trainer.py:66-103.  A 4-key optim_args selects separate modal/rec groups; for the ID model every
        parameter is a 'rec' parameter, so both spellings resolve to one AdamW."""
        a = self.optim_args
        m = self.model.module
        if len(a) == 4 and self.config["decay_check_name"]:
            # trainer.py:73-91: the two groups split by a name fragment instead of by 'visual_encoder' (no shipped YAML sets it):
            # per-tensor launches with host scalars -- eager steps, planes re-split every forward
            from ..optim import FragmentAdamW

            return FragmentAdamW(self.model, self.config["decay_check_name"], a["modal_lr"], a["modal_decay"], a["rec_lr"],
                                 a["rec_decay"])
        if len(a) == 4:
            rec = PxrAdamW(m, lr=a["rec_lr"], weight_decay=a["rec_decay"])
            modal = [p for n, p in m.named_parameters() if "visual_encoder" in n and p.requires_grad]
            if not modal:
                # the reference still builds TWO param groups here (an empty modal group + the rec group, trainer.py:93-96):
                # keep that shape in torch-layout checkpoints so they load into its optimizer
                rec.empty_leading_group = {"lr": a["modal_lr"], "weight_decay": a["modal_decay"]}
                return rec
            # the visual-encoder group (trainer.py:86-89): the same fused AdamW kernel over the encoder's flat buffer
            from ..optim import VisualAdamW

            return OptimizerGroup(VisualAdamW(m.visual_encoder, lr=a["modal_lr"], weight_decay=a["modal_decay"]), rec)
        return PxrAdamW(m, lr=a["learning_rate"], weight_decay=a["weight_decay"])

    # ---------------------------------------------------------------------------------------------- training
    def _train_epoch(self, train_data, epoch_idx, loss_func=None, show_progress=False):
        self.model.train()
        total = torch.zeros((), dtype=torch.float32, device=self.device)
        self._one = torch.ones((), dtype=torch.float32, device=self.device)
        # opt-in (`lookahead_rows: True`): see SASRec.set_next_batch for why it is off by default on MI355X
        lookahead = (bool(self.config["lookahead_rows"]) and hasattr(self.model.module, "set_next_batch")
                     and not self.use_modality)

        def with_next(it):
            """(batch, ids of the batch after it | None): one batch of look-ahead out of the prefetch queue."""
            prev = None
            for cur in it:
                if prev is not None:
                    yield prev, (cur[0] if lookahead and cur[0].shape == prev[0].shape else None)
                prev = cur
            if prev is not None:
                yield prev, None

        for data, next_items in with_next(_Prefetcher(train_data, self.device)):
            self._steps_done = getattr(self, "_steps_done", 0) + 1
            self._h2_stale_resume()
            if self.use_graph:
                if self._gstep is None and data[0].shape[0] == self.config["train_batch_size"]:
                    from ..graph import GraphedTrainStep

                    if getattr(self, "_graph_loss", None) is None:      # (kept across re-captures: a step captured in the middle
                        self._graph_loss = torch.zeros((), dtype=torch.float32, device=self.device)   # of an epoch adds to the same sum)
                    self._gstep = GraphedTrainStep(self.model, self.optimizer, data[0], data[1], warmup=0,
                                                   clip_grad_norm=self.clip_grad_norm, loss_sum=self._graph_loss,
                                                   lookahead=lookahead,
                                                   h2_stale_scales=getattr(self.model.module, "h2_stale_scales", None))
                if self._gstep is not None and self._gstep.matches(data[0], data[1]):
                    # the replay adds its loss to self._graph_loss on the device
                    try:
                        self._gstep(data[0], data[1], next_items=next_items)
                    except ops.H2StaleOverflow as e:      # raised by the status poll AFTER the replay: the step itself has run
                        self._h2_stale_fallback(e)
                    continue
            self.optimizer.zero_grad()
            if self.use_modality:   # assemble the image batch on the device from the HBM-resident store
                from ..data.images import interleave_pos_neg

                data = (self._image_store(train_data).batch(interleave_pos_neg(data[0])), data[1])
            elif next_items is not None:
                self.model.module.set_next_batch(next_items)   # its table rows are caught up beside this step's GEMMs
            losses = self.model(data)
            losses.backward(self._one)     # preallocated unit gradient: no ones_like fill per step
            if hasattr(self.model, "sync_gradients"):
                self.model.sync_gradients(defer_flat=True)    # the consumers below wait for it
            if self.clip_grad_norm:                                   # trainer.py:123-124
                clip_grad_norm_(self.model, **self.clip_grad_norm)
            try:
                self.optimizer.step()
            except ops.H2StaleOverflow as e:              # (the optimizer's housekeeping polls the status word; the step has run)
                self._h2_stale_fallback(e)
            total = total + losses.detach()
        if getattr(self, "_graph_loss", None) is not None:     # (also when the captured step was dropped on the epoch's last steps)
            total = total + self._graph_loss
            self._graph_loss.zero_()
        total_loss = float(total.item())          # the only host sync of the epoch
        try:
            ops.raise_on_bad_indices(self.device)     # ... and where an out-of-catalogue item id surfaces (IndexError)
        except ops.H2StaleOverflow as e:
            self._h2_stale_fallback(e)
        self._check_nan(total_loss)
        return total_loss

    H2_STALE_BACKOFF = (256, 1 << 16)      # steps on exact scales after the first overflow; the cap of the (x 4) back-off

    def _h2_stale_fallback(self, err):
        """A gradient outgrew the headroom of its stale h2 scale in ONE step (its largest elements were saturated there, nothing
        non-finite was written): say so and go on with per-step exact scales -- the captured step is dropped and re-captured.
        One rank: for a while -- gradient magnitudes move fastest in the first steps of a run and after a learning-rate change;
        after H2_STALE_BACKOFF[0] steps (x 4 with every further overflow) the stale scales are re-seeded by an exact pass and
        switched on again (_h2_stale_resume).  Data parallel: for good (see below)."""
        m = self.model.module
        if getattr(m, "h2_stale_scales", False):
            m.h2_stale_scales = False
            self._gstep = None
            if self.world > 1:
                # a re-capture runs a dry step WITH its collectives, which the other ranks would not match: this rank issues its steps
                # eagerly from here on (the same collectives in the same order as a replaying rank's)
                self.use_graph = False
                self.logger.warning("%s -- continuing with exact per-step scales", err)
                return
            wait = getattr(self, "_stale_backoff", self.H2_STALE_BACKOFF[0])
            self._stale_backoff = min(4 * wait, self.H2_STALE_BACKOFF[1])
            self._stale_resume_at = getattr(self, "_steps_done", 0) + wait
            self.logger.warning("%s -- continuing with exact per-step scales for the next %d steps", err, wait)

    def _h2_stale_resume(self):
        """Back to the stale scales once the back-off of _h2_stale_fallback has run out: their state is re-seeded by the next
        step's exact pass (the dry step of the re-capture), the captured step is re-captured on them."""
        at = getattr(self, "_stale_resume_at", None)
        if at is None or getattr(self, "_steps_done", 0) < at:
            return
        self._stale_resume_at = None
        m = self.model.module
        if self.world > 1 or not hasattr(m, "h2_stale_scales") or m.h2_stale_scales:
            return
        m.h2_stale_scales = True
        sites = getattr(m, "_h2_sites", None)
        if sites is not None:
            sites.seeded_for = None
            sites.run_max.zero_()
        self._gstep = None
        self.logger.info("fp16 two-plane gradients back under the recent steps' scales (re-seeded)")

    def _check_nan(self, loss):
        if np.isnan(loss):
            raise ValueError("Training loss is nan")

    def _valid_epoch(self, valid_data, show_progress=False):
        self._barrier()
        valid_result = self.evaluate(valid_data, load_best_model=False, show_progress=show_progress)
        valid_score = calculate_valid_score(valid_result, self.valid_metric)
        self._barrier()
        return valid_score, valid_result

    def _barrier(self):
        if self.world > 1:
            torch.distributed.barrier()

    def _checkpoint_tensors(self):
        """(model state_dict, optimizer state) on the host.  A row-sharded model gathers its table and the table
        moments with collectives, so EVERY rank must get here; otherwise only rank 0 does the work."""
        m = self.model.module
        sharded = getattr(m, "_sharded", False)
        if not sharded and self.rank != 0:
            return None, None
        model_sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
        if sharded:
            from ..model.sharded import optimizer_state_full

            rec = self.optimizer.opts[-1] if isinstance(self.optimizer, OptimizerGroup) else self.optimizer
            from ..optim import native_to_torch_state

            opt_sd = native_to_torch_state(optimizer_state_full(rec, m), m)
        else:
            opt_sd = self.optimizer.state_dict(layout="torch")
        # torch.optim.AdamW.state_dict() layout (reference trainer.py:153): per-parameter tensors moved to the host
        opt_sd = {"state": {i: {k: (v.detach().cpu() if isinstance(v, torch.Tensor) else v) for k, v in st.items()}
                            for i, st in opt_sd["state"].items()}, "param_groups": opt_sd["param_groups"]}
        return model_sd, opt_sd

    def _save_checkpoint(self, epoch, verbose=True):
        model_sd, opt_sd = self._checkpoint_tensors()
        if self.rank == 0:
            state = {
                "config": dict(self.config.final_config_dict) if hasattr(self.config, "final_config_dict") else dict(self.config),
                "epoch": epoch,
                "cur_step": self.cur_step,
                "best_valid_score": self.best_valid_score,
                "state_dict": model_sd,
                "optimizer": opt_sd,
                # this path's dropout "RNG state" (the reference stores torch's generator states for the same purpose)
                "dropout_step": self.model.module.dropout_step() if hasattr(self.model.module, "dropout_step") else 0,
                "rng_state": torch.get_rng_state(),
                "cuda_rng_state": torch.cuda.get_rng_state() if torch.cuda.is_available() else None,
            }
            state["config"].pop("device", None)
            torch.save(state, self.saved_model_file)
            if verbose:
                self.logger.info(f"Saving current: {self.saved_model_file}")
        self._barrier()

    def resume_checkpoint(self, resume_file):
        """trainer.py:164-190: restores epoch / step / best score / optimizer / RNG (NOT the model weights,
        exactly like the reference; load them with evaluate(load_best_model=True) or load_state_dict)."""
        checkpoint = torch.load(str(resume_file), map_location="cpu", weights_only=False)
        self.start_epoch = checkpoint["epoch"] + 1
        self.cur_step = checkpoint["cur_step"]
        self.best_valid_score = checkpoint["best_valid_score"]
        if str(checkpoint["config"]["model"]).lower() != str(self.config["model"]).lower():
            self.logger.warning("Architecture configuration given in config file is different from that of checkpoint.")
        if getattr(self.model.module, "_sharded", False):
            from ..model.sharded import load_optimizer_state_full

            load_optimizer_state_full(self.optimizer, self.model.module, checkpoint["optimizer"])
        else:
            self.optimizer.load_state_dict(checkpoint["optimizer"])
        if hasattr(self.model.module, "set_dropout_step"):
            self.model.module.set_dropout_step(checkpoint.get("dropout_step", 0))
        self._gstep = None          # a captured step graph belongs to the pre-resume state
        torch.set_rng_state(checkpoint["rng_state"])
        if checkpoint.get("cuda_rng_state") is not None and torch.cuda.is_available():
            torch.cuda.set_rng_state(checkpoint["cuda_rng_state"])
        self.logger.info("Checkpoint loaded. Resume training from epoch {}".format(self.start_epoch))

    def fit(self, train_data, valid_data=None, verbose=True, saved=True, show_progress=False, callback_fn=None):
        if saved and self.start_epoch >= self.epochs:
            self._save_checkpoint(-1, verbose=verbose)
        for epoch_idx in range(self.start_epoch, self.epochs):
            if self.config["need_training"] is None or self.config["need_training"]:
                train_data.sampler.set_epoch(epoch_idx)
                t0 = time()
                train_loss = self._train_epoch(train_data, epoch_idx, show_progress=show_progress)
                self.train_loss_dict[epoch_idx] = train_loss
                t1 = time()
                if verbose:
                    des = self.config["loss_decimal_place"] or 4
                    self.logger.info(("epoch %d training [time: %.2fs, train loss: %." + str(des) + "f]")
                                     % (epoch_idx, t1 - t0, train_loss))
            if self.eval_step <= 0 or not valid_data:
                if saved:
                    self._save_checkpoint(epoch_idx, verbose=verbose)
                continue
            if (epoch_idx + 1) % self.eval_step == 0:
                v0 = time()
                valid_score, valid_result = self._valid_epoch(valid_data, show_progress=show_progress)
                self.best_valid_score, self.cur_step, stop_flag, update_flag = early_stopping(
                    valid_score, self.best_valid_score, self.cur_step, max_step=self.stopping_step,
                    bigger=self.valid_metric_bigger)
                if verbose:
                    self.logger.info("epoch %d evaluating [time: %.2fs, valid_score: %f]" % (epoch_idx, time() - v0, valid_score))
                    self.logger.info("valid result: \n" + dict2str(valid_result))
                if update_flag:
                    if saved:
                        self._save_checkpoint(epoch_idx, verbose=verbose)
                    self.best_valid_result = valid_result
                if callback_fn:
                    callback_fn(epoch_idx, valid_score)
                if stop_flag:
                    if verbose:
                        self.logger.info("Finished training, best eval result in epoch %d"
                                         % (epoch_idx - self.cur_step * self.eval_step))
                    break
        return self.best_valid_score, self.best_valid_result

    # ---------------------------------------------------------------------------------------------- evaluation
    def _image_store(self, loader=None):
        if getattr(self, "_images", None) is None:
            from ..data.images import ImageStore

            dl = loader.dataset.dataload if hasattr(loader.dataset, "dataload") else loader.batcher.dataload
            self._images = ImageStore.from_config(self.config, dl, self.device)
        return self._images

    @torch.no_grad()
    def compute_item_feature(self, config, data, loader=None):
        """trainer.py:339-358: IDNet = the table itself; PixelNet = encode every item image in batches of 100."""
        if self.use_modality:
            store = self._image_store(loader)
            feats = []
            for s in range(0, store.n, 100):
                ids = torch.arange(s, min(store.n, s + 100), device=self.device)
                feats.append(self.model.module.compute_item(store.batch(ids)))   # id 0 -> zero image (batchset.py:58-60)
            self.item_feature = torch.cat(feats)
        else:
            self.item_feature = self.model.module.compute_item_all()
        # the fused scoring's main pass reads the item vectors as pre-split planes: made once per evaluation (0.4 ms for
        # 400 K x 512), reused by every batch of users
        self._item_planes = self._item_norm_max = None
        use_p = self.config["eval_planes"]
        if self.fused_topk and (use_p is None or bool(use_p)) and ops.score_planes_supported(self.item_feature.data):
            buf = getattr(self, "_item_planes_buf", None)
            N_, D_ = self.item_feature.shape
            if buf is None or (buf.rows, buf.cols) != (N_, D_) or buf.buf.device != self.item_feature.device:
                buf = self._item_planes_buf = ops.Planes.alloc(N_, D_, self.item_feature.device)
            feat = self.item_feature.data.contiguous()
            self._item_planes = ops.split_planes(feat, buf)
            # ... and the largest row norm: the margin of the threshold pass on 3 of the 6 bf16 products (ops.score_topk)
            self._item_norm_max = ops.row_norm_max(feat)

    @torch.no_grad()
    def _full_sort_batch_eval(self, batched_data):
        """Reference-literal path (trainer.py:327-337): full scores, then the two -inf masks."""
        user, history_index, positive_u, positive_i = batched_data
        scores = self.model.module.predict(user.to(self.device), self.item_feature)
        scores = scores.view(-1, self.tot_item_num)
        scores[:, 0] = -np.inf
        if history_index is not None:
            hu, hi = history_index
            scores[(hu.to(self.device), hi.to(self.device))] = -np.inf
        return scores, positive_u, positive_i

    def _eval_prefetch(self, loader, depth=4):
        """Iterate `loader` on a background thread (numpy / torch host work releases the GIL), packing each batch for one copy."""
        q = queue.Queue(maxsize=depth)

        def run():
            try:
                for b in loader:
                    q.put(self._pack_eval_batch(b))
            except BaseException as e:      # surfaced in the consumer
                q.put(e)
            q.put(None)

        threading.Thread(target=run, daemon=True).start()
        while True:
            b = q.get()
            if b is None:
                return
            if isinstance(b, BaseException):
                raise b
            yield b

    @staticmethod
    def _pack_eval_batch(batched_data, pin=True):
        """(item_seq, (history_u, history_i), positive_u, item_target) of seq_eval_collate -> ONE pinned int64 host buffer
        (sequences | history items | CSR row pointers | targets) + what is needed to cut it up again: one asynchronous copy per
        batch instead of four pageable .to(device) calls, each a host synchronisation (a 200 000-user evaluation: 6.5 -> 1.x s,
        tools/diag/trainer_throughput.py).  Batches of another shape are passed through."""
        user, history_index, positive_u, positive_i = batched_data
        if not (isinstance(user, torch.Tensor) and user.dtype == torch.int64 and not user.is_cuda and positive_i.dtype == torch.int64
                and (history_index is None or history_index[1].dtype == torch.int64)):
            return batched_data
        B0 = user.shape[0]
        parts, n_h = [user.reshape(-1)], None
        if history_index is not None:
            counts = torch.bincount(history_index[0], minlength=B0)        # (grouped by user in batch order: collate_fn.py:27-28)
            hp = torch.zeros(B0 + 1, dtype=torch.int64)
            hp[1:] = torch.cumsum(counts, 0)
            parts += [history_index[1].reshape(-1), hp]
            n_h = int(history_index[1].numel())
        parts.append(positive_i.reshape(-1))
        pk = torch.cat(parts)
        return ("packed", pk.pin_memory() if pin else pk, tuple(user.shape), n_h, int(positive_i.numel()))

    @torch.no_grad()
    def _full_sort_batch_topk(self, batched_data):
        """Fused path: encoder -> pxr_score_topk_f32 (scores never reach HBM)."""
        m = self.model.module
        from ..optim import has_item_table
        ptr = items = None
        if isinstance(batched_data[0], str):          # packed by _pack_eval_batch (on the prefetch thread)
            _, pk, ushape, n_h, n_p = batched_data
            pk = pk.to(self.device, non_blocking=True)
            n_u = int(torch.Size(ushape).numel())
            user_d = pk[:n_u].view(ushape)
            if n_h is not None:
                items = pk[n_u:n_u + n_h]
                ptr = pk[n_u + n_h:n_u + n_h + ushape[0] + 1].to(torch.int32)
            positive_i = pk[pk.numel() - n_p:]
        else:
            user, history_index, positive_u, positive_i = batched_data
            user_d = user.to(self.device)
            if history_index is not None:
                ptr, items = ops.history_csr(history_index[0], history_index[1], user.shape[0], self.device)
        # models whose item vectors come from an encoder (MOSASRec, FSASRec) read the sequence's rows from item_feature
        out, last = (m.encode_last(user_d, self.item_feature) if (self.use_modality or not has_item_table(m))
                     else m.encode_last(user_d))
        B, L, D = out.shape
        idx, _ = ops.score_topk(last, L * D, B, self.item_feature.data, max(self.config["topk"]), ptr, items,
                                table_planes=getattr(self, "_item_planes", None), table_norm_max=getattr(self, "_item_norm_max", None))
        return idx, positive_i

    def distributed_concat(self, tensor, num_total_examples):
        if self.world > 1:
            outs = [tensor.clone() for _ in range(self.world)]
            torch.distributed.all_gather(outs, tensor)
            tensor = torch.cat(outs, dim=0)
        return tensor.sum() / num_total_examples

    @torch.no_grad()
    def evaluate(self, eval_data, load_best_model=True, model_file=None, show_progress=False):
        if not eval_data:
            return
        if load_best_model:
            checkpoint_file = model_file or self.saved_model_file
            checkpoint = torch.load(checkpoint_file, map_location="cpu", weights_only=False)
            self.model.module.load_state_dict(checkpoint["state_dict"])
            self.logger.info("Loading model structure and parameters from {}".format(checkpoint_file))
        self.model.eval()
        self.tot_item_num = eval_data.dataset.dataload.item_num
        self.compute_item_feature(self.config, eval_data.dataset.dataload, eval_data)
        self.eval_collector._pending.clear()             # (blocks an aborted evaluation may have left behind)
        batches = eval_data
        if self.fused_topk and self.device.type == "cuda":
            batches = self._eval_prefetch(eval_data)      # batch construction + packing on a background thread
        for batched_data in batches:
            if self.fused_topk:
                idx, positive_i = self._full_sort_batch_topk(batched_data)
                self.eval_collector.eval_topk_collect(idx, positive_i)
            else:
                scores, positive_u, positive_i = self._full_sort_batch_eval(batched_data)
                self.eval_collector.eval_batch_collect(scores, positive_u.to(self.device), positive_i.to(self.device))
        ops.raise_on_bad_indices(self.device)
        num_total_examples = len(eval_data.sampler.dataset)
        struct = self.eval_collector.get_data_struct()
        result = self.evaluator.evaluate(struct)
        places = 5 if self.config["metric_decimal_place"] is None else self.config["metric_decimal_place"]
        for k, v in result.items():
            r = self.distributed_concat(torch.tensor([v], dtype=torch.float64).to(self.device), num_total_examples).cpu()
            result[k] = round(r.item(), places)
        return result
