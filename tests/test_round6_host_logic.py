"""Round-6 host logic that needs no GPU: the decay_check_name optimizer is built (reference trainer.py:73-91), its group split and
checkpoint layout follow torch.optim.AdamW's, the ops.StatusPoll / planes housekeeping bookkeeping."""
import torch


def test_trainer_builds_the_fragment_optimizer_instead_of_raising():
    from pixelrec_amd.optim import FragmentAdamW
    from pixelrec_amd.trainer.trainer import Trainer

    class Cfg(dict):
        def __getitem__(self, k):
            return self.get(k)

    t = Trainer.__new__(Trainer)
    t.config = Cfg(decay_check_name="LayerNorm")
    t.optim_args = {"modal_lr": 3e-3, "modal_decay": 0.02, "rec_lr": 1e-3, "rec_decay": 0.1}

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.module = torch.nn.Linear(4, 4)

    t.model = M()
    opt = t._build_optimizer()
    assert isinstance(opt, FragmentAdamW) and opt.fragment == "LayerNorm"


def test_fragment_groups_and_checkpoint_layout_follow_torch_adamw():
    """Group membership and the numbering of the per-parameter state are torch.optim.AdamW's for the reference's two fragment groups."""
    from pixelrec_amd.optim import FragmentAdamW

    class Wrapped(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.module = torch.nn.Sequential()
            self.module.add_module("dense", torch.nn.Linear(4, 4))
            self.module.add_module("LayerNorm", torch.nn.LayerNorm(4))

    dp = Wrapped()
    opt = FragmentAdamW(dp, "LayerNorm", 3e-3, 0.02, 1e-3, 0.1)
    named = [(n, p) for n, p in dp.named_parameters()]
    g0 = [p for n, p in named if "LayerNorm" in n]
    g1 = [p for n, p in named if "LayerNorm" not in n]
    ref = torch.optim.AdamW([{"params": g0, "lr": 3e-3, "weight_decay": 0.02}, {"params": g1, "lr": 1e-3, "weight_decay": 0.1}])
    mine, theirs = opt.state_dict(), ref.state_dict()
    assert [g["params"] for g in mine["param_groups"]] == [g["params"] for g in theirs["param_groups"]]
    for a, b in zip(mine["param_groups"], theirs["param_groups"]):
        assert (a["lr"], a["weight_decay"], tuple(a["betas"]), a["eps"]) == (b["lr"], b["weight_decay"], tuple(b["betas"]), b["eps"])
    assert opt.groups() == (["module.LayerNorm.weight", "module.LayerNorm.bias"], ["module.dense.weight", "module.dense.bias"])


def test_trainer_falls_back_to_exact_scales_when_a_stale_scale_overflows():
    """ops.H2StaleOverflow (status bit 128) is not fatal for a training run: the Trainer says so, turns the stale scales off and drops
    its captured step (the next full batch re-captures on exact per-step scales)."""
    import logging

    from pixelrec_amd import ops
    from pixelrec_amd.trainer.trainer import Trainer

    class Inner(torch.nn.Module):
        h2_stale_scales = True

    class Wrapped(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.module = Inner()

    t = Trainer.__new__(Trainer)
    t.model, t._gstep, t.world, t.use_graph = Wrapped(), object(), 1, True
    records = []

    class H(logging.Handler):
        def emit(self, r):
            records.append(r.getMessage())

    t.logger = logging.getLogger("pxr-test-stale")
    t.logger.addHandler(H())
    t._steps_done = 1000
    t._h2_stale_fallback(ops.H2StaleOverflow("a gradient exceeded its headroom"))
    assert t.model.module.h2_stale_scales is False and t._gstep is None and t.use_graph is True
    # one rank: exact scales for a while, then back (re-seeded, re-captured); the wait grows x 4 with every further overflow
    assert t._stale_resume_at == 1000 + Trainer.H2_STALE_BACKOFF[0] and "for the next 256 steps" in records[0]
    t._gstep, t._steps_done = "exact-scale graph", 1000 + Trainer.H2_STALE_BACKOFF[0] - 1
    t._h2_stale_resume()
    assert t.model.module.h2_stale_scales is False and t._gstep == "exact-scale graph"          # not yet
    t._steps_done += 1
    t._h2_stale_resume()
    assert t.model.module.h2_stale_scales is True and t._gstep is None and t._stale_resume_at is None
    t._h2_stale_fallback(ops.H2StaleOverflow("again, later"))
    assert t._stale_resume_at == t._steps_done + 4 * Trainer.H2_STALE_BACKOFF[0]
    del records[1:]
    t._stale_resume_at = None
    t.model.module.h2_stale_scales, t.world = True, 4        # data parallel: no re-capture (its dry step would issue unmatched collectives)
    t._h2_stale_fallback(ops.H2StaleOverflow("on one rank of four"))
    assert t.use_graph is False and len(records) == 2
    records.pop()
    assert records and "exact per-step scales" in records[0]
    t._gstep = "kept"
    t._h2_stale_fallback(ops.H2StaleOverflow("again"))          # already off: nothing to do, nothing dropped
    assert t._gstep == "kept" and len(records) == 1
    assert issubclass(ops.H2StaleOverflow, RuntimeError)


def test_trainer_pins_the_host_thread_pool_unless_the_caller_chose(monkeypatch):
    """A Trainer built outside main.py (which exports OMP_NUM_THREADS=1 like the reference's launcher, main.py:5) limits torch's
    intra-op pool to one thread: with one thread per core of a 256-core host every small torch.cat of the batcher woke the whole team
    (B = 512 batches: 32 ms instead of 3).  OMP_NUM_THREADS or `host_threads` in the YAML override it."""
    from pixelrec_amd.trainer.trainer import host_threads_for

    class Cfg(dict):
        def __getitem__(self, k):
            return self.get(k)

    monkeypatch.delenv("OMP_NUM_THREADS", raising=False)
    assert host_threads_for(Cfg()) == 1 and host_threads_for(Cfg(host_threads=4)) == 4 and host_threads_for(Cfg(host_threads=0)) == 0
    monkeypatch.setenv("OMP_NUM_THREADS", "8")
    assert host_threads_for(Cfg()) == 0 and host_threads_for(Cfg(host_threads=2)) == 2
