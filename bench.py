#!/usr/bin/env python
"""bench.py -- user-sequences/sec of the SASRec (IDNet) training step at emb=512, seq_len=50 (BASELINE.json).

One "step" = zero_grad -> forward -> backward -> gradient exchange -> AdamW over ALL parameters incl. the whole
400 001 x 512 item table (the reference's trainer.py:116-125 loop body), on one synthetic batch of B sequences per
GPU that is already resident in HBM.  Workload = BASELINE.json configs[1]: SASRec IDNet, N=400 001 items, D=512,
L=50, 4 heads, inner 2x, 2 layers, fp32, dropout 0.1, AdamW lr 1e-4 wd 0.1, B=64 per GPU (the reference's
train_batch_size, overall/ID.yaml:19).  Synthetic ids: Zipf(1.0) positives through a fixed permutation, uniform
negatives, ~30 % left-padded sequences.

  python bench.py [--gpus N --steps K --warmup W]          # N>1 is launched by torch.distributed.run

Prints ONE JSON line on rank 0 (see the task contract): value = whole-job sequences/s; `roofline` = the dominant
kernel of the step -- the fp32-MFMA GEMM kernel behind every nn.Linear forward/backward (22 launches, ~60 % of the
step's GPU time; MFMA-bound) -- `cpu_baseline` = the CPU oracle (plain PyTorch fp32 restatement of the reference step)
timed on this host, plus extra roofline objects for the north-star targets (embedding gather GB/s, full-catalog
scoring GEMM MFMA utilisation) and for the dense AdamW table sweep that the default lazy optimizer avoids.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

NS = dict(n_items=400_001, D=512, L=50, H=4, inner=2, n_layers=2)
HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TF = 157.3    # MI355X_MICROARCH.md: fp32-input MFMA = fp32 vector peak


def model_config(p_drop=0.1):
    return {"n_layers": NS["n_layers"], "n_heads": NS["H"], "embedding_size": NS["D"], "inner_size": NS["inner"],
            "hidden_dropout_prob": p_drop, "attn_dropout_prob": p_drop, "hidden_act": "gelu",
            "layer_norm_eps": 1e-12, "initializer_range": 0.02, "MAX_ITEM_LIST_LENGTH": NS["L"], "seed": 2020}


def ev_pair():
    return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def time_kernel(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = ev_pair()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def cpu_baseline(batch, budget_s=20.0):
    """The CPU oracle (checker, never the product) timed on this host's cores on a bounded sample."""
    from oracle import sasrec_oracle as O
    from pixelrec_amd import synth

    # all cores up to 32 threads: beyond that the 3.3 GB dense optimizer sweep and the intra-op thread pool stop
    # scaling (measured on the 256-core GPU host: 256 threads ran 60x SLOWER than 8 threads on the dev container)
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cores = torch.get_num_threads()
    params = O.synth_params(NS["n_items"], NS["D"], NS["L"], NS["n_layers"], NS["inner"], seed=1, perturb=False)
    cfg = {"n_layers": NS["n_layers"], "n_heads": NS["H"], "layer_norm_eps": 1e-12}
    tr = O.OracleTrainer(params, cfg, lr=1e-4, weight_decay=0.1)
    rng = np.random.default_rng(7)
    zipf = synth.ZipfItems(NS["n_items"], seed=2020)
    items, mask = synth.train_batch(NS["n_items"], batch, NS["L"], rng, zipf)
    items, mask = torch.from_numpy(items), torch.from_numpy(mask)
    tr.step(items, mask)  # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    steps = 0
    while steps < 10 and (time.perf_counter() - t0) < budget_s:
        tr.step(items, mask)
        steps += 1
    dt = time.perf_counter() - t0
    # the reference launcher pins OMP_NUM_THREADS=1 (main.py:5): one step at one thread, for the record
    torch.set_num_threads(1)
    t1 = time.perf_counter()
    tr.step(items, mask)
    dt1 = time.perf_counter() - t1
    torch.set_num_threads(cores)
    return {"value": batch * steps / dt, "unit": "sequences/s", "cores": cores, "kind": "port",
            "sample": f"{steps} full training steps (fwd+bwd+dense AdamW, dropout off) of B={batch} at the same "
                      f"N=400001/D=512/L=50 config, torch {torch.__version__} CPU fp32, {dt / max(steps, 1):.2f} s/step",
            "one_thread": {"value": batch / dt1, "unit": "sequences/s", "cores": 1,
                           "sample": f"1 step, {dt1:.2f} s (the reference's own launcher setting, OMP_NUM_THREADS=1)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=64, help="sequences per GPU per step (reference: 64)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the gather / scoring roofline micro-runs")
    ap.add_argument("--no-overlap", action="store_true", help="weight gradients on the main stream (A/B knob)")
    ap.add_argument("--no-group", action="store_true", help="per-layer weight-gradient GEMMs instead of one grouped launch")
    ap.add_argument("--no-graph", action="store_true", help="issue the step eagerly instead of replaying a hipGraph")
    ap.add_argument("--no-gemm-events", action="store_true", help="do not bracket GEMM launches with HIP events")
    ap.add_argument("--clock-ramp-s", type=float, default=0.3, help="seconds of extra untimed steps before the warm-up")
    ap.add_argument("--emb", type=int, default=NS["D"], help="embedding size (default: the north-star 512)")
    ap.add_argument("--heads", type=int, default=NS["H"])
    ap.add_argument("--items", type=int, default=NS["n_items"], help="catalogue size incl. the padding id")
    ap.add_argument("--seq-len", type=int, default=NS["L"])
    ap.add_argument("--table-sharding", action="store_true",
                    help="row-shard the item table over the ranks (pixelrec_amd/model/sharded.py) instead of replicating it")
    ap.add_argument("--force-collectives", action="store_true",
                    help="1-GPU validation knob: create a 1-rank RCCL group and run every gradient collective anyway")
    ap.add_argument("--graph-collectives", action="store_true",
                    help="capture the RCCL collectives inside the step hipGraph too (opt-in for world > 1)")
    args = ap.parse_args()
    custom = (args.emb, args.heads, args.items, args.seq_len) != (NS["D"], NS["H"], NS["n_items"], NS["L"]) or args.table_sharding
    NS.update(D=args.emb, H=args.heads, n_items=args.items, L=args.seq_len)   # other BASELINE configs on request

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", device_id=dev, rank=rank, world_size=world)

    from pixelrec_amd import ops, synth
    from pixelrec_amd.model import SASRec
    from pixelrec_amd.optim import PxrAdamW
    from pixelrec_amd.parallel import DataParallel

    torch.manual_seed(2020)

    class DL:
        item_num = NS["n_items"]

    B, L, D, N = args.batch, NS["L"], NS["D"], NS["n_items"]
    if args.table_sharding:
        from pixelrec_amd.model import ShardedDataParallel, ShardedSASRec as SASRec   # noqa: F811
    with torch.device(dev):
        model = SASRec(model_config(0.1), DL())  # random-init weights of the reference architecture
    model.train()
    if args.no_group:
        model.group_weight_grads = False
        model.overlap_weight_grads = not args.no_overlap
    dp = (ShardedDataParallel(model, force_collectives=args.force_collectives) if args.table_sharding
          else DataParallel(model, force_collectives=args.force_collectives))
    opt = PxrAdamW(model, lr=1e-4, weight_decay=0.1)

    # synthetic batches, rank-distinct, resident in HBM before the timed region
    rng = np.random.default_rng(2020 + 1000 * rank)
    zipf = synth.ZipfItems(N, seed=2020)
    pool = []
    for _ in range(8):
        it, mk = synth.train_batch(N, B, L, rng, zipf)
        pool.append((torch.from_numpy(it).to(dev), torch.from_numpy(mk).to(dev)))

    gemm_events = []
    # replaying RCCL collectives from a hipGraph is opt-in: it cannot be validated on the 1-GPU development box
    use_graph = (not args.no_graph) and ((world == 1 and not args.force_collectives) or args.graph_collectives)
    use_graph = use_graph and not args.table_sharding      # the sharded forward issues collectives: eager only
    gstep = None
    if use_graph:
        from pixelrec_amd.graph import GraphedTrainStep

        gstep = GraphedTrainStep(dp, opt, *pool[0])

    one = torch.ones((), dtype=torch.float32, device=dev)

    def step(i, record=False):
        if gstep is not None:
            return gstep(*pool[i % len(pool)])
        opt.zero_grad()
        loss = dp(pool[i % len(pool)])
        loss.backward(one)       # preallocated d(loss)/d(loss): no ones_like fill per step
        dp.sync_gradients(defer_flat=True)
        opt.step()
        return loss

    def instrumented_step(i):
        """Same step, eager, with HIP events around every GEMM launch (run AFTER the timed region)."""
        ops.GEMM_TIMING = gemm_events
        opt.zero_grad()
        loss = dp(pool[i % len(pool)])
        loss.backward(one)       # preallocated d(loss)/d(loss): no ones_like fill per step
        dp.sync_gradients(defer_flat=True)
        opt.step()
        ops.GEMM_TIMING = None
        return loss

    # clock ramp: a cold process starts with the GPU at its idle clocks (sclk ~600 MHz); ~0.3 s of the same work before
    # the W counted warm-up steps keeps short runs from timing the governor instead of the step (untimed, like warm-up)
    t_ramp, i_ramp = time.perf_counter(), 0
    while args.clock_ramp_s > 0:
        for _ in range(8):
            step(i_ramp)
            i_ramp += 1
        torch.cuda.synchronize()
        if time.perf_counter() - t_ramp >= args.clock_ramp_s:
            break
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i, record=True)
    t_enqueued = time.perf_counter() - t0   # host time to issue the K steps (== total when host-bound)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach())
    n_inst = 0
    if not args.no_gemm_events:
        n_inst = min(args.steps, 20)
        for i in range(n_inst):
            instrumented_step(i)
        torch.cuda.synchronize()

    if rank != 0:
        if dist.is_initialized():
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel: the fp32-MFMA GEMM (all launches of the timed steps, HIP events) ------
    g_s = sum(s.elapsed_time(e) for s, e, _, _ in gemm_events) * 1e-3 or float("nan")
    g_fl = sum(f for _, _, f, _ in gemm_events)
    n_launch = len(gemm_events)
    per_kernel = {}
    for s_, e_, f_, tag in gemm_events:     # one entry per kernel instantiation, comparable with rocprofv3's rows
        k = per_kernel.setdefault(tag, [0, 0.0, 0.0])
        k[0] += 1; k[1] += s_.elapsed_time(e_) * 1e-3; k[2] += f_
    per_kernel = {t: {"launches_per_step": c / max(n_inst, 1), "avg_kernel_us": sec / c * 1e6,
                      "tflops": fl / sec / 1e12} for t, (c, sec, fl) in per_kernel.items()}
    roof = {"bound": "mfma", "kernel": "gemm_kernel (v_mfma_f32_32x32x2_f32; every nn.Linear fwd/bwd of the step)",
            "achieved": g_fl / g_s / 1e12, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
            "frac": g_fl / g_s / 1e12 / MFMA_F32_PEAK_TF, "traffic": None,
            "launches_per_step": n_launch / max(n_inst, 1), "avg_kernel_us": g_s / max(n_launch, 1) * 1e6,
            "algorithmic_flops_per_step": g_fl / max(n_inst, 1),
            "gemm_time_per_step_us": g_s / max(n_inst, 1) * 1e6, "kernels": per_kernel,
            "note": f"HIP events around every GEMM launch of {n_inst} extra eager steps issued right after the timed "
                    "region (bracketing launches inside it would make the step host-bound)"}

    out = {
        "metric": (f"user-sequences/sec at emb={D} seq_len={L} (SASRec IDNet training step: fwd+bwd+AdamW)" if custom else
                   "user-sequences/sec at emb=512 seq_len=50 (SASRec IDNet training step: fwd+bwd+AdamW)"),
        "value": world * B * args.steps / dt, "unit": "sequences/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": (f"custom shape (NOT the headline config): SASRec IDNet emb={D} seq_len={L}, {N} items, "
                                f"{NS['H']} heads, inner 2x, 2 layers" if custom else
                                "BASELINE.json configs[1]: SASRec IDNet emb=512 seq_len=50, 400001 items, 4 heads, "
                                "inner 2x, 2 layers, BPR loss vs 1 sampled negative, dropout 0.1, AdamW lr 1e-4 wd 0.1"),
                   "batch_per_gpu": B, "global_batch": world * B, "seq_len": L, "n_items": N, "embedding_size": D,
                   "parallelism": f"dp{world}" + ("+row-sharded-table" if args.table_sharding else ""),
                   "hip_graph": bool(use_graph)},
        "final_loss": final_loss, "host_enqueue_ms_per_step": t_enqueued / args.steps * 1e3,
        "roofline": roof,
    }

    if not args.no_extras and world == 1 and B == 64 and not custom:
        # (0) the same step at throughput-oriented batch sizes (SURVEY.md §8d asks for B=64 AND 512 / 2048 per GPU)
        from pixelrec_amd.graph import GraphedTrainStep as _G

        out["throughput_batches"] = []
        for Bt in (512, 2048):
            bt = [tuple(torch.from_numpy(a).to(dev) for a in synth.train_batch(N, Bt, L, rng, zipf)) for _ in range(4)]
            g2 = _G(dp, opt, *bt[0]) if use_graph else None

            def step_b(i):
                if g2 is not None:
                    return g2(*bt[i % 4])
                opt.zero_grad()
                dp(bt[i % 4]).backward(one)
                dp.sync_gradients(defer_flat=True)
                opt.step()

            for i in range(3):
                step_b(i)
            torch.cuda.synchronize()
            tb = time.perf_counter()
            n_b = 20 if Bt == 512 else 8
            for i in range(n_b):
                step_b(i)
            torch.cuda.synchronize()
            tb = (time.perf_counter() - tb) / n_b
            out["throughput_batches"].append({"batch_per_gpu": Bt, "value": Bt / tb, "unit": "sequences/s",
                                              "ms_per_step": tb * 1e3, "steps": n_b})
            del g2, bt

    if not args.no_extras and world == 1 and not custom:
        # (0b) what this box's HBM delivers on a plain device copy (quote the measured peak next to the
        # datasheet one): 1 GiB read + 1 GiB written per launch
        src = torch.empty(1 << 28, dtype=torch.float32, device=dev).normal_()
        dst = torch.empty_like(src)
        t_c = time_kernel(lambda: dst.copy_(src), iters=10)
        out["hbm_stream_copy"] = {"achieved": 2.0 * src.numel() * 4 / t_c / 1e9, "unit": "GB/s", "peak": HBM_PEAK_GBS,
                                  "frac": 2.0 * src.numel() * 4 / t_c / 1e9 / HBM_PEAK_GBS,
                                  "note": "torch D2D copy of 1 GiB (read + write counted)"}
        del src, dst
        # (1) embedding gather, north-star HBM target: uniform ids (worst case for caches), B=2048-equivalent rows
        n_rows = 2048 * 2 * (L + 1)
        idx = torch.randint(1, N, (n_rows,), device=dev)
        table = model.item_embedding.weight.data
        t_g = time_kernel(lambda: ops.embed_gather(table, idx))
        gb = 2.0 * n_rows * D * 4
        out["roofline_gather"] = {"bound": "hbm", "kernel": "embed_gather_kernel", "achieved": gb / t_g / 1e9,
                                  "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / t_g / 1e9 / HBM_PEAK_GBS,
                                  "rows": n_rows, "avg_kernel_us": t_g * 1e6,
                                  "note": "417792 B/sequence x 2048 sequences, uniform random ids"}
        # (2) full-catalog scoring GEMM [1024,512] x [512,400001] on the fp32 MFMA
        model.eval()
        seq = torch.from_numpy(synth.eval_batch(N, 1024, L, np.random.default_rng(3), zipf)[0]).to(dev)
        _, last = model.encode_last(seq)
        scores = torch.empty(1024, N, dtype=torch.float32, device=dev)
        t_s = time_kernel(lambda: ops.gemm(True, True, 1024, N, D, last, L * D, table, D, scores, N, ops.EPI_NONE,
                                           use_ws=False), iters=10)
        fl = 2.0 * 1024 * N * D
        out["roofline_scoring"] = {"bound": "mfma", "kernel": "gemm_kernel<128,128,KC,KC>", "achieved": fl / t_s / 1e12,
                                   "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s", "frac": fl / t_s / 1e12 / MFMA_F32_PEAK_TF,
                                   "avg_kernel_us": t_s * 1e6, "note": "409.6 MFLOP/user x 1024 users, exact fp32 MFMA"}
        del scores
        # (3) the dense AdamW table sweep (what `table_update="dense"` runs every step; the default lazy optimizer
        # replays untouched rows on demand instead): pure HBM streaming of p, m, v
        tm, tv = torch.zeros_like(table), torch.zeros_like(table)
        slot = torch.full((N,), -1, dtype=torch.int32, device=dev)
        tcopy = table.clone()
        t_a = time_kernel(lambda: ops.adamw_table(tcopy, tm, tv, slot, None, 1e-4, 0.9, 0.999, 1e-8, 0.1, 1), iters=10)
        ab = 24.0 * N * D + 4.0 * N
        out["roofline_adamw_dense_sweep"] = {"bound": "hbm", "kernel": "adamw_table_kernel", "achieved": ab / t_a / 1e9,
                                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ab / t_a / 1e9 / HBM_PEAK_GBS,
                                             "avg_kernel_us": t_a * 1e6,
                                             "note": "24 B x N x D (read+write p,m,v) + 4 B x N slot map per launch"}
        del tm, tv, slot, tcopy

    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(B)
    elif world == 1:
        out["cpu_baseline"] = None

    # RCCL prints its version banner through C stdio (NCCL_DEBUG=VERSION is exported on the GPU boxes) and that buffer
    # is only flushed at exit: flush it now so the JSON line is the LAST line on stdout
    import ctypes
    ctypes.CDLL(None).fflush(None)
    print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
