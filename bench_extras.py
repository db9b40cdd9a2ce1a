"""bench_extras.py -- everything `python bench.py` measures BESIDE the contract's timed region (run after it, on rank 0 of a 1-GPU run,
skipped with --no-extras): the same step at throughput-oriented batch sizes on both arithmetics, the HBM copy rate of the box, the
north-star kernel rooflines (embedding gather standalone and as run, full-catalogue scoring literal / fused / default), the dense AdamW
sweep, what the lazy table schedule costs against it, the PixelNet step (BASELINE configs[2]) and the review targets.  All of it lands in
the full record (the side file bench.py names in `extras`); the one-line stdout record only carries a digest (bench.compact_line).

`run(ctx, out)`: ctx = the namespace bench.main() hands over (model, optimizer state, batch stream, step closures, flags)."""
import argparse
import os
import time

import numpy as np
import torch

from bench import (HBM_PEAK_GBS, MFMA_BF16_PEAK_TF, OPERANDS_B3, OPERANDS_H2, _env, executed_products, mfma_roof, pixelnet_run,
                   time_kernel)


def run(ctx, out):
    args, world, custom, dev = ctx.args, ctx.world, ctx.custom, ctx.dev
    B, L, D, N = ctx.B, ctx.L, ctx.D, ctx.N
    model, dp, opt, rng, zipf = ctx.model, ctx.dp, ctx.state.opt, ctx.rng, ctx.zipf
    use_graph, b3, dt, t_eager = ctx.use_graph, ctx.b3, ctx.dt, ctx.t_eager
    eager_step, next_batch = ctx.eager_step, ctx.next_batch
    from pixelrec_amd import ops, synth
    from pixelrec_amd.optim import PxrAdamW

    if not args.no_extras and world == 1 and B == 64 and not custom:
        # (0) the same step at throughput-oriented batch sizes (SURVEY.md §8d asks for B=64 AND 512 / 2048 per GPU)
        from pixelrec_amd.graph import GraphedTrainStep as _G

        out["throughput_batches"] = []
        # every batch size on BOTH arithmetics: the library's default (fp16 two-plane operands) and the
        # six-product bf16x3 operands (PXR_SEQ_H2=0)
        for Bt, h2_env in ((512, "auto"), (512, "0"), (2048, "auto"), (2048, "0")):
          with _env(PXR_SEQ_H2=h2_env):
              n_bt = 26 if Bt == 512 else 12     # distinct batches: warm-up + timed + instrumented steps never repeat one
              bt = [tuple(torch.from_numpy(a).to(dev) for a in synth.train_batch(N, Bt, L, rng, zipf)) for _ in range(n_bt)]
              g2 = _G(dp, opt, *bt[0], lookahead=not (not args.lookahead)) if use_graph else None
              cur = [0]

              def step_b(eager=False):
                  b = bt[cur[0] % n_bt]
                  cur[0] += 1
                  nxt = None if (not args.lookahead) else bt[cur[0] % n_bt][0]
                  if g2 is not None and not eager:
                      return g2(*b, next_items=nxt)
                  return eager_step(b, nxt)

              for i in range(3):
                  step_b()
              torch.cuda.synchronize()
              tb = time.perf_counter()
              n_b = 20 if Bt == 512 else 8
              for i in range(n_b):
                  step_b()
              torch.cuda.synchronize()
              tb = (time.perf_counter() - tb) / n_b
              out["throughput_batches"].append({"batch_per_gpu": Bt, "value": Bt / tb, "unit": "sequences/s",
                                                "ms_per_step": tb * 1e3, "steps": n_b, "PXR_SEQ_H2": h2_env})
              # the fused gather (ln_fwd_kernel<GATHER>) where it runs, at this batch size
              evs = []
              ops.GEMM_TIMING = evs
              for i in range(2):
                  step_b(eager=True)
              ops.GEMM_TIMING = None
              torch.cuda.synchronize()
              fam = {}
              for s_, e_, w, t in evs:
                  key = ("gemm fwd / dX (gemm_p3_kernel)" if t.startswith("gemm") else "grouped weight gradients" if t.startswith("grouped_dw")
                         else "attention" if t.startswith("attn") else "layernorm (+ fused gather)" if t.startswith("ln_") else
                         "table optimizer rows (adamw_rows)" if t.startswith("adamw_rows") else "other instrumented (loss, sort, segsum, flat optimizer)")
                  fam[key] = fam.get(key, 0.0) + s_.elapsed_time(e_) * 1e3 / 2
              gfl = sum(w for _, _, w, t in evs if t.startswith("gemm") or t.startswith("grouped_dw")) / 2
              gex = sum(w * executed_products(t) for _, _, w, t in evs if t.startswith("gemm") or t.startswith("grouped_dw")) / 2
              on_h2 = any("HALF" in t for _, _, _, t in evs)
              gus = sum(v for k, v in fam.items() if k.startswith("gemm") or k.startswith("grouped"))
              fam["everything not bracketed with events (attention, the other LayerNorm sites, loss, id sort, segment sums, "
                  "flat optimizer, launch gaps) = step time - the rows above"] = tb * 1e6 - sum(fam.values())
              out["throughput_batches"][-1]["kernel_families_us_per_step"] = fam
              operands = (OPERANDS_H2 if on_h2 else OPERANDS_B3) if b3 else "f32_input_mfma"
              out["throughput_batches"][-1]["operands"] = operands
              out["throughput_batches"][-1]["gemm_family"] = {**mfma_roof(gfl, gus * 1e-6, b3, gex if on_h2 else None), "us_per_step": gus,
                                                              "operands": operands}
              gl = [(s_.elapsed_time(e_) * 1e-3, w) for s_, e_, w, t in evs if t.startswith("ln_fwd_kernel<GATHER>")]
              if gl and "roofline_gather_fused" in out:
                  sec = sum(x for x, _ in gl) / len(gl)
                  out["roofline_gather_fused"].append({"bound": "hbm", "kernel": "ln_fwd_kernel<GATHER>", "batch_per_gpu": Bt,
                                                       "operands": operands,
                                                       "achieved": gl[0][1] / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                       "frac": gl[0][1] / sec / 1e9 / HBM_PEAK_GBS, "avg_kernel_us": sec * 1e6})
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
        # (1b) the gather AS THE STEP RUNS IT -- ln_fwd_kernel<GATHER>: rows + position rows -> LayerNorm -> dropout -> y, xhat and the
        # two fp16 planes of y -- timed alone, back to back (the per-batch entries above come from HIP events around eagerly issued
        # launches and carry the 4-5 us an event pair adds to a 15-200 us kernel)
        if hasattr(model, "_p") and b3 and getattr(model, "_planes_on", lambda: False)():
            alone = []
            for Bt in (64, 512, 2048):
                ids_t = torch.from_numpy(synth.train_batch(N, Bt, L, rng, zipf)[0][:, 0, :L].copy()).to(dev)
                fn = lambda: ops.input_ln_fwd(table, ids_t, L, Bt, L, model._p("pos"), model._p("ln0.w"), model._p("ln0.b"), 1e-12, 0.1,
                                              12345, 0, save=True, planes="h2")
                t_l = time_kernel(fn, iters=20)
                by = Bt * L * D * (4.0 * 3 + 4.0)
                alone.append({"batch_per_gpu": Bt, "avg_kernel_us": t_l * 1e6, "achieved": by / t_l / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": by / t_l / 1e9 / HBM_PEAK_GBS})
            # HBM / fabric bytes per launch by FETCH_SIZE / WRITE_SIZE passes of the same launches (committed file, NOT measured here)
            tf = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r06", "pmc", "gather_traffic_summary.json")
            if os.path.exists(tf) and not custom:
                import json as _json
                tr = _json.load(open(tf))
                for e_ in alone:
                    grid = (e_["batch_per_gpu"] * L + 3) // 4 * 256
                    hit = [v for k, v in tr.items() if "ln_fwd_kernel" in k and k.endswith(f"grid {grid}")]
                    e_["traffic"] = hit[0]["hbm_bytes_per_launch"] if hit else None
                    e_["algorithmic_bytes"] = e_["batch_per_gpu"] * L * D * 16.0
            out["roofline_gather_fused_alone"] = {"bound": "hbm", "kernel": "ln_fwd_kernel<GATHER> (h2 planes; Zipf ids of the bench stream)",
                                                  "batches": alone, "traffic_source": "profiles/r06/pmc/gather_traffic_summary.json",
                                                  "note": "bytes = B*L table rows read + y + xhat (fp32) + two fp16 planes of y written; at B = 512 "
                                                          "the 157 MB of outputs fit the 256 MB MALL, at B = 2048 (630 MB) they stream to HBM"}
        # (2) full-catalog scoring GEMM [1024,512] x [512,400001] on the fp32 MFMA
        model.eval()
        seq = torch.from_numpy(synth.eval_batch(N, 1024, L, np.random.default_rng(3), zipf)[0]).to(dev)
        _, last = model.encode_last(seq)
        scores = torch.empty(1024, N, dtype=torch.float32, device=dev)
        t_s = time_kernel(lambda: ops.gemm(True, True, 1024, N, D, last, L * D, table, D, scores, N, ops.EPI_NONE,
                                           use_ws=False), iters=10)
        fl = 2.0 * 1024 * N * D
        out["roofline_scoring_literal"] = {"bound": "mfma", "kernel": "gemm_b3_kernel<128,128,KC,KC>" if b3 else "gemm_kernel<128,128,KC,KC>",
                                           **mfma_roof(fl, t_s, b3), "avg_kernel_us": t_s * 1e6,
                                           "note": "the literal scoring product (scores written: 1.6 GB), fp32 operands split inside the main "
                                                   "loop; 409.6 MFLOP/user x 1024 users (algorithmic fp32 flops; gemm_mode " + ops.gemm_mode() + ")"}
        # the path the PRODUCT evaluates with (trainer.evaluate): table split once per evaluation, then per batch of 1024 users the
        # fused scoring + masks + top-10 whose main pass is score_thresh_p3_kernel on the planes; checked here against the literal
        # sequence (GEMM -> the two -inf masks -> torch.topk) on the same operands
        ev_b = synth.eval_batch(N, 1024, L, np.random.default_rng(3), zipf)
        hu_t, hi_t = torch.from_numpy(ev_b[1]), torch.from_numpy(ev_b[2])
        ptr, hitems = ops.history_csr(hu_t, hi_t, 1024, dev)
        last2 = torch.as_strided(last, (1024, D), (L * D, 1)).contiguous()
        scores[:, 0] = -float("inf")
        scores[(hu_t.to(dev), hi_t.to(dev))] = -float("inf")
        lit_v, lit_i = torch.topk(scores, 10, dim=-1)
        del scores
        if b3 and ops.score_planes_supported(table):
            t_sp = time_kernel(lambda: ops.split_planes(table), iters=3, warm=1)
            tpl = ops.split_planes(table)
            vmax = ops.row_norm_max(table)
            t_nm = time_kernel(lambda: ops.row_norm_max(table), iters=3, warm=1)
            # six-product schedule (round 3's) first: its ids / values are what the reduced-product default must reproduce bit for bit
            os.environ["PXR_TOPK_PRODUCTS"] = "6"
            f6_i, f6_v = ops.score_topk(last2, D, 1024, table, 10, ptr, hitems, table_planes=tpl, table_norm_max=vmax)
            t_f6 = time_kernel(lambda: ops.score_topk(last2, D, 1024, table, 10, ptr, hitems, table_planes=tpl, table_norm_max=vmax), iters=10)
            os.environ.pop("PXR_TOPK_PRODUCTS", None)
            f_i, f_v = ops.score_topk(last2, D, 1024, table, 10, ptr, hitems, table_planes=tpl, table_norm_max=vmax)
            same_ids = bool(torch.equal(f_i, lit_i))
            same_bits = bool(torch.equal(f_i, f6_i) and torch.equal(f_v, f6_v))
            t_f = time_kernel(lambda: ops.score_topk(last2, D, 1024, table, 10, ptr, hitems, table_planes=tpl, table_norm_max=vmax), iters=10)
            with ops.ScoreClock(dev) as sc3:
                for _ in range(12):
                    ops.score_topk(last2, D, 1024, table, 10, ptr, hitems, table_planes=tpl, table_norm_max=vmax)
            out["roofline_scoring"] = {"bound": "mfma", "kernel": "score_thresh_p3_kernel (main pass of pxr_score_topk_planes_f32 on the pre-split "
                                       "table, all six bf16 products; + sample pass score_topk_kernel, topk_tau, topk_cand_merge in the same call)",
                                       **mfma_roof(fl, t_f6, b3), "avg_call_us": t_f6 * 1e6, "identical_top10": bool(torch.equal(f6_i, lit_i)),
                                       "max_abs_value_diff_vs_literal": float((f_v - lit_v).abs().max()),
                                       "table_split_us_once_per_evaluation": t_sp * 1e6,
                                       "note": "the WHOLE fused scoring + history / padding masks + top-10 call per 1024 users x 400 001 items "
                                               "(scores never reach HBM), timed end to end and priced as if all of it were the scoring product; "
                                               "the main-pass kernel alone: profiles/r04 eval kernel stats"}
            # the clock the part SUSTAINS inside the default threshold kernel (power-limited under MFMA load with random operands): the
            # kernel records its own shader-clock cycles against the 100 MHz reference (ops.ScoreClock); peak of the bf16 / fp16 pipe at
            # that clock: 256 CUs x 4 SIMDs x 1024 flop/clk (v_mfma_f32_32x32x16: 32768 flop in 8 passes of 4 clk) -- 2.5 PFLOP/s at ~2.4 GHz
            score_ghz = sc3.ghz()
            peak_sus = 256 * 4 * 1024 * score_ghz * 1e9 / 1e12 if score_ghz > 0 else float("nan")
            ex3 = ops.topk_products() * fl / t_f / 1e12
            out["roofline_scoring_fused_topk"] = {
                "ms_per_1024_users": t_f * 1e3, "identical_top10": same_ids, "products_in_threshold_pass": ops.topk_products(),
                "identical_ids_and_values_to_six_product_schedule": same_bits, "six_product_schedule_ms": t_f6 * 1e3,
                "row_norm_max_us_once_per_evaluation": t_nm * 1e6, "speedup_vs_literal_gemm_alone": t_s / t_f,
                "sustained_clock_ghz": score_ghz, "peak_at_sustained_clock": peak_sus,
                "executed_tflops_whole_call": ex3, "frac": ex3 / MFMA_BF16_PEAK_TF, "frac_of_sustained_peak": ex3 / peak_sus,
                "sustained_clock_note": "shader-clock cycles / 100 MHz reference ticks recorded by workgroup 0 of score_thresh_fast_kernel "
                                        "itself over 12 calls (pxr_score_topk_clock_out); the six-product pass runs at 1.70-1.79 GHz in the "
                                        "lab (profiles/r06/lab)",
                "note": "the product's default (trainer.evaluate): threshold pass on 3 of the 6 bf16 products (score_thresh_fast_kernel, "
                        "256 x 256 tiles, one accumulator set), threshold lowered by a rigorous per-user bound, survivors that can reach the "
                        "top 10 re-scored with all six products in the full pass's MFMA order (topk_rescore_kernel): bit-identical output at "
                        "about half the MFMA work -- the executed products are fewer, so this entry is a time, not a roofline fraction"}
            del tpl
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

    if not args.no_extras and world == 1 and not custom and B == 64 and args.table_update == "lazy" and not args.table_sharding:
        # (4) what the lazy schedule costs against the dense sweep it replaces: the amortised flush
        # (every row brought current before an evaluation / checkpoint), and the same eager step with
        # table_update="dense" minus its measured sweep kernel.
        model.train()
        model._table_hooks = None
        opt_d = PxrAdamW(model, lr=1e-4, weight_decay=0.1, table_update="dense")
        model._table_hooks = None            # (dense: nothing to catch up before the forward)
        opt_l, opt = opt, opt_d
        ctx.state.opt = opt_d                # (eager_step steps whatever optimizer the shared state names)
        evs = []
        for _ in range(3):
            eager_step(next_batch())
        torch.cuda.synchronize()
        td = time.perf_counter()
        for _ in range(20):
            eager_step(next_batch())
        torch.cuda.synchronize()
        td = (time.perf_counter() - td) / 20
        ops.GEMM_TIMING = evs
        for _ in range(5):
            eager_step(next_batch())
        ops.GEMM_TIMING = None
        torch.cuda.synchronize()
        sw = [s_.elapsed_time(e_) * 1e-3 for s_, e_, w, t in evs if t.startswith("adamw_table_kernel")]
        sweep = sum(sw) / max(len(sw), 1)
        opt = opt_l
        ctx.state.opt = opt_l
        model.register_table_hooks(opt)
        out["lazy_vs_dense"] = {"lazy_graphed_ms_per_step": dt / args.steps * 1e3 if use_graph else None,
                                "lazy_eager_ms_per_step": (t_eager or float("nan")) * 1e3, "dense_eager_ms_per_step": td * 1e3,
                                "dense_sweep_kernel_ms": sweep * 1e3, "dense_minus_sweep_ms": (td - sweep) * 1e3,
                                "lazy_graphed_over_dense_minus_sweep": (dt / args.steps / (td - sweep)) if use_graph else None,
                                "lazy_eager_over_dense_minus_sweep": (t_eager or float("nan")) / (td - sweep),
                                "note": "what the lazy table schedule costs against the dense sweep it replaces.  `dense` (eager only: its "
                                        "sweep takes host scalars) sweeps p, m, v of all rows every step and is device-bound, so "
                                        "dense - sweep = the device time of everything else in the step; the lazy step as the headline runs "
                                        "it (captured graph: input rows' catch-up in front of the forward, the other rows' beside it, the "
                                        "apply at the end, on the aged non-repeating stream) is compared with that.  The EAGER lazy step is "
                                        "bound by the host issuing ~50 launches and two stream joins, not by the device: its ratio says "
                                        "nothing about the schedule"}
        del opt_d

    if not args.no_extras and world == 1 and not custom and B == 64:
        # (5) BASELINE configs[2]-shaped PixelNet step (ViT-B/16 tower trained end to end, 352 images per step), a few steps: the
        # line `bench.py --model pixelnet` prints, embedded so that the default run carries it
        gstep = None
        torch.cuda.empty_cache()
        pa = argparse.Namespace(**{**vars(args), "encoder": "clip-vit-base-patch16", "batch": 64})
        def px_entry(px):
            return {"metric": px["metric"], "value": px["value"], "unit": px["unit"], "ms_per_step": px["ms_per_step"],
                    "operands": px["operands"], "images_per_s": px["images_per_s"], "steps": px["steps"], "config": px["config"],
                    "gemm_family": {k: px["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "gemm_time_per_step_ms",
                                                                   "launches_per_step", "kernel")},
                    "phases_ms": px["data_parallel_phases"]["ms"]}

        try:
            out["pixelnet"] = px_entry(pixelnet_run(pa, steps=5, warmup=2, init_dist=False, n_inst=2))
            # the same step on the six-product operands everywhere, same session
            with _env(PXR_TOWER_H2="0", PXR_SEQ_H2="0"):
                torch.cuda.empty_cache()
                out["pixelnet_six_products"] = px_entry(pixelnet_run(pa, steps=5, warmup=2, init_dist=False, n_inst=2))
        except Exception as e:  # noqa: BLE001  (the headline line must not die with an extra)
            out.setdefault("pixelnet", {"error": f"{type(e).__name__}: {e}"})
            out.setdefault("pixelnet_six_products", {"error": f"{type(e).__name__}: {e}"})

    if "roofline_gather" in out:
        # where the two north-star kernel targets stand, in one place: the gather as a standalone kernel and
        # as the step runs it (fused into the input LayerNorm, which also writes y, xhat and the planes of y); the scoring
        # product as the literal GEMM, as the fused six-product top-k pass, and against the peak at the clock the part sustained
        as_run = {}
        for e_ in out.get("roofline_gather_fused", []):
            key = f"as_run_b{e_['batch_per_gpu']}" + ("_" + e_["operands"] if e_["batch_per_gpu"] != B else "")
            as_run[key] = e_["frac"]
        as_run_big = [e_["frac"] for e_ in out.get("roofline_gather_fused", []) if e_["batch_per_gpu"] >= 512]
        alone_ = {f"alone_b{e_['batch_per_gpu']}": e_["frac"] for e_ in out.get("roofline_gather_fused_alone", {}).get("batches", [])}
        if alone_:      # the kernel timed alone is the better measurement of the same launch
            as_run_big = [v for k, v in alone_.items() if k != "alone_b64"]
            as_run.update(alone_)
        sc_ = out.get("roofline_scoring", {})
        out["targets"] = {
            "gather_ge_0.70_of_hbm_peak": {"standalone_embed_gather_kernel": out["roofline_gather"]["frac"], **as_run,
                                           "met": bool(as_run_big and min(as_run_big) >= 0.70),
                                           "hbm_stream_copy_frac": out.get("hbm_stream_copy", {}).get("frac"),
                                           "note": "`met` = the gather AS THE STEP RUNS IT (fused into the input LayerNorm, priced with "
                                                   "everything that launch reads and writes) at every batch >= 512; the standalone kernel "
                                                   "(never launched by the step) is listed for reference.  hbm_stream_copy_frac: what a plain "
                                                   "1 GiB device copy reaches on this box (read + write counted) -- at 102 400 rows, where the "
                                                   "outputs no longer fit the 256 MB MALL, the fused launch runs at ~0.85-0.9 of that"},
            "scoring_ge_0.60_of_mfma_peak": {"literal_gemm": out.get("roofline_scoring_literal", {}).get("frac"),
                                             "fused_topk_six_products": sc_.get("frac"),
                                             "fused_topk_default_three_products": out.get("roofline_scoring_fused_topk", {}).get("frac"),
                                             "sustained_clock_ghz": out.get("roofline_scoring_fused_topk", {}).get("sustained_clock_ghz"),
                                             "frac_of_peak_at_sustained_clock": out.get("roofline_scoring_fused_topk", {}).get("frac_of_sustained_peak"),
                                             "met": bool((sc_.get("frac") or 0) >= 0.60),
                                             "note": "fractions of the dense bf16 MFMA peak on EXECUTED products (6 per fp32 multiply); not met.  "
                                                     "frac_of_peak_at_sustained_clock prices the same rate against 256 CUs x 4 SIMDs x 1024 "
                                                     "flop/clk at the clock the main-pass kernel recorded for itself: the part runs these "
                                                     "kernels at 1.55-1.9 GHz (power limit), and three main-loop structures finish within 5 % "
                                                     "of each other at different clocks (profiles/r06/lab) -- the ceiling is the part's, see DESIGN.md"}}

    if "targets" in out:
        # ... and the step-level figures the round-4 review set (B = 64 step, B = 2 048 throughput, PixelNet step), with both arithmetics
        tb = {(t_["batch_per_gpu"], t_["operands"]): t_ for t_ in out.get("throughput_batches", []) if "batch_per_gpu" in t_}
        b2k_h2, b2k_six = tb.get((2048, OPERANDS_H2), {}), tb.get((2048, OPERANDS_B3), {})
        px, px6 = out.get("pixelnet", {}), out.get("pixelnet_six_products", {})
        out["targets"].update({
            "b64_step_le_0.88_ms": {"ms_per_step": out["ms_per_step"], "operands": out.get("operands"),
                                    "six_products_ms_per_step": out.get("six_products", {}).get("ms_per_step"),
                                    "met": bool(out["ms_per_step"] <= 0.88),
                                    "met_on_six_products": bool((out.get("six_products", {}).get("ms_per_step") or 9e9) <= 0.88)},
            "b2048_ge_150k_sequences_per_s": {"value": b2k_h2.get("value"), "six_products_value": b2k_six.get("value"),
                                              "met": bool((b2k_h2.get("value") or 0) >= 150e3)},
            "pixelnet_step_le_60_ms": {"ms_per_step": px.get("ms_per_step"), "six_products_ms_per_step": px6.get("ms_per_step"),
                                       "met": bool((px.get("ms_per_step") or 9e9) <= 60.0)}})
        rr_ = out.get("roofline_adamw_rows", {})
        if rr_:
            # the three lazy-table launches of a step: kernel time inside the replayed graph, from the committed
            # rocprofv3 summary; the HIP-event figure of this run's eager steps beside it
            rp_us = (rr_.get("rocprofv3") or {}).get("us_per_step")
            out["targets"]["lazy_adamw_rows_le_50_us"] = {"rocprofv3_us_per_step": rp_us, "eager_events_us_per_step": rr_.get("us_per_step"),
                                                          "met": bool(rp_us is not None and rp_us <= 50.0)}

