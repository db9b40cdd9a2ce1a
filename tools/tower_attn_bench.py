"""Fused tower attention against the three-launch path it replaces (batched QK^T, softmax rows, batched PV) at the shipped
PixelNet shape: 352 images x 12 heads, T = 197, head size 64 (ViT-B/16) and 16 heads, T = 257 (ViT-L/14).
usage (GPU box): python tools/tower_attn_bench.py"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch

from pixelrec_amd import ops


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    out = []
    for name, n, heads, T in (("vit-b16", 352, 12, 197), ("vit-l14", 352, 16, 257), ("vit-b16 eval 1024", 1024, 12, 197)):
        d = 64
        H, ld, Tp = heads * d, 3 * heads * d, (T + 3) & ~3
        qkv = torch.randn(n * T, ld, device="cuda")
        S = torch.empty(n * heads, T, Tp, device="cuda")
        ctx = torch.empty(n, T, H, device="cuda")
        bh = n * heads

        def mat():
            ops.gemm_batched(True, True, T, T, d, qkv, 2 * H, ld, qkv, 0, ld, S, 0, Tp, bh, heads, (T * ld, d), (T * ld, d),
                             (heads * T * Tp, T * Tp))
            ops.softmax_rows(S, bh * T, T, Tp, d ** -0.5)
            ops.gemm_batched(True, False, T, d, T, S, 0, Tp, qkv, H, ld, ctx, 0, H, bh, heads, (heads * T * Tp, T * Tp),
                             (T * ld, d), (T * H, d))
            return ops.split_planes(ctx.view(n * T, H))

        fused_f32 = lambda: ops.tower_attn_fwd(qkv, n, T, heads, d, 2 * H, 0, H, d ** -0.5)
        fused_pl = lambda: ops.tower_attn_fwd(qkv, n, T, heads, d, 2 * H, 0, H, d ** -0.5, ctx=False, planes=True)
        t_mat, t_f, t_p = timed(mat), timed(fused_f32), timed(fused_pl)
        flops = 4.0 * T * T * d * bh
        r = {"shape": name, "images": n, "heads": heads, "T": T, "materialized_plus_split_us": t_mat, "fused_fp32_out_us": t_f,
             "fused_planes_out_us": t_p, "algorithmic_tflops_fused": flops / t_p / 1e6,
             "frac_of_bf16_pipe_6_products": 6 * flops / t_p / 1e6 / 2500.0}
        err = float((fused_f32()[0] - ctx).abs().max())
        r["max_abs_diff_vs_materialized"] = err
        print(json.dumps(r)); out.append(r)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/tower_attn_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
