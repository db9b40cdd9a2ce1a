"""Run tools/diag/libgemm_timeline.so (see gemm_timeline.hip) on the training-step GEMM shapes and print where a launch's
time goes: effective shader clock, launch ramp, tiles per CU (grid quantisation), main-loop cycles per K tile, tail."""
import ctypes
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(HERE, "libgemm_timeline.so"))
lib.gemm_timeline.restype = ctypes.c_int
P, I, L = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
lib.gemm_timeline.argtypes = [I, I, I, I, I, P, L, P, L, P, L, I, P, P]
dev = "cuda"
STAMP = np.dtype([("c0", "<u8"), ("c1", "<u8"), ("c2", "<u8"), ("r0", "<u8"), ("r2", "<u8"), ("xcc", "<u4"),
                  ("hwid", "<u4"), ("bid", "<u4"), ("tile", "<u4")])
out = {}
for name, akc, bkc, M, N, K in [("fwd_qkv", 1, 1, 3200, 1536, 512), ("fwd_o", 1, 1, 3200, 512, 512),
                                ("fwd_f1", 1, 1, 3200, 1024, 512), ("fwd_f2", 1, 1, 3200, 512, 1024),
                                ("dx_qkv", 1, 0, 3200, 512, 1536), ("dx_o", 1, 0, 3200, 512, 512)]:
    A = torch.randn(M, K, device=dev)
    B = torch.randn(N, K, device=dev) if bkc else torch.randn(K, N, device=dev)
    C = torch.empty(M, N, device=dev)
    tiles = ((M + 63) // 64) * ((N + 63) // 64)
    st = torch.zeros(tiles * STAMP.itemsize, dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for it in range(4):
        if it == 3:
            ev[0].record()
        rc = lib.gemm_timeline(akc, bkc, M, N, K, A.data_ptr(), K, B.data_ptr(), K if bkc else N, C.data_ptr(), N, 1,
                               st.data_ptr(), s)
        assert rc == 0
    ev[1].record()
    torch.cuda.synchronize()
    a = np.frombuffer(st.cpu().numpy().tobytes(), dtype=STAMP)
    t0 = a["r0"].min()
    wall = (a["r2"].max() - t0) / 100.0                       # us (100 MHz)
    life_us = (a["r2"] - a["r0"]) / 100.0
    cyc = (a["c2"] - a["c0"]).astype(np.float64)
    clk = np.median(cyc / np.maximum(life_us, 1e-3)) / 1e3     # GHz
    cu = (a["xcc"].astype(np.int64) & 0xF) * 65536 + (a["hwid"].astype(np.int64) & 0xFF00)
    ucu, cnt = np.unique(cu, return_counts=True)
    per_cu_end = np.array([(a["r2"][cu == u].max() - t0) / 100.0 for u in ucu])
    nk = (K + 31) // 32
    rec = {"tiles": int(tiles), "event_us": ev[0].elapsed_time(ev[1]) * 1e3, "span_us_first_start_to_last_end": float(wall),
           "start_spread_us": float((a["r0"].max() - t0) / 100.0), "clock_ghz_median": float(clk),
           "wg_life_us": {"min": float(life_us.min()), "median": float(np.median(life_us)), "max": float(life_us.max())},
           "mainloop_cycles_per_ktile": {"median": float(np.median((a["c1"] - a["c0"]) / nk)),
                                         "p90": float(np.percentile((a["c1"] - a["c0"]) / nk, 90))},
           "epilogue_cycles_median": float(np.median(a["c2"] - a["c1"])),
           "cus_used": int(len(ucu)), "tiles_per_cu_hist": {int(k): int(v) for k, v in zip(*np.unique(cnt, return_counts=True))},
           "cu_finish_us": {"p10": float(np.percentile(per_cu_end, 10)), "median": float(np.median(per_cu_end)),
                            "max": float(per_cu_end.max())},
           "ideal_us_at_157TF": 2.0 * M * N * K / 157.3e12 * 1e6}
    # do WGs sharing a CU run concurrently or back to back?  overlap fraction of lifetimes per CU
    out[name] = rec
    print(name, json.dumps(rec))
json.dump(out, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "gemm_timeline.json"), "w"), indent=1)
