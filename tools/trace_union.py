"""GEMM-family busy time of a rocprofv3 --kernel-trace of bench.py: the weight-gradient launches co-run with the
input-gradient chain (a second stream / a parallel graph branch), so the family's time per step is the UNION of the
kernels' [start, end] intervals, not the sum of their durations.
usage: python tools/trace_union.py <kernel_trace.csv> <steps_in_trace> [out.json]"""
import csv
import json
import sys


def main():
    f, steps = sys.argv[1], int(sys.argv[2])
    iv, per = [], {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        key = n.replace("void ", "").split("(")[0][:100]
        p = per.setdefault(key, [0, 0])
        p[0] += 1; p[1] += e - s
        if "gemm_kernel" in n or "grouped_dw" in n or "gemm_b3_kernel" in n or "gemm_sk_kernel" in n:
            iv.append((s, e))
    iv.sort()
    union, cs, ce = 0, None, None
    for s, e in iv:
        if ce is None or s > ce:
            if ce is not None:
                union += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if ce is not None:
        union += ce - cs
    total = sum(e - s for s, e in iv)
    flops = 80530636800.0
    out = {"gemm_launches": len(iv), "steps": steps, "gemm_union_us_per_step": union / steps * 1e-3,
           "gemm_sum_of_durations_us_per_step": total / steps * 1e-3,
           "tflops_union": flops / (union / steps * 1e-9) / 1e12 if union else None,
           "frac_of_157.3": flops / (union / steps * 1e-9) / 1e12 / 157.3 if union else None,
           "per_kernel_avg_us": {k: round(v[1] / v[0] * 1e-3, 2) for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]},
           "per_kernel_calls": {k: v[0] for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]}}
    s = json.dumps(out, indent=1)
    print(s)
    if len(sys.argv) > 3:
        open(sys.argv[3], "w").write(s)


if __name__ == "__main__":
    main()
