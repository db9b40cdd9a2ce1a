"""Summarise rocprofv3 counter-collection CSVs (one --pmc pass each) per kernel name -> JSON.
usage: python tools/pmc_summarise.py out.json pass1_counter_collection.csv [pass2...]   (values averaged per launch)."""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "").replace("pxr::", "")
    return name.split("(")[0][:120]


def main():
    out, files = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for f in files:
        seen = set()
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            a = acc[k][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
            key = (f, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                d = dur[k]
                d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; d[1] += 1
    res = {}
    for k, cs in acc.items():
        if not any(t in k for t in ("gemm", "grouped_dw", "score_topk", "adamw", "ln_", "attn", "embed", "segsum", "bpr",
                                    "sort", "merge", "flash", "vit", "score_thresh", "rescore", "tower", "split_planes")):
            continue
        res[k] = {c: v[0] / v[1] for c, v in cs.items()}
        res[k]["launches"] = max(v[1] for v in cs.values())
        res[k]["avg_us_under_pmc"] = dur[k][0] / max(dur[k][1], 1)
        if "FETCH_SIZE" in res[k] or "WRITE_SIZE" in res[k]:
            # units KB; gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md §HBM)
            res[k]["hbm_bytes_per_launch"] = (2.0 * res[k].get("FETCH_SIZE", 0.0) + res[k].get("WRITE_SIZE", 0.0)) * 1024.0
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print(json.dumps(res, indent=1, sort_keys=True)[:6000])


if __name__ == "__main__":
    main()
