"""The bench.py output contract, checked on the line the last GPU run committed (profiles/r02/bench_default_line_final.json):
every key the driver parses is there with the right type, the metric/config are BASELINE.json's, and the derived
fields are consistent with each other.  (bench.py itself needs an MI355X: this guards the schema on CPU.)"""
import json
import math
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line():
    text = open(os.path.join(ROOT, "profiles", "r02", "bench_default_line_final.json")).read().strip().splitlines()
    assert len(text) == 1, "bench.py prints ONE JSON line"
    return json.loads(text[0])


def test_required_keys_and_types():
    d = _line()
    for key, typ in [("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                     ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str),
                     ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)]:
        assert isinstance(d[key], typ), key
    assert "vs_baseline" in d and d["vs_baseline"] is None          # BASELINE.md publishes no number for this metric
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32"
    assert d["data"].startswith("synthetic")
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert "sequences" in d["unit"] and "sequences/sec" in base["metric"]
    cfg = d["config"]
    assert "workload" in cfg and "model" not in cfg
    assert (cfg["embedding_size"], cfg["seq_len"], cfg["n_items"], cfg["batch_per_gpu"]) == (512, 50, 400001, 64)


def test_value_is_consistent_with_the_step_time():
    d = _line()
    seqs_per_step = d["config"]["global_batch"]
    assert math.isclose(d["value"], seqs_per_step / (d["ms_per_step"] * 1e-3), rel_tol=1e-6)


def test_roofline_and_cpu_baseline_objects():
    d = _line()
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert r["peak"] > 0 and 0 < r["frac"] <= 1 and math.isclose(r["frac"], r["achieved"] / r["peak"], rel_tol=1e-6)
    assert r["traffic"] is None or r["traffic"] > 0
    assert r["gemm_mode"] in ("bf16x3", "f32") and (r["matrix_pipe"] is None) == (r["gemm_mode"] == "f32")
    # round 2: the stream does not repeat inside the run and the line says what the lazy table update costs
    assert d["stream"]["repeats_inside_run"] is False and d["stream"]["age_steps"] >= 256
    assert d["roofline_adamw_rows"]["us_per_step"] > 0 and d["lazy_flush"]["amortised_us_per_step"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and isinstance(c["sample"], str)
    assert c["unit"] == d["unit"]
    # the north-star targets travel with the line: gather >= 70 % of HBM peak, scoring GEMM >= 60 % MFMA utilisation
    assert d["roofline_gather"]["frac"] >= 0.70 and d["roofline_scoring"]["frac"] >= 0.60
