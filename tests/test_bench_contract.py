"""The bench line contract (one JSON object: metric/value/unit/..., a `roofline` object -- `achieved` from the HBM bytes the
kernel really moves, the SURVEY 8(d) algorithmic-byte figure beside it -- and a `cpu_baseline` object), checked on the
committed lines of the last profiled run (profiles/r02_final_bench_*.json) and on
bench.py's own source: the product path may not fall back to the oracle, the oracle is only the cpu_baseline leg."""
import json
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINES = [os.path.join(ROOT, "profiles", f) for f in ("r02_final_bench_c4.json", "r02_final_bench_c2.json")]


@pytest.mark.parametrize("path", LINES, ids=[os.path.basename(p) for p in LINES])
def test_committed_bench_line_has_the_contract_fields(path):
    r = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in r, k
    assert r["higher_is_better"] is True and r["scaling"] == "weak" and r["vs_baseline"] is None      # BASELINE.md publishes no number
    assert r["dtype"] == "f64" and r["data"] == "synthetic" and "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] > 0 and r["ms_per_step"] > 0 and r["n_gpus"] == 1
    rf = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1      # a fraction of a physical peak
    # achieved = bytes really moved per launch (PMC-measured, else the layout's compulsory bytes) / measured launch time
    assert abs(rf["achieved"] - rf["bytes_per_launch_used_for_achieved"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["achieved"]
    assert rf["bytes_per_launch_used_for_achieved"] in (rf["traffic"], rf["compulsory_bytes_this_layout"])
    assert (rf["traffic"] is None) == (rf["traffic_source"] is None)
    assert abs(rf["algorithmic_achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * rf["algorithmic_achieved"]
    cb = r["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("port", "reference") and cb["unit"] == r["unit"] and cb["cores"] >= 1
    assert "extrapolated" not in json.dumps(cb) and cb["lm_iterations"] == cb["lm_iterations_gpu"] == r["lm_iterations_per_resolve"]
    assert cb["cores"] == cb["host"]["threads_used"] <= cb["host"]["affinity"]


def test_bench_source_keeps_the_oracle_in_its_lane():
    src = open(os.path.join(ROOT, "bench.py")).read()
    # the oracle is imported exactly once, inside the cpu_baseline leg
    assert len(re.findall(r"from oracle import", src)) == 1
    i = src.index("from oracle import")
    assert "CPU baseline" in src[max(0, i - 900):i]
    assert "no CPU fallback" in src          # a missing GPU is an error, not a detour
