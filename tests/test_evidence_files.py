"""The committed evidence parses and carries what bench.py and DESIGN.md quote from it (no GPU needed)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FINAL = os.path.join(ROOT, "profiles", "r02_final")


def _line(name):
    with open(os.path.join(FINAL, name)) as f:
        for line in reversed(f.read().strip().splitlines()):
            if line.startswith("{"):
                return json.loads(line)
    raise AssertionError(f"{name}: no JSON line")


def test_pmc_traffic_has_both_configs_and_its_calibration():
    with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
        t = json.load(f)
    for key in ("project_views_c2", "project_views_c4"):
        e = t[key]
        assert e["bytes"] == round(e["fetch_kb_raw"] * 1024 * e["fetch_correction"] + e["write_kb"] * 1024) or \
            abs(e["bytes"] - (e["fetch_kb_raw"] * 1024 * e["fetch_correction"] + e["write_kb"] * 1024)) < 2048
        assert e["dispatches"] >= 8 and os.path.exists(os.path.join(ROOT, e["source"].split(",")[0]))
    cal = t["calibration"]["kernels"]
    assert len(cal) == 4 and all(abs(k["factor"] - 2.0) < 0.01 for k in cal.values())      # the x2 of the guide holds for gathers
    assert abs(t["calibration"]["correction_used"] - 2.0) < 0.01


def test_bench_lines_of_the_evidence_run_keep_the_contract():
    for name, shape in (("bench_c2_with_cpu_baseline.json", "c2"), ("bench_c2.json", "c2"), ("bench_c4.json", "c4")):
        d = _line(name)
        for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data", "config", "roofline"):
            assert key in d, (name, key)
        assert d["unit"] == "scenes/s" and d["n_gpus"] == 1 and d["vs_baseline"] is None and d["dtype"] == "f64"
        assert d["config"]["workload"].startswith(shape) and "model" not in d["config"]
        r = d["roofline"]
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
        assert abs(r["achieved"] - r["algorithmic_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) < 1e-6 * r["achieved"]
        assert r["compulsory"]["bytes"] < r["algorithmic_bytes_per_launch"]
        assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    cb = _line("bench_c2_with_cpu_baseline.json")["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "scenes/s" and cb["cores"] >= 1 and 0 < cb["value"] < 1
    assert 5 < cb["sample_seconds"] < 60
