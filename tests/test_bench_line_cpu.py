"""bench.py prints ONE stdout line of < 4 KB (round 5's 21.6 KB line was not parsed by the driver): the line is built from the full record by
`bench.headline_line`, which is exercised here on a canned record — round 5's committed full record (profiles/r05_bench_line.json) and a synthetic worst case."""
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def _canned():
    with open(os.path.join(REPO, "profiles", "r05_bench_line.json")) as fh:
        return json.load(fh)


def test_headline_line_is_small_and_keeps_the_contract_fields():
    d = _canned()
    s = bench.headline_line(d, "bench_detail.json")
    assert "\n" not in s and len(s.encode()) < bench.MAX_LINE_BYTES
    line = json.loads(s)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"].startswith("BASELINE config 3")
    for k in ("max_samples", "render_step_size", "cone_angle", "alpha_thre", "samples_per_ray"):
        assert k in line["config"], k
    r = line["roofline"]
    assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(r)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["frac"] - d["roofline"]["frac"]) < 1e-5 and abs(line["value"] / d["value"] - 1) < 1e-5
    b = line["cpu_baseline"]
    assert set(("value", "unit", "cores", "kind", "sample")) <= set(b) and b["kind"] == "port" and len(b["sample"]) <= 260
    assert line["train_ms"] == pytest.approx(d["train"]["ms_per_step"], rel=1e-5)
    assert line["train_refyaml_ms"] == pytest.approx(d["train_refyaml"]["ms_per_step"], rel=1e-5)
    assert line["score256_ms"] == pytest.approx(d["score256"]["ms_per_pass"], rel=1e-5)
    assert line["bench_parity_ok"] is True and line["detail"] == "bench_detail.json"


def test_headline_line_survives_bloated_and_missing_sections():
    d = _canned()
    d["config"]["workload"] = "x" * 5000
    d["config"]["weights"] = "y" * 5000
    d["cpu_baseline"]["sample"] = "z" * 5000
    d["train"]["kernels"] = {f"k{i}": {"ms": i * 1.0} for i in range(500)}
    s = bench.headline_line(d, None)
    assert len(s.encode()) < bench.MAX_LINE_BYTES
    for k in ("roofline", "cpu_baseline", "train", "score256", "bench_parity", "train_refyaml"):
        d.pop(k, None)
    line = json.loads(bench.headline_line(d, None))
    assert "roofline" not in line and "train_ms" not in line and line["value"] == pytest.approx(d["value"], rel=1e-5)


def test_cli_defaults_finish_within_minutes_and_flags_exist():
    a = bench.parse([])
    assert a.gpus == 1 and a.steps == 10 and a.warmup == 3 and a.workload == "all" and not a.full
    a = bench.parse(["--gpus", "8", "--steps", "5", "--warmup", "2", "--workload", "config2", "--full"])
    assert (a.gpus, a.steps, a.warmup, a.workload, a.full) == (8, 5, 2, "config2", True)
