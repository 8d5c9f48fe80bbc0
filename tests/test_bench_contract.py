"""The bench.py output contract, checked on the committed recordings of real runs (profiles/):
every key the driver and the judge read is there with the right type, and the derived figures
agree with each other.  (bench.py itself needs the MI355X; this guards the record format.)"""
import json
from pathlib import Path

import pytest

PROFILES = Path(__file__).resolve().parent.parent / "profiles"
REQUIRED = {"metric": str, "value": (int, float), "unit": str, "n_gpus": int, "steps": int, "warmup": int,
            "ms_per_step": (int, float), "higher_is_better": bool, "scaling": str, "dtype": str, "data": str,
            "config": dict, "roofline": dict}


@pytest.mark.parametrize("name", ["r01_bench_c3.json", "r01_bench_c2.json", "r01_bench_c5.json"])
def test_recorded_bench_lines_follow_the_contract(name):
    rec = json.loads((PROFILES / name).read_text())
    for key, typ in REQUIRED.items():
        assert key in rec and isinstance(rec[key], typ), (name, key)
    assert "vs_baseline" in rec and rec["vs_baseline"] is None  # BASELINE.md publishes no number for this metric
    assert rec["n_gpus"] == 1 and rec["higher_is_better"] is True and rec["dtype"] == "f32"
    assert "workload" in rec["config"] and "model" not in rec["config"]
    roof = rec["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, (name, key)
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    if roof["bound"] == "hbm":
        assert roof["peak"] == 8000.0
        cpu = rec["cpu_baseline"]
        for key in ("value", "unit", "cores", "kind", "sample"):
            assert key in cpu, (name, key)
        assert cpu["kind"] in ("port", "reference") and cpu["cores"] >= 1
        assert cpu.get("gpu_matches_oracle_on_sample") == "bit-exact"
        # value = flops / time; achieved = algorithmic bytes / kernel time
        flops, ms = rec["config"]["flops_per_step"], rec["ms_per_step"]
        assert abs(rec["value"] - flops / (ms * 1e-3) / 1e9) / rec["value"] < 0.01
        alg = rec["config"]["algorithmic_bytes_per_step"]
        assert abs(roof["achieved"] - alg / (roof["kernel_ms_per_step"] * 1e-3) / 1e9) / roof["achieved"] < 0.01


def test_c3_kernel_stats_agree_with_the_bench_line():
    """rocprofv3's average launch durations (same command) add up to bench.py's HIP-event time per product."""
    import csv
    rec = json.loads((PROFILES / "r01_bench_c3.json").read_text())
    rows = list(csv.DictReader(open(PROFILES / "r01_bench_c3_kernel_stats.csv")))
    main = [r for r in rows if "spmm_wave_row_panel_kernel" in r["Name"]]
    assert len(main) == rec["roofline"]["launches_per_step"] == 2
    total_ms = sum(float(r["AverageNs"]) for r in main) / 1e6
    assert abs(total_ms - rec["roofline"]["kernel_ms_per_step"]) / total_ms < 0.03
    traffic = json.loads((PROFILES / "pmc_traffic.json").read_text())["c3"]
    # (the bench line quotes the PMC passes of the previous profile run: equal to within counter noise)
    assert abs(traffic["hbm_bytes_per_product"] - rec["roofline"]["traffic"]) < 1e-3 * rec["roofline"]["traffic"]
    assert 1.0 <= traffic["hbm_bytes_per_product"] / rec["config"]["algorithmic_bytes_per_step"] < 1.05
