"""CPU: bench.py's `cpu_baseline` leg (the oracle timed on host cores) runs and reports the contract fields;
the JSON-line schema helpers do not need a GPU."""
import argparse
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_cpu_baseline_leg_small(oracle_built):
    b = _load_bench()
    args = argparse.Namespace(image_size=32, subdivide=1, batch=2)
    r = b.cpu_baseline(args, 1)
    assert r["unit"] == "images/s" and r["kind"] == "port" and r["cores"] >= 1 and r["value"] > 0
    assert "sample" in r and "train_s1" in r["sample"]
    assert b.METRIC.startswith("train images/sec") and b.HBM_PEAK_GBS == 8000.0
