"""bench.py's one-line JSON contract (the driver parses it): keys, types, and the roofline / cpu_baseline objects."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REQUIRED = {"metric": str, "value": float, "unit": str, "n_gpus": int, "steps": int, "warmup": int, "ms_per_step": float,
            "higher_is_better": bool, "scaling": str, "dtype": str, "data": str, "config": dict, "roofline": dict}


def test_bench_help_needs_no_gpu():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "--gpus" in out.stdout and "--steps" in out.stdout and "--warmup" in out.stdout


@pytest.mark.gpu
def test_bench_line_contract():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "16384"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    j = json.loads(lines[0])
    for k, t in REQUIRED.items():
        assert k in j, k
        assert isinstance(j[k], t) or (t is float and isinstance(j[k], int)), (k, type(j[k]))
    assert "vs_baseline" in j and j["vs_baseline"] is None  # BASELINE.md holds no published number
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True and j["scaling"] == "weak"
    assert j["dtype"] == "f64" and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    cb = j["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] in ("reference", "port") and cb["value"] > 0 and cb["cores"] >= 1
    assert cb["parity_gpu_vs_det_oracle"]["bit_identical"] is True
    assert j["value"] > 1e6  # the north star's floor, even at this small batch
