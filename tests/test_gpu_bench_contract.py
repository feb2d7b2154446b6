"""The bench line the driver parses (bench.py's contract): one JSON line on stdout with the metric of BASELINE.json, the
whole-job value, the timing fields, `roofline` for the dominant kernel and -- at N = 1 -- `cpu_baseline`.  Run as the driver
runs it (a subprocess, `--gpus 1 --steps K --warmup W`), with the optional legs switched off so that the test stays short."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, timeout=600):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *flags], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # ONE JSON line
    return json.loads(lines[0])


def test_driver_form_line_has_the_contract_fields():
    d = _run("--gpus", "1", "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline", "--min-seconds", "0.3")
    assert d["metric"] == "walker-lnprob evals/sec" and d["unit"] == "evals/s"
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    cfg = d["config"]
    assert "MgII" in cfg["workload"] and cfg["walkers_per_gpu"] == 512 and cfg["pixels"] == [4096]
    assert "model" not in cfg
    # value = walkers x steps / time of the timed block; ms_per_step is that block's
    assert d["value"] == pytest.approx(512 / (d["ms_per_step"] * 1e-3), rel=1e-9)
    assert 1e6 < d["value"] < 1e9                                  # (north_star's target is >= 1e6 on this configuration)
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == pytest.approx(8000.0)
    assert r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    # achieved = SURVEY 8(d)'s algorithmic bytes per eval x the evals of one launch / the kernel's average duration
    bytes_per_launch = (24 * 4096 + 8 * 6 + 8) * 512
    assert r["algorithmic_bytes_per_launch"] == bytes_per_launch
    assert r["achieved"] == pytest.approx(bytes_per_launch / (r["avg_kernel_ms"] * 1e-3) / 1e9, rel=1e-9)
    assert "walker_kernel" in r["kernel"]
    assert r["traffic"] is None or 0 < r["traffic"] < bytes_per_launch     # (the spectra are shared through L2 / MALL)


def test_cpu_baseline_object_at_one_gpu():
    os.environ.setdefault("BENCH_CPU_CORES", "4")
    try:
        d = _run("--steps", "50", "--warmup", "5", "--no-extras", "--min-seconds", "0.3", timeout=900)
    finally:
        os.environ.pop("BENCH_CPU_CORES", None)
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0
    assert c["unit"] == "walker-lnprob evals/s"
    assert d["value"] / c["value"] > 100                            # (a reported baseline, not a target)
