"""bench.py: the host-side baseline helpers (CPU) and the JSON line the driver reads (GPU)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_cpu_baseline_helpers_on_a_small_sample():
    import bench
    from spcies_amd import benchmarks
    assert 1 <= bench.host_threads() <= len(os.sched_getaffinity(0))
    cfg = benchmarks.config("C2")
    v = benchmarks.ingredients(cfg)
    out = bench.cpu_baseline(cfg, v, 96, 2, seconds=0.5)
    assert out["kind"] == "port" and out["cores"] == 2 and out["unit"] == "solves/s"
    assert out["value"] > 0 and out["one_thread_value"] > 0 and "96 seeded C2 instances" in out["sample"]
    tmpl = bench.cpu_reference_baseline(cfg, 2, seconds=0.3)  # None unless oracle/_ref/libbench_C2_lax.so was built here
    assert tmpl is None or (tmpl["kind"] == "template" and tmpl["value"] > 0)


@pytest.mark.gpu
def test_bench_line_has_every_contract_field():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "8192",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in out, key
    assert out["unit"] == "solves/s" and out["n_gpus"] == 1 and out["steps"] == 3 and out["warmup"] == 1
    assert out["higher_is_better"] is True and out["scaling"] == "weak" and out["dtype"] == "f64" and out["data"] == "synthetic"
    assert out["vs_baseline"] is None and "workload" in out["config"] and "model" not in out["config"]
    rf = out["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["bound"] in ("hbm", "mfma") and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert abs(out["value"] - 8192 / out["ms_per_step"] * 1e3) / out["value"] < 1e-6
