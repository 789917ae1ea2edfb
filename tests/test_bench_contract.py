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


def _json_line(stdout):
    return json.loads([l for l in stdout.splitlines() if l.startswith("{")][-1])


def test_bench_self_launches_two_ranks_dry_run():
    """`python bench.py --gpus 2` with no launcher environment: the parent starts two fresh rank processes (gloo on CPU
    in --dry-run: no solver, no GPU), both rendezvous on 127.0.0.1, rank 0's JSON line is relayed."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--dry-run"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["dry_run"] is True and out["n_gpus"] == 2 and out["rccl_ranks_seen"] == 2
    assert len(out["per_rank_ms_per_step"]) == 2 and out["config"]["launch"] == "self"
    assert abs(out["ms_per_step"] - max(out["per_rank_ms_per_step"])) < 1e-9  # MAX over ranks
    assert out["scaling"] == "weak" and out["steps"] == 2 and out["warmup"] == 1


def test_bench_under_torch_distributed_run_dry_run():
    """The driver's launch line for N > 1 (torch.distributed.run, 127.0.0.1), rehearsed on CPU."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2",
                        "--warmup", "0", "--batch", "64", "--dry-run"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["rccl_ranks_seen"] == 2 and out["config"]["launch"] == "torchrun"


def test_bench_gpus_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE=3" in r.stderr


@pytest.mark.gpu
def test_bench_line_has_every_contract_field():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--batch", "8192",
                        "--no-cpu-baseline", "--configs", "C5_soc", "--config-steps", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT)
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
    assert "traffic_source" in rf and "hbm_frac" in rf and out["rccl_ranks_seen"] == 1
    assert out["pcie_inclusive"]["solves_per_s"] > 0
    c = out["configs"]["C5_soc"]
    assert "error" not in c, c
    assert c["oracle_check"]["max_abs_du"] <= 1e-10 and c["roofline"]["frac"] > 0
