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


def test_bench_parent_stops_everything_when_a_rank_dies():
    """A rank that dies before the rendezvous must not leave the others waiting in it: the self-launching parent polls its
    children, stops the survivors (fresh child processes, by PID), says which rank failed and returns non-zero - well inside
    30 s, not after a collective timeout."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    t = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "64",
                        "--dry-run", "--fail-rank", "1"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    dt = time.monotonic() - t
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert dt < 30.0, dt
    assert "rank 1 exited with code 3" in r.stderr and "--fail-rank" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no JSON line from a failed run
    assert "--fail-rank" in open(os.path.join(ROOT, "gpurun_out", "rank1.err")).read()


def test_bench_parent_enforces_the_launch_timeout():
    """Rank 0 missing: rank 1 sits in the rendezvous; the parent's overall limit ends it and the exit code says so."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    t = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "64",
                        "--dry-run", "--fail-rank", "-2", "--launch-timeout", "8", "--init-timeout", "60"], capture_output=True, text=True,
                       timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and time.monotonic() - t < 40.0
    assert "--launch-timeout" in r.stderr


def test_bench_multi_launch_dry_run():
    """`--launch multi`: one process, N devices through spcies_hip_create_multi.  Dry run = the shard arithmetic the library
    applies (spcies_hip_shard_range), no device."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--launch", "multi", "--gpus", "8", "--batch", "1000", "--dry-run"],
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["launch"] == "multi" and out["n_gpus"] == 8 and out["batch_total"] == 8000
    assert out["shards"] == [[1000 * g, 1000] for g in range(8)]


def test_merge_multi_leg_into_the_json_line():
    import bench
    merged = bench._merge_multi_leg('noise\n{"value": 1.0, "n_gpus": 2}\n', {"launch": "multi", "value": 2.0})
    out = _json_line(merged)
    assert out["value"] == 1.0 and out["multi_launch"]["value"] == 2.0 and merged.count("\n{") == 1


def test_design_traffic_reads_only_a_profile_of_the_same_variant():
    """`roofline.traffic` of a side configuration comes from the committed PMC summary of the SAME (config, variant) - a profile of
    another kernel for the same configuration (C4: MFMA4G, 857 GB per launch) must not be quoted for the default one (MFMA4R)."""
    import bench
    t4r, src = bench.design_traffic("C4", "mfma4r")
    t4g, srcg = bench.design_traffic("C4", "mfma4g")
    assert src and "C4_mfma4r" in src and srcg and "C4_mfma4g" in srcg
    assert 0.05 * t4g < t4r < 0.2 * t4g  # the on-chip kernel moves about an eighth of the streaming one's bytes
    assert bench.design_traffic("C4", "stream") == (None, None)


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
