"""The torch.distributed leg of SURVEY 8e on the hardware a builder has: ONE GPU.

The N > 1 path (spcies_amd/distributed.py, bench.py run_rank) is one process per GPU, one RCCL broadcast of the problem blob, no
collective inside the iteration.  gloo covers the plumbing at world size 2 on CPU (tests/test_distributed_gloo.py); what gloo cannot
cover is RCCL itself - communicator set-up on a device, collectives on DEVICE tensors, teardown next to a live solver handle.  These
tests run that code at world size 1 in a FRESH child process (the test process holds a GPU context and is never re-exec'ed)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CHILD = os.path.join(ROOT, "tests", "_rccl_world1_child.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_process_group_of_one_runs_every_collective_gloo(tmp_path):
    """CPU twin: with a process group initialised the collectives RUN also at world size 1 (they used to be skipped)."""
    r = subprocess.run([sys.executable, CHILD, "gloo", str(tmp_path), "37", "0", "1", str(_free_port())], capture_output=True, text=True,
                       timeout=300, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.load(open(tmp_path / "report.json"))
    assert rep == {"ranks_seen": 1, "blob_len": rep["blob_len"], "blob_equal": True, "backend": "gloo", "world": 1, **{}}
    from spcies_amd import benchmarks
    x0, _, _ = benchmarks.sample_batch(benchmarks.config("C2"), 37)
    assert np.array_equal(np.load(tmp_path / "x0.npy"), x0)


def test_bench_force_dist_dry_run():
    for flag, launch in (("--force-dist", "single"), ("--self-launch", "self")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "64",
                            "--dry-run", flag], capture_output=True, text=True, timeout=300, cwd=ROOT, env=_env())
        assert r.returncode == 0, r.stderr[-2000:]
        out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert out["config"]["process_group"] == "gloo" and out["config"]["launch"] == launch and out["rccl_ranks_seen"] == 1


@pytest.mark.gpu
def test_rccl_world_size_one_on_the_device(tmp_path):
    """RCCL on the GPU: init_process_group("nccl", world_size=1, device_id=cuda:0), the blob broadcast on device tensors, all_reduce,
    barrier, all_gather, destroy_process_group - then the solver built from the BROADCAST bytes solves 1 003 instances and returns
    exactly what a handle built in this process from the local blob returns."""
    total = 1003
    r = subprocess.run([sys.executable, CHILD, "nccl", str(tmp_path), str(total), "0", "1", str(_free_port())], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=_env())
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    rep = json.load(open(tmp_path / "report.json"))
    assert rep["backend"] == "nccl" and rep["world"] == 1 and rep["ranks_seen"] == 1 and rep["blob_equal"] is True
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver
    cfg = benchmarks.config("C2")
    s = HipSolver(benchmarks.ingredients(cfg))
    assert s.variant == rep["variant"]
    x0, xr, ur = benchmarks.sample_batch(cfg, total)
    u, k, e, _ = s(x0, xr, ur, want_sol=False)
    s.close()
    assert np.array_equal(np.load(tmp_path / "u.npy"), u) and np.array_equal(np.load(tmp_path / "k.npy"), k)
    assert (k == 200).all() and (e == -1).all()


@pytest.mark.gpu
def test_bench_runs_its_rccl_leg_on_one_gpu():
    """bench.py --gpus 1 --self-launch: the self-launcher (fresh rank process, watchdog, per-rank stderr files), `nccl` rendezvous, the
    broadcast / all_reduce / barriers / all_gather around the timed region and the teardown - the code `--gpus 8` executes."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--batch", "8192",
                        "--self-launch", "--no-cpu-baseline", "--no-configs", "--no-pcie"], capture_output=True, text=True, timeout=900,
                       cwd=ROOT, env=_env())
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["process_group"] == "nccl" and out["config"]["launch"] == "self" and out["rccl_ranks_seen"] == 1
    assert out["n_gpus"] == 1 and out["config"]["all_k_200_eflag_-1"] is True and out["value"] > 0
    assert "teardown_error" not in out
    assert out["multi_launch"].get("last_shard_equals_single_device") is True, out["multi_launch"]


@pytest.mark.gpu
def test_bench_under_torchrun_with_one_rank():
    """The driver's N > 1 launch line - `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...` - with N = 1 and --force-dist on the one GPU here: the launcher's environment (RANK / LOCAL_RANK / WORLD_SIZE /
    TORCHELASTIC_RUN_ID), the `nccl` rendezvous through the launcher's store, every collective of the N > 1 path, the teardown, and rank 0's
    one-process `multi_launch` leg in a fresh child."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--batch", "8192", "--force-dist", "--no-cpu-baseline", "--no-configs", "--no-pcie"], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=_env())
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-3000:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"]["process_group"] == "nccl" and out["config"]["launch"] == "torchrun" and out["rccl_ranks_seen"] == 1
    assert out["n_gpus"] == 1 and out["config"]["all_k_200_eflag_-1"] is True and "teardown_error" not in out
    assert out["multi_launch"].get("last_shard_equals_single_device") is True, out["multi_launch"]
