#!/usr/bin/env python3
"""bench.py - headline metric of BASELINE.json: MPC solves/sec (whole node), laxMPC-ADMM,
12-state oscillating masses, N=15, 200 iterations (config C2), B = 65 536 instances per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of B synthetic instances that are already
resident in HBM (x0, xr, ur in; u, k, e_flag out).  The batch shards trivially: every rank solves its
own B instances (weak scaling), the only collective is a one-time RCCL broadcast of the problem blob.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SOLVE = 27498 * 200          # 27 498 flop/iteration x 200 iterations
IO_BYTES_PER_SOLVE = 232              # compulsory HBM bytes per solve (per-instance reference)
PEAK_FP64_MFMA_TFLOPS = 78.6          # MI355X dense FP64 matrix peak = 256 CU x 128 flop/clk x 2.4 GHz (spec)
PEAK_HBM_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec


def traffic_from_profile(variant):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_<variant>_pmc_summary.txt: separate FETCH_SIZE / WRITE_SIZE runs of this same command),
    corrected as MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE x 2 on gfx950, values in KB."""
    path = os.path.join(ROOT, "profiles", f"r01_{variant}_pmc_summary.txt")
    if not os.path.exists(path):
        return None
    fetch = write = None
    for line in open(path):
        if f"admm_{variant}_kernel" not in line:
            continue
        val = float(line.split("mean=")[1].split()[0])
        if line.startswith("FETCH_SIZE"):
            fetch = val
        elif line.startswith("WRITE_SIZE"):
            write = val
    if fetch is None or write is None:
        return None
    return (2.0 * fetch + write) * 1024.0


def host_threads():
    """Threads the CPU baseline may use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0))
    try:  # cgroup v2, then v1
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = max(1, min(n, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    return n


def cpu_baseline(cfg, v, n_sample, threads, seconds=12.0):
    """Oracle (the C port of the reference's loop nests) on the host cores, bounded sample of the same workload:
    every instance is solved by the scalar code, instances are spread over `threads` OpenMP threads.
    The sample is sized from a short probe so that the timed run takes about `seconds`; the one-thread rate is
    measured on a small sample of the same instances and reported next to it."""
    from oracle import oracle
    from spcies_amd import benchmarks
    probe = 64 * threads
    x0, xr, ur = benchmarks.sample_batch(cfg, max(probe, 1024), seed=cfg.seed + 1)
    oracle.admm_banded_batch(v, x0[:8], xr[:8], ur[:8], want_sol=False)  # warm
    t = time.perf_counter()
    oracle.admm_banded_batch(v, x0[:1024], xr[:1024], ur[:1024], want_sol=False)
    dt1 = time.perf_counter() - t
    t = time.perf_counter()
    oracle.admm_banded_batch(v, x0[:probe], xr[:probe], ur[:probe], want_sol=False, threads=threads)
    rate = probe / (time.perf_counter() - t)
    if not n_sample:
        n_sample = int(min(max(rate * seconds, probe), 1 << 21))
    x0, xr, ur = benchmarks.sample_batch(cfg, n_sample, seed=cfg.seed + 1)
    t = time.perf_counter()
    oracle.admm_banded_batch(v, x0, xr, ur, want_sol=False, threads=threads)
    dt = time.perf_counter() - t
    return {"value": n_sample / dt, "unit": "solves/s", "cores": threads, "kind": "port",
            "one_thread_value": 1024 / dt1, "speedup_over_one_thread": (n_sample / dt) / (1024 / dt1),
            "sample": f"{n_sample} seeded C2 instances, 200 iterations each, oracle/admm_banded_oracle.c "
                      f"(gcc -O3 -ffp-contract=off), {threads} OpenMP threads over instances, {dt:.1f} s "
                      f"(+ 1024 of them on 1 thread, {dt1:.1f} s)"}


def cpu_reference_baseline(cfg, threads, seconds=10.0):
    """The reference's C template for this configuration - `code_laxMPC_ADMM_C.c` instantiated with the C2 constants and
    compiled by `__graft_entry__.build()` where /root/reference exists (oracle/_ref/libbench_C2_lax.so, gcc -O3 like the
    toolbox's mex build) - called once per instance, instances spread over `threads` host threads.  None if the object is
    not there."""
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "_ref", "libbench_C2_lax.so")
    if not os.path.exists(so):
        return None
    import ctypes as C
    from concurrent.futures import ThreadPoolExecutor
    import numpy as np
    from spcies_amd import benchmarks
    n, m, dim = cfg.sys.n, cfg.sys.m, cfg.param.N * (cfg.sys.n + cfg.sys.m)
    fn = C.CDLL(so).laxMPC_ADMM
    fn.restype = None

    class Sol(C.Structure):
        _fields_ = [("z", C.c_double * dim), ("v", C.c_double * dim), ("lam", C.c_double * dim), ("t", C.c_double * 4)]

    def work(args):
        x0, xr, ur = args
        sol, u, k, e = Sol(), (C.c_double * m)(), C.c_int(0), C.c_int(0)
        dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
        for i in range(x0.shape[0]):
            fn(dp(x0[i]), dp(xr[i]), dp(ur[i]), u, C.byref(k), C.byref(e), C.byref(sol))
        return k.value

    def run(count, nthreads):
        x0, xr, ur = benchmarks.sample_batch(cfg, count, seed=cfg.seed + 1)
        parts = [(x0[i::nthreads].copy(), xr[i::nthreads].copy(), ur[i::nthreads].copy()) for i in range(nthreads)]
        t = time.perf_counter()
        with ThreadPoolExecutor(nthreads) as ex:
            ks = list(ex.map(work, parts))
        assert all(k == cfg.solver_options["k_max"] for k in ks)
        return count / (time.perf_counter() - t)
    rate1 = run(512, 1)
    rate = run(64 * threads, threads)
    n_sample = int(min(max(rate * seconds, 64 * threads), 1 << 20))
    t = time.perf_counter()
    value = run(n_sample, threads)
    dt = time.perf_counter() - t
    return {"value": value, "unit": "solves/s", "cores": threads, "kind": "template", "one_thread_value": rate1,
            "sample": f"{n_sample} seeded C2 instances, 200 iterations each, the reference's generated laxMPC_ADMM solver "
                      f"(formulations/+laxMPC/code_laxMPC_ADMM_C.c instantiated for C2, gcc -O3), one call per instance from "
                      f"{threads} host threads, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--variant", default="auto", choices=["auto", "stream", "mfma", "mfma4"])
    ap.add_argument("--cpu-sample", type=int, default=0, help="instances of the CPU baseline (0: about 12 s of work, sized by a probe)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0: all host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    from spcies_amd import benchmarks, blob as blobmod, distributed as spdist
    from spcies_amd.solver import HipSolver

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    cfg = benchmarks.config("C2")
    # rank 0 factorises the controller once; the blob travels by one RCCL broadcast (xGMI)
    v = benchmarks.ingredients(cfg) if rank == 0 else None
    blob = blobmod.pack(v) if rank == 0 else None
    blob = spdist.broadcast_blob(blob, dev)
    solver = HipSolver(blob, device=local_rank)
    solver.set_variant(args.variant)
    variant = solver.variant

    B = args.batch
    # weak scaling: rank r owns global instances [r*B, (r+1)*B) of the seeded stream
    x0, xr, ur = spdist.shard_inputs(cfg, B, rank)
    tx0 = torch.from_numpy(x0).to(dev)
    txr = torch.from_numpy(xr).to(dev)
    tur = torch.from_numpy(ur).to(dev)
    tu = torch.empty((B, cfg.sys.m), dtype=torch.float64, device=dev)
    tk = torch.empty(B, dtype=torch.int32, device=dev)
    te = torch.empty(B, dtype=torch.int32, device=dev)
    solver.reserve(B)
    stream = torch.cuda.current_stream(dev).cuda_stream  # the stream the kernel is launched on

    def step():
        solver.solve_device(tx0, txr, tur, tu, tk, te, stream=stream)

    for _ in range(args.warmup):
        step()
    # Untimed queue priming: the HIP runtime grows its per-queue signal / kernarg pools the first time more
    # launches are in flight than ever before, and that one-off growth (tens of ms, tools/wall_jitter.py) would
    # otherwise land inside the timed region, which queues all K launches back to back.
    torch.cuda.synchronize(dev)
    for _ in range(args.steps):
        step()

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    fence()
    dt = time.perf_counter() - t0
    kernel_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream, over the timed region
    tdt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tdt, op=dist.ReduceOp.MAX)
    dt = float(tdt.item())

    # sanity on what was computed inside the timed region
    k_ok = bool((tk == 200).all().item()) and bool((te == -1).all().item())
    u_host = tu[:64].cpu().numpy()

    if rank == 0:
        out = {
            "metric": "MPC solves/sec (whole node), laxMPC-ADMM 12-state N=15, 200 iters",
            "value": world * B * args.steps / dt,
            "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: laxMPC-ADMM, 12-state osc-masses (n=12, m=2), N=15, rho=15, tol=0, "
                                   "k_max=200, batch=65536 random x0 / per-instance (xr, ur) per GPU",
                       "batch_per_gpu": B, "variant": variant, "all_k_200_eflag_-1": k_ok},
        }
        secs = kernel_ms * 1e-3
        if variant in ("mfma", "mfma4"):
            ach = FLOP_PER_SOLVE * B / secs / 1e12
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic_from_profile(variant),
                               "kernel_ms": kernel_ms, "flop_per_solve": FLOP_PER_SOLVE}
        else:
            # STREAM variant: state is streamed through HBM by design; algorithmic bytes are only the I/O
            ach = IO_BYTES_PER_SOLVE * B / secs / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": ach / PEAK_HBM_GBS, "traffic": traffic_from_profile("stream"), "kernel_ms": kernel_ms,
                               "algorithmic_bytes_per_solve": IO_BYTES_PER_SOLVE}
        if world == 1 and not args.no_cpu_baseline:
            if v is None:
                v = benchmarks.ingredients(cfg)
            threads = args.cpu_threads or host_threads()
            port = cpu_baseline(cfg, v, args.cpu_sample, threads)
            # The baseline is the port.  The reference's C template instantiated for C2 is timed next to it when its object is
            # there; it is NOT a reference build (MATLAB, which prints the constants of a generated solver, is absent: the
            # constants block is ours), so it rides along as information and does not change `kind`.
            tmpl = cpu_reference_baseline(cfg, threads)
            if tmpl is not None:
                port["reference_template"] = {"value": tmpl["value"], "one_thread_value": tmpl["one_thread_value"],
                                              "cores": tmpl["cores"], "note": tmpl["sample"] + "; constants printed by this "
                                              "repository's generator under the reference's dec_var.m rules, not by MATLAB"}
            out["cpu_baseline"] = port
            # the same run also re-checks the GPU result against the oracle on the first 64 instances
            from oracle import oracle
            uo, *_ = oracle.admm_banded_batch(v, x0[:64], xr[:64], ur[:64], want_sol=False)
            out["config"]["max_abs_du_vs_oracle_first64"] = float(np.abs(u_host - uo).max())
        print(json.dumps(out), flush=True)
    solver.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
