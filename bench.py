#!/usr/bin/env python3
"""bench.py - headline metric of BASELINE.json: MPC solves/sec (whole node), laxMPC-ADMM,
12-state oscillating masses, N=15, 200 iterations (config C2), B = 65 536 instances per GPU.

    python bench.py --gpus N --steps K --warmup W          # any N: for N > 1 it starts its own N rank processes
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W          # or under a launcher (RANK / WORLD_SIZE in the env)

One "step" = one pass of the hot path over one batch of B synthetic instances that are already
resident in HBM (x0, xr, ur in; u, k, e_flag out).  The batch shards trivially: every rank solves its
own B instances (weak scaling), the only collective is a one-time RCCL broadcast of the problem blob.
Rank 0 prints ONE JSON line.  At N = 1 the line also carries the other BASELINE.json configurations
(`configs`: C3, C4, C5 soc, C5 HMPC at their per-GPU shard), the host-buffer (PCIe-inclusive) rate and the
CPU baseline.
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_SOLVE = 27498 * 200          # 27 498 flop/iteration x 200 iterations
IO_BYTES_PER_SOLVE = 232              # compulsory HBM bytes per solve (per-instance reference)
PEAK_FP64_MFMA_TFLOPS = 78.6          # MI355X dense FP64 matrix peak = 256 CU x 128 flop/clk x 2.4 GHz (spec)
PEAK_HBM_GBS = 8000.0                 # MI355X_MICROARCH.md: 8 TB/s spec

# The other BASELINE.json configurations, at the per-GPU shard of SURVEY.md section 8d (C4, C5: 1/8 of the 8-GPU batch).
# flop = algorithmic flop per solve (SURVEY 8d / section 2.4 formulas; HMPC: the reference's default dense M1 product,
# 2 (dim + n_s)^2 per iteration, code_HMPC_ADMM_split_C.c:174-190), io = compulsory HBM bytes per solve.
EXTRA_CONFIGS = {
    "C3": dict(name="C3", B=262144, flop=47.5e3 * 100, io=232,
               what="configs[2]: equMPC-FISTA, 12-state, N=30, 100 iterations, batch=262144"),
    "C4": dict(name="C4", B=131072, flop=101e3 * 200, io=360,
               what="configs[3]: MPCT-EADMM, 20-state, N=20, 200 iterations, 1/8 shard (131072) of batch=1048576"),
    # the general-Q/R branch of the same solver (IS_DIAG == 0, code_MPCT_EADMM_C.c:184-217, 321-366): dense n x n / m x m inverse blocks instead of the
    # vector H3i - 2 N n^2 + 2 (N + 1)(n^2 + m^2) more flop per iteration than the diagonal branch
    "C4_nd": dict(name="C4_nd", B=131072, flop=133.5e3 * 200, io=360,
                  what="configs[3] shape with general (non-diagonal) Q, R: MPCT-EADMM IS_DIAG == 0, 20-state, N=20, 200 iterations, batch=131072"),
    # laxMPC-ADMM past the register-resident headline kernel: the C2 plant at the configs[2] horizon N = 30 (214 slab registers against MFMA4's
    # 112) - the MFMA4R variant of admm_r.hpp (w on the chip, blocks streamed); flop per iteration scales with the horizon (2 x C2's 27 498)
    "C2_N30": dict(name="C2_lax_N30", B=65536, flop=2 * 27498.0 * 200, io=232,
                   what="laxMPC-ADMM, C2 plant (n=12, m=2) at N=30, 200 iterations, batch=65536: past MFMA4's register file"),
    # ... and with vector rho + stage-wise bounds (code_laxMPC_ADMM_C.c:323-348, 490-568: no SCALAR_RHO, VAR_BOUNDS): the block program (BSP) does not fit
    # the LDS at this horizon; round 5: MFMA4R with the middle stages' row constants in the chunk stream (MFMA4G before: state through HBM)
    "C2_N30_gen": dict(name="C2_lax_N30_gen", B=65536, flop=2 * 27498.0 * 200, io=232,
                       what="laxMPC-ADMM with vector rho and VAR_BOUNDS, C2 plant (n=12, m=2) at N=30, 200 iterations, batch=65536"),
    "C5_soc": dict(name="C5_soc", B=65536, flop=23.4e3 * 200, io=240,
                   what="configs[4]a: ellipMPC-ADMM-soc, 12-state, N=15, 200 iterations, 1/8 shard (65536) of batch=524288"),
    "C5_hmpc": dict(name="C5_HMPC_SADMM", B=65536, flop=2.0 * 282 * 282 * 200, io=232,
                    what="configs[4]b: HMPC-SADMM split, 12-state, N=15, 200 iterations, 1/8 shard (65536) of batch=524288"),
    # the time-varying laxMPC-ADMM solver at the C2 shape, ONE MODEL PER INSTANCE (A, B, Q, R, LB, UB arrive with the
    # call: 232 doubles per instance next to x0, xr, ur) - the on-line factorisation plus 200 iterations.  Sixteen instances no longer
    # share their matrices, so this path is HBM-bound by design: `flop` as C2 plus the update phase, `io` = inputs + model + outputs.
    "C2_tv": dict(name="C2_lax", tv=True, B=65536, flop=27498.0 * 200 + 95e3, io=232 + 232 * 8,
                  what="SURVEY 8f rank 1: time-varying laxMPC-ADMM, C2 shape (n=12, m=2, N=15), one model per instance, 200 iterations, batch=65536"),
    # ... and its FISTA twin (code_laxMPC_FISTA_C.c with TIME_VARYING == 1): 100 iterations
    "C2_tv_fista": dict(name="C2_lax_FISTA", tv=True, B=65536, flop=23.8e3 * 100 + 95e3, io=232 + 232 * 8,
                        what="SURVEY 8f rank 1: time-varying laxMPC-FISTA, C2 shape (n=12, m=2, N=15), one model per instance, 100 iterations, batch=65536"),
    # ... at the plant of configs[3] (n = 20, m = 2, N = 20): past the register file - round 5: MFMA4R in its LDS form (admm_tvl_kernel.inc: the packed
    # triangles of S_l = Bi_l Bi_l' in the LDS, 34 KB per instance, one instance per SIMD); STREAM (27 k / 15 k solves/s) before, and no path at all before
    # round 5.  flop: 2 (2 n (n + m) + 3 n^2) + elementwise per stage and iteration (the reference's count), plus the on-line factorisation
    "C4_tv": dict(name="C4_lax_ADMM", tv=True, B=16384, flop=90e3 * 200 + 350e3, io=(20 + 20 + 2 + 2) * 8 + 506 * 8,
                  what="SURVEY 8f rank 1 at the configs[3] plant: time-varying laxMPC-ADMM, n=20, m=2, N=20, one model per instance, 200 iterations, batch=16384"),
    "C4_tv_fista": dict(name="C4_lax_FISTA", tv=True, B=16384, flop=78e3 * 100 + 350e3, io=(20 + 20 + 2 + 2) * 8 + 506 * 8,
                        what="SURVEY 8f rank 1 at the configs[3] plant: time-varying laxMPC-FISTA, n=20, m=2, N=20, one model per instance, 100 iterations, batch=16384"),
}


def traffic_from_profile(variant):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (newest round first;
    profiles/rNN_<variant>_pmc_summary.txt: separate FETCH_SIZE / WRITE_SIZE runs of this same command), corrected as
    MI355X_MICROARCH.md section HBM prescribes: FETCH_SIZE x 2 on gfx950, values in KB.  Returns (bytes, file) -
    a constant read from the repository, NOT a measurement of the run that prints it (`traffic_source` says so)."""
    for stem in (f"r05_C2_{variant}", f"r04_C2_{variant}", f"r03_C2_{variant}", f"r02_C2_{variant}", f"r02_{variant}", f"r01_{variant}"):
        path = os.path.join(ROOT, "profiles", f"{stem}_pmc_summary.txt")
        if not os.path.exists(path):
            continue
        fetch = write = None
        for line in open(path):
            if f"admm_{variant}_kernel" not in line and f"admm_{variant}u_kernel" not in line:  # (mfma4u: the unit-box form of MFMA4)
                continue
            val = float(line.split("mean=")[1].split()[0])
            if line.startswith("FETCH_SIZE"):
                fetch = val
            elif line.startswith("WRITE_SIZE"):
                write = val
        if fetch is not None and write is not None:
            return (2.0 * fetch + write) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


_PROFILE_TAG = {"C3": "C3", "C4": "C4", "C4_nd": "C4nd", "C2_N30": "C2N30", "C2_N30_gen": "C2N30gen", "C5_soc": "C5soc", "C5_hmpc": "C5hmpc", "C2_tv": "C2tv",
                "C2_tv_fista": "C2tvfista", "C4_tv": "C4tv", "C4_tv_fista": "C4tvfista"}


def design_traffic(key, variant):
    """HBM bytes per launch of a configuration's kernel from its committed rocprofv3 PMC passes (profiles/rNN_<config>_<variant>_pmc_summary.txt,
    newest round first; FETCH_SIZE x 2 + WRITE_SIZE, KB) - the DESIGN traffic, a constant read from the repository like `roofline.traffic`
    above.  Only a profile of the SAME variant counts."""
    for rnd in ("r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rnd}_{_PROFILE_TAG.get(key, key)}_{variant}_pmc_summary.txt")
        if not os.path.exists(path):
            continue
        fetch = write = None
        for line in open(path):
            if line.startswith("FETCH_SIZE"):
                fetch = float(line.split("mean=")[1].split()[0])
            elif line.startswith("WRITE_SIZE"):
                write = float(line.split("mean=")[1].split()[0])
        if fetch is not None and write is not None:
            return (2.0 * fetch + write) * 1024.0, os.path.relpath(path, ROOT)
    return None, None


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


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a launcher's environment starts its own N rank processes
# ---------------------------------------------------------------------------------------------------------------------
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _out_dir():
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    return d


def _tail(path, nbytes=2000):
    try:
        with open(path, "rb") as f:
            f.seek(0, os.SEEK_END)
            f.seek(max(0, f.tell() - nbytes))
            return f.read().decode(errors="replace")
    except OSError:
        return ""


def _stop(procs, grace=5.0):
    """Terminate the children this process started (exact PIDs, never a pattern), then kill what ignores the signal."""
    for p in procs:
        if p.poll() is None:
            p.terminate()
    deadline = time.monotonic() + grace
    for p in procs:
        try:
            p.wait(timeout=max(0.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()


def launch_ranks(args, argv):
    """Parent of a self-launched multi-GPU run.  It makes NO GPU call (a process that initialised the GPU must not be
    replaced or forked on this pool): it starts N fresh children of this script with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT set and WATCHES them: every rank's stderr goes to gpurun_out/rank<r>.err, rank 0's stdout (the
    JSON line) to gpurun_out/rank0.out; the first rank that exits non-zero - or the overall --launch-timeout - ends the
    others (they would otherwise sit in the RCCL rendezvous or a barrier until its watchdog fires), the failing rank's last
    stderr lines are relayed and the parent exits non-zero.  On success rank 0's line is relayed, with the one-process
    `multi` leg (spcies_hip_create_multi over the same N devices, run afterwards in a fresh child) merged in."""
    port = int(os.environ.get("MASTER_PORT", 0)) or _free_port()
    out_dir = _out_dir()
    procs, files = [], []
    out0 = os.path.join(out_dir, "rank0.out")
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", SPCIES_BENCH_SELF_LAUNCHED="1")
        ferr = open(os.path.join(out_dir, f"rank{r}.err"), "w")
        fout = open(out0, "w") if r == 0 else subprocess.DEVNULL
        files += [ferr] + ([fout] if r == 0 else [])
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=fout, stderr=ferr))
    deadline = time.monotonic() + args.launch_timeout
    failed, why = None, ""
    while True:
        codes = [p.poll() for p in procs]
        bad = [r for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed, why = bad[0], f"rank {bad[0]} exited with code {codes[bad[0]]}"
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failed = next(r for r, c in enumerate(codes) if c is None)
            why = f"--launch-timeout {args.launch_timeout:.0f} s passed; ranks still running: {[r for r, c in enumerate(codes) if c is None]}"
            break
        time.sleep(0.05)
    if failed is not None:
        _stop(procs)
    for f in files:
        f.close()
    if failed is not None:
        sys.stderr.write(f"bench.py: {why}; the other ranks were stopped.  Last lines of gpurun_out/rank{failed}.err:\n"
                         f"{_tail(os.path.join(out_dir, f'rank{failed}.err'))}\n")
        sys.stderr.flush()
        rc = procs[failed].returncode
        return abs(rc) if rc not in (None, 0) else 124
    out = open(out0).read()
    if not args.no_multi_leg and not args.dry_run:
        out = _merge_multi_leg(out, run_multi_leg_child(args))
    sys.stdout.write(out)
    sys.stdout.flush()
    return 0


def _merge_multi_leg(out, leg):
    lines = out.splitlines()
    for i in range(len(lines) - 1, -1, -1):
        if lines[i].startswith("{"):
            try:
                d = json.loads(lines[i])
            except ValueError:
                continue
            d["multi_launch"] = leg
            lines[i] = json.dumps(d)
            break
    return "\n".join(lines) + "\n"


# ---------------------------------------------------------------------------------------------------------------------
# "launch": "multi" - ONE process, N devices through spcies_hip_create_multi (what a mex or plain-C caller has: no launcher,
# no torch), host batch of N x B instances in page-locked buffers, split into contiguous shards by the library
# ---------------------------------------------------------------------------------------------------------------------
def run_multi_leg_child(args, timeout=240.0):
    """Run the multi leg in a FRESH child (the caller may hold a GPU context; it is never re-exec'ed) and return its JSON."""
    out_dir = _out_dir()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT",
                                                          "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, os.path.abspath(__file__), "--launch", "multi", "--gpus", str(args.gpus), "--steps", str(max(1, min(args.steps, 20))),
           "--warmup", str(max(1, min(args.warmup, 3))), "--batch", str(args.batch)]
    err_path = os.path.join(out_dir, "multi_leg.err")
    try:
        with open(err_path, "w") as ferr:
            p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=ferr, text=True)
            try:
                out, _ = p.communicate(timeout=timeout)
            except subprocess.TimeoutExpired:
                _stop([p])
                return {"error": f"multi leg did not finish within {timeout:.0f} s", "stderr_tail": _tail(err_path, 600)}
        if p.returncode != 0:
            return {"error": f"multi leg exited with code {p.returncode}", "stderr_tail": _tail(err_path, 600)}
        for line in reversed(out.splitlines()):
            if line.startswith("{"):
                return json.loads(line)
        return {"error": "multi leg printed no JSON line"}
    except Exception as ex:  # the headline line must survive a failing side leg
        return {"error": f"{type(ex).__name__}: {ex}"}


def run_multi(args):
    """One process, --gpus devices, spcies_hip_create_multi.  Wall-clock per call of spcies_hip_multi_solve_batch over the whole
    host batch (H2D + solve + D2H on every device side by side): a PCIe-inclusive rate, reported next to the RCCL leg's
    device-resident `value`, never instead of it.  --dry-run: the sharding arithmetic only (no device)."""
    import ctypes as C

    import numpy as np

    from spcies_amd import _lib, benchmarks, blob as blobmod
    lib = _lib.load()
    cfg = benchmarks.config("C2")
    G, B = args.gpus, args.batch
    total = G * B
    shards = []
    for g in range(G):
        lo, cnt = C.c_long(), C.c_long()
        _lib.check(lib.spcies_hip_shard_range(total, G, g, C.byref(lo), C.byref(cnt)))
        shards.append((lo.value, cnt.value))
    out = {"launch": "multi", "n_gpus": G, "batch_total": total, "shards": shards, "unit": "solves/s",
           "note": "one process, spcies_hip_create_multi over n_gpus devices, page-locked host buffers, wall time per "
                   "spcies_hip_multi_solve_batch call (H2D + solve + D2H), u / k / e_flag out"}
    if args.dry_run:
        out["dry_run"] = True
        assert sum(c for _, c in shards) == total and all(shards[i][0] + shards[i][1] == shards[i + 1][0] for i in range(G - 1))
        print(json.dumps(out), flush=True)
        return
    ndev = C.c_int(0)
    _lib.check(lib.spcies_hip_device_count(C.byref(ndev)))
    if ndev.value < G:
        raise SystemExit(f"--launch multi --gpus {G}: only {ndev.value} device(s) visible")
    v = benchmarks.ingredients(cfg)
    blob = blobmod.pack(v)
    t = time.perf_counter()
    mh = C.c_void_p()
    ids = (C.c_int * G)(*range(G))
    _lib.check(lib.spcies_hip_create_multi(blob, len(blob), ids, G, C.byref(mh)))
    out["create_s"] = time.perf_counter() - t
    n, m = cfg.sys.n, cfg.sys.m

    def pinned(shape, dtype):
        p = C.c_void_p()
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        _lib.check(lib.spcies_hip_host_alloc(nbytes, C.byref(p)))
        arr = np.frombuffer((C.c_char * nbytes).from_address(p.value), dtype=dtype).reshape(shape)
        return p, arr
    bufs = {}
    for name, shape, dt in (("x0", (total, n), np.float64), ("xr", (total, n), np.float64), ("ur", (total, m), np.float64),
                            ("u", (total, m), np.float64), ("k", (total,), np.int32), ("e", (total,), np.int32)):
        bufs[name] = pinned(shape, dt)
    from spcies_amd import distributed as spdist
    for g, (lo, cnt) in enumerate(shards):  # the instances rank g of the RCCL leg draws
        x0, xr, ur = spdist.shard_inputs(cfg, cnt, g)
        bufs["x0"][1][lo:lo + cnt], bufs["xr"][1][lo:lo + cnt], bufs["ur"][1][lo:lo + cnt] = x0, xr, ur
    dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
    cast = lambda name, ty: C.cast(bufs[name][0], ty)
    tim = _lib.Timing()

    def step():
        _lib.check(lib.spcies_hip_multi_solve_batch(mh, cast("x0", dp), cast("xr", dp), cast("ur", dp), 1, total, cast("u", dp),
                                                    cast("k", ip), cast("e", ip), None, None, None, C.byref(tim)))
    for _ in range(args.warmup):
        step()
    calls = []
    for _ in range(args.steps):
        t = time.perf_counter()
        step()
        calls.append(time.perf_counter() - t)
    # (a synchronous host call per step: the median is the steady rate - the first calls on fresh page-locked buffers and a cold
    # clock can take several times as long; mean and extremes are reported next to it)
    dt = float(np.median(calls)) * args.steps
    k, e = bufs["k"][1], bufs["e"][1]
    out.update(value=total * args.steps / dt, steps=args.steps, warmup=args.warmup, ms_per_step=dt / args.steps * 1e3,
               ms_per_step_mean=float(np.mean(calls)) * 1e3, ms_per_step_min=float(np.min(calls)) * 1e3, ms_per_step_max=float(np.max(calls)) * 1e3,
               last_call_ms={"h2d": tim.update_time, "solve": tim.solve_time, "d2h": tim.polish_time, "run": tim.run_time},
               all_k_200_eflag_m1=bool((k == 200).all() and (e == -1).all()))
    hits, misses = C.c_long(), C.c_long()
    lib.spcies_hip_rtc_cache_stats(C.byref(hits), C.byref(misses))
    out["rtc_cache"] = {"hits": hits.value, "misses": misses.value}
    # the first shard against a single-device handle on device 0 (same instances): identical results
    from spcies_amd.solver import HipSolver
    s0 = HipSolver(blob, device=0)
    lo, cnt = shards[-1]
    cnt = min(cnt, 4096)
    u1, k1, e1, _ = s0(bufs["x0"][1][lo:lo + cnt].copy(), bufs["xr"][1][lo:lo + cnt].copy(), bufs["ur"][1][lo:lo + cnt].copy(), want_sol=False)
    out["last_shard_equals_single_device"] = bool(np.array_equal(u1, bufs["u"][1][lo:lo + cnt]) and np.array_equal(k1, k[lo:lo + cnt]))
    s0.close()
    lib.spcies_hip_multi_destroy(mh)
    for p, _ in bufs.values():
        lib.spcies_hip_host_free(p)
    print(json.dumps(out), flush=True)


# ---------------------------------------------------------------------------------------------------------------------
# the other BASELINE.json configurations (N = 1 only), each with its own roofline and an in-region oracle check
# ---------------------------------------------------------------------------------------------------------------------
def _oracle_check(cfg, v, x0, xr, ur, u_gpu, k_gpu, count=32, model=None):
    """GPU result of the timed region against the oracle on the first `count` instances."""
    import numpy as np
    from oracle import oracle
    a = (x0[:count], xr[:count], ur[:count])
    if model is not None:  # time-varying solver: update phase + iteration of the oracle on the same per-instance models
        o = (oracle.fista_tv_batch if cfg.method == "FISTA" else oracle.admm_tv_batch)(v, *a, model[:count], True, want_sol=False)
    elif cfg.formulation == "HMPC":
        o = oracle.admm_hmpc_batch(v, *a, want_sol=False) if getattr(cfg, "submethod", "") == "split" \
            else oracle.hmpc_dense_batch(v, *a, want_sol=False)
    elif cfg.formulation == "ellipMPC" and getattr(cfg, "submethod", "") == "soc":
        o = oracle.admm_soc_batch(v, *a, cfg.param.r, want_sol=False)
    elif cfg.method == "FISTA":
        o = oracle.fista_banded_batch(v, *a, want_sol=False)
    elif cfg.method == "EADMM":
        o = oracle.eadmm_mpct_batch(v, *a, want_sol=False)
    else:
        o = oracle.admm_banded_batch(v, *a, want_sol=False)
    return {"instances": count, "max_abs_du": float(np.abs(u_gpu[:count] - o[0]).max()),
            "k_equal": bool(np.array_equal(k_gpu[:count], o[1]))}


def bench_config(spec, dev, steps, warmup, key=""):
    import torch
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver
    cfg = benchmarks.config(spec["name"])
    B = spec["B"]
    tv = bool(spec.get("tv"))
    v = benchmarks.ingredients(cfg, time_varying=True) if tv else benchmarks.ingredients(cfg)
    solver = HipSolver(v, device=dev.index)
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    t = lambda a: torch.from_numpy(a).to(dev)
    tx0, txr, tur = t(x0), t(xr), t(ur)
    tu = torch.empty((B, cfg.sys.m), dtype=torch.float64, device=dev)
    tk = torch.empty(B, dtype=torch.int32, device=dev)
    te = torch.empty(B, dtype=torch.int32, device=dev)
    extra = None
    if cfg.formulation == "ellipMPC" and getattr(cfg, "submethod", "") == "soc":
        extra = torch.full((1,), float(cfg.param.r), dtype=torch.float64, device=dev)
    extra_stride, model = 0, None
    if tv:  # one model per instance: the design model jittered by 2 % (A, B, Q, R) / 5 % (bounds), seeded
        import numpy as np
        rng = np.random.default_rng(cfg.seed + 7)
        sysm, prm = cfg.sys, cfg.param
        LB = np.concatenate([np.ravel(sysm.LBx), np.ravel(sysm.LBu)])
        UB = np.concatenate([np.ravel(sysm.UBx), np.ravel(sysm.UBu)])
        jit = lambda a, sc: np.asarray(a, float)[None] * (1.0 + sc * (2 * rng.random((B,) + np.shape(a)) - 1))
        model, extra_stride = solver._pack_model((jit(sysm.A, 0.02), jit(sysm.B, 0.02), jit(np.diag(prm.Q), 0.02), jit(np.diag(prm.R), 0.02),
                                                  jit(LB, 0.05), jit(UB, 0.05)), B)
        extra = t(model)
    solver.reserve(B)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def step():
        solver.solve_device_ex(tx0, txr, tur, tu, tk, te, extra=extra, extra_stride=extra_stride, stream=stream)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(steps):
        step()
    ev1.record()
    torch.cuda.synchronize(dev)
    ms = ev0.elapsed_time(ev1) / steps
    tf = spec["flop"] * B / (ms * 1e-3) / 1e12
    gbs = spec["io"] * B / (ms * 1e-3) / 1e9
    variant = solver.variant
    out = {"workload": spec["what"], "batch": B, "variant": variant, "steps": steps, "kernel_ms": ms,
           "solves_per_s": B / (ms * 1e-3),
           "roofline": {"bound": "hbm" if variant in ("mfma4g", "stream", "tile") else "mfma",  # (what binds the DESIGN; `achieved` is algorithmic flop either way)
                        "achieved": tf, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP64_MFMA_TFLOPS,
                        "flop_per_solve": spec["flop"], "algorithmic_io_gbs": gbs, "hbm_frac_algorithmic": gbs / PEAK_HBM_GBS},
           "oracle_check": _oracle_check(cfg, v, x0, xr, ur, tu[:32].cpu().numpy(), tk[:32].cpu().numpy(), model=model)}
    if tv and variant == "stream":  # STREAM re-reads the instance's factors in every iteration: bound by that design traffic (hbm_frac_design)
        out["roofline"]["bound"] = "hbm"
    # (MFMA4R for the time-varying solver keeps the factors in registers: no traffic inside the iteration; one instance per wavefront, the
    # dependent chain of 2 N block solves - issue / latency bound far below the matrix peak, which `frac` shows)
    traffic, src = design_traffic(key, variant)
    if traffic is not None:  # what the kernel really moves through HBM (design bytes, not algorithmic ones) against the 8 TB/s peak
        out["roofline"].update(traffic=traffic, traffic_source=src, hbm_frac_design=traffic / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS)
    solver.close()
    return out


def pcie_inclusive(solver, cfg, B, reps=5):
    """Host-buffer entry point (what a mex binds): pageable numpy buffers in and out through spcies_hip_solve_batch;
    wall time per call including H2D / D2H.  Never the bench `value`."""
    from spcies_amd import benchmarks
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    solver(x0, xr, ur, want_sol=False)
    best, tim = None, None
    for _ in range(reps):
        t = time.perf_counter()
        u, k, e, sol = solver(x0, xr, ur, want_sol=False)
        dt = time.perf_counter() - t
        if best is None or dt < best:
            best, tim = dt, sol
    return {"solves_per_s": B / best, "ms_per_call": best * 1e3, "batch": B,
            "h2d_ms": tim.update_time, "solve_ms": tim.solve_time, "d2h_ms": tim.polish_time,
            "note": "spcies_hip_solve_batch from pageable host buffers, best of %d calls, u/k/e_flag out" % reps}


# ---------------------------------------------------------------------------------------------------------------------
def run_rank(args):
    from datetime import timedelta

    import numpy as np
    import torch
    import torch.distributed as dist

    from spcies_amd import benchmarks, blob as blobmod, distributed as spdist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.fail_rank == rank:  # tests/test_bench_contract.py: a rank that dies at start-up
        sys.stderr.write(f"rank {rank}: --fail-rank\n")
        raise SystemExit(3)
    if args.fail_rank == -2 and rank == 0:  # ... and a rank that hangs before the rendezvous
        time.sleep(3600)
    dry = args.dry_run
    # --force-dist: the torch.distributed leg (RCCL rendezvous, blob broadcast, all_reduce, barriers, all_gather, teardown) also at
    # world size 1 - the code an 8-GPU run executes, on the one GPU a builder's box has
    use_dist = world > 1 or args.force_dist
    if use_dist and world == 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if dry:  # CPU rehearsal of the launch / rendezvous / timing plumbing (tests/test_bench_contract.py): no solver, no GPU
        dev = torch.device("cpu")
        if use_dist:
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=args.init_timeout))
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a HIP device (no CPU fallback)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: local GPU {local_rank} does not exist ({torch.cuda.device_count()} visible)")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
        if use_dist:
            # a finite rendezvous / collective timeout: a rank that died at start-up must not leave the others waiting for minutes
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=timedelta(seconds=args.init_timeout))

    def sync():
        if not dry:
            torch.cuda.synchronize(dev)

    cfg = benchmarks.config("C2")
    # rank 0 factorises the controller once; the blob travels by one RCCL broadcast (xGMI)
    v = benchmarks.ingredients(cfg) if rank == 0 else None
    blob = blobmod.pack(v) if rank == 0 else None
    blob = spdist.broadcast_blob(blob, None if dry else dev)
    ranks_seen = 1
    if use_dist:  # every rank adds one on its device: the sum is the number of ranks RCCL actually connected
        ones = torch.ones(1, dtype=torch.int32, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())

    B = args.batch
    x0, xr, ur = spdist.shard_inputs(cfg, B, rank)  # weak scaling: every rank draws its own B instances, seeded by (seed, rank)
    solver, variant = None, "dry-run"
    if not dry:
        from spcies_amd.solver import HipSolver
        solver = HipSolver(blob, device=local_rank)
        solver.set_variant(args.variant)
        variant = solver.variant
        tx0, txr, tur = (torch.from_numpy(a).to(dev) for a in (x0, xr, ur))
        tu = torch.empty((B, cfg.sys.m), dtype=torch.float64, device=dev)
        tk = torch.empty(B, dtype=torch.int32, device=dev)
        te = torch.empty(B, dtype=torch.int32, device=dev)
        solver.reserve(B)
        stream = torch.cuda.current_stream(dev).cuda_stream  # the stream the kernel is launched on

    def step():
        if not dry:
            solver.solve_device(tx0, txr, tur, tu, tk, te, stream=stream)

    for _ in range(args.warmup):
        step()
    # Untimed queue priming: the HIP runtime grows its per-queue signal / kernarg pools the first time more
    # launches are in flight than ever before, and that one-off growth (tens of ms, tools/wall_jitter.py) would
    # otherwise land inside the timed region, which queues all K launches back to back.
    sync()
    for _ in range(args.steps):
        step()

    def fence():
        sync()
        if use_dist:
            dist.barrier()
            sync()

    if not dry:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
    fence()
    t0 = time.perf_counter()
    if not dry:
        ev0.record()
    for _ in range(args.steps):
        step()
    if not dry:
        ev1.record()
    fence()
    dt_local = time.perf_counter() - t0
    kernel_ms = 0.0 if dry else ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream, over the timed region
    per_rank_ms = [dt_local / args.steps * 1e3]
    dt = dt_local
    if use_dist:
        tdt = torch.tensor([dt_local], dtype=torch.float64, device=dev)
        gathered = [torch.zeros_like(tdt) for _ in range(world)]
        dist.all_gather(gathered, tdt)
        per_rank_ms = [float(g.item()) / args.steps * 1e3 for g in gathered]
        dt = max(float(g.item()) for g in gathered)  # MAX over ranks

    # sanity on what was computed inside the timed region
    k_ok = None if dry else (bool((tk == 200).all().item()) and bool((te == -1).all().item()))
    u_host = None if dry else tu[:64].cpu().numpy()

    if rank == 0:
        out = {
            "metric": "MPC solves/sec (whole node), laxMPC-ADMM 12-state N=15, 200 iters",
            "value": None if dry else world * B * args.steps / dt,
            "unit": "solves/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[1]: laxMPC-ADMM, 12-state osc-masses (n=12, m=2), N=15, rho=15, tol=0, "
                                   "k_max=200, batch=65536 random x0 / per-instance (xr, ur) per GPU",
                       "batch_per_gpu": B, "variant": variant, "all_k_200_eflag_-1": k_ok,
                       "launch": "torchrun" if os.environ.get("TORCHELASTIC_RUN_ID") else
                                 ("self" if os.environ.get("SPCIES_BENCH_SELF_LAUNCHED") else "single"),
                       "process_group": (dist.get_backend() if use_dist else None)},
            "rccl_ranks_seen": ranks_seen, "per_rank_ms_per_step": per_rank_ms,
        }
        if dry:
            out["dry_run"] = True
        secs = kernel_ms * 1e-3
        if dry:
            pass
        elif variant in ("mfma", "mfma4"):
            ach = FLOP_PER_SOLVE * B / secs / 1e12
            traffic, src = traffic_from_profile(variant)
            io_gbs = IO_BYTES_PER_SOLVE * B / secs / 1e9
            out["roofline"] = {"bound": "mfma", "achieved": ach, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": ach / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic, "traffic_source": src,
                               "kernel_ms": kernel_ms, "flop_per_solve": FLOP_PER_SOLVE,
                               "hbm_frac": (traffic / secs / 1e9 / PEAK_HBM_GBS) if traffic else io_gbs / PEAK_HBM_GBS,
                               "hbm_frac_algorithmic": io_gbs / PEAK_HBM_GBS}
        else:
            # STREAM variant: state is streamed through HBM by design; algorithmic bytes are only the I/O
            ach = IO_BYTES_PER_SOLVE * B / secs / 1e9
            traffic, src = traffic_from_profile("stream")
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                               "frac": ach / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": src, "kernel_ms": kernel_ms,
                               "algorithmic_bytes_per_solve": IO_BYTES_PER_SOLVE}
        if world == 1 and not dry:
            if v is None:
                v = benchmarks.ingredients(cfg)
            # the same run re-checks the GPU result of the timed region against the oracle on the first 64 instances
            from oracle import oracle
            uo, *_ = oracle.admm_banded_batch(v, x0[:64], xr[:64], ur[:64], want_sol=False)
            out["config"]["max_abs_du_vs_oracle_first64"] = float(np.abs(u_host - uo).max())
            if not args.no_pcie:
                out["pcie_inclusive"] = pcie_inclusive(solver, cfg, B)
            if not args.no_configs:
                out["configs"] = {}
                for key, spec in EXTRA_CONFIGS.items():
                    if args.configs and key not in args.configs.split(","):
                        continue
                    t = time.perf_counter()
                    try:
                        out["configs"][key] = bench_config(spec, dev, args.config_steps, 2, key)
                    except Exception as ex:  # a failing side configuration must not lose the headline line
                        out["configs"][key] = {"error": f"{type(ex).__name__}: {ex}"}
                    out["configs"][key]["wall_s"] = time.perf_counter() - t
            if not args.no_cpu_baseline:
                threads = args.cpu_threads or host_threads()
                port = cpu_baseline(cfg, v, args.cpu_sample, threads)
                # The baseline is the port.  The reference's C template instantiated for C2 is timed next to it when its object
                # is there; it is NOT a reference build (MATLAB, which prints the constants of a generated solver, is absent:
                # the constants block is ours), so it rides along as information and does not change `kind`.
                tmpl = cpu_reference_baseline(cfg, threads)
                if tmpl is not None:
                    port["reference_template"] = {"value": tmpl["value"], "one_thread_value": tmpl["one_thread_value"],
                                                  "cores": tmpl["cores"], "note": tmpl["sample"] + "; constants printed by this "
                                                  "repository's generator under the reference's dec_var.m rules, not by MATLAB"}
                out["cpu_baseline"] = port
    # From here on nothing may lose the measured line: a peer that died after its last timed step makes the closing barrier
    # raise or time out, and the one-process leg is a child with its own failure modes - rank 0 prints in a `finally`.
    try:
        if solver is not None:
            solver.close()
        if use_dist:
            try:
                dist.barrier()
                dist.destroy_process_group()
            except Exception as ex:  # noqa: BLE001 - reported in the line, never instead of it
                if rank == 0:
                    out["teardown_error"] = f"{type(ex).__name__}: {ex}"[:300]
        if rank == 0:
            # Under a launcher (torchrun: the driver's N > 1 contract) rank 0 also reports the one-process leg: the other ranks
            # are past their last collective and exiting (a short pause lets them release their GPUs, so that the leg is not timed
            # against their teardown), the leg runs in a FRESH child process (this one keeps its idle context on GPU 0 and is
            # never replaced), bounded by a timeout; its failure is reported inside `multi_launch`, the line survives.
            # Self-launched runs do this in the parent instead (launch_ranks).
            if use_dist and not dry and not args.no_multi_leg and not os.environ.get("SPCIES_BENCH_SELF_LAUNCHED"):
                time.sleep(2.0)
                try:
                    out["multi_launch"] = run_multi_leg_child(args)
                except Exception as ex:  # noqa: BLE001
                    out["multi_launch"] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    finally:
        if rank == 0:
            print(json.dumps(out), flush=True)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)  # 0.73 s timed region at C2: long enough for an outside sampler to see the GPU busy
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=65536, help="instances per GPU")
    ap.add_argument("--variant", default="auto", choices=["auto", "stream", "mfma", "mfma4"])
    ap.add_argument("--cpu-sample", type=int, default=0, help="instances of the CPU baseline (0: about 12 s of work, sized by a probe)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0: all host cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the other BASELINE.json configurations (N = 1 runs them by default)")
    ap.add_argument("--configs", default="", help="comma-separated subset of " + ",".join(EXTRA_CONFIGS))
    ap.add_argument("--config-steps", type=int, default=10)  # timed launches per side configuration (after 2 warm-ups)
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-buffer (PCIe-inclusive) leg")
    ap.add_argument("--dry-run", action="store_true", help="CPU/gloo rehearsal of the multi-rank plumbing: no solver, no GPU")
    ap.add_argument("--launch", default="ranks", choices=["ranks", "multi"],
                    help="ranks: one process per GPU under torch.distributed (default, the driver's contract); multi: ONE process, "
                         "--gpus devices through spcies_hip_create_multi from page-locked host buffers")
    ap.add_argument("--no-multi-leg", action="store_true", help="N > 1: skip the one-process create_multi leg reported as `multi_launch`")
    ap.add_argument("--launch-timeout", type=float, default=540.0, help="self-launched ranks: overall limit in seconds before the parent stops them")
    ap.add_argument("--init-timeout", type=float, default=300.0, help="torch.distributed rendezvous / collective timeout in seconds")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (nccl = RCCL; gloo with --dry-run) and run every "
                    "collective of the N > 1 path also at --gpus 1")
    ap.add_argument("--self-launch", action="store_true", help="go through the self-launcher (fresh rank processes, watchdog) also at --gpus 1; implies --force-dist")
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)  # tests: this rank exits 3 before the rendezvous
    args = ap.parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.launch == "multi":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        run_multi(args)
        return
    if args.self_launch:
        args.force_dist = True
        if "--force-dist" not in argv:
            argv = argv + ["--force-dist"]
    if (args.gpus > 1 or args.self_launch) and "WORLD_SIZE" not in os.environ:
        # must be set before any rank touches the GPU (ROCr reads it at initialisation): the children inherit it
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
        raise SystemExit(launch_ranks(args, argv))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # before `import torch` / any HIP call of this process
    run_rank(args)


if __name__ == "__main__":
    main()
