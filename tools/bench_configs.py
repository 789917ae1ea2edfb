#!/usr/bin/env python3
"""Kernel timings of the BASELINE.json configurations other than the headline one (run on the GPU box).

    python tools/bench_configs.py            # C2 (all ADMM variants), C3 (equMPC-FISTA), C4 (MPCT-EADMM shard)

Inputs resident in HBM, hipEvents on the launch stream (spcies_hip_time_device), 3 launches after 1 warm-up.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver


def run(name, B, variant=None, reps=3):
    cfg = benchmarks.config(name)
    if cfg.formulation in ("ellipMPC", "HMPC"):
        return run_ex(cfg, name, B, variant)
    v = benchmarks.ingredients(cfg)
    s = HipSolver(v)
    if variant:
        s.set_variant(variant)
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(a).to(dev)
    tx0, txr, tur = t(x0), t(xr), t(ur)
    tu = torch.empty((B, cfg.sys.m), dtype=torch.float64, device=dev)
    tk = torch.empty(B, dtype=torch.int32, device=dev)
    te = torch.empty(B, dtype=torch.int32, device=dev)
    s.reserve(B)
    st = torch.cuda.current_stream(dev).cuda_stream
    s.time_device(tx0, txr, tur, tu, tk, te, stream=st, reps=1)
    ms = s.time_device(tx0, txr, tur, tu, tk, te, stream=st, reps=reps)
    out = dict(config=name, formulation=cfg.formulation, method=cfg.method, n=cfg.sys.n, m=cfg.sys.m, N=cfg.param.N,
               B=B, variant=s.variant, kernel_ms=round(ms, 3), solves_per_s=round(B / ms * 1e3),
               k_unique=np.unique(tk.cpu().numpy()).tolist()[:4])
    print(json.dumps(out), flush=True)
    s.close()


def run_ex(cfg, name, B, variant=None):
    """Solvers with a 6-field record (_ex entry points): host-buffer call, kernel time from the timing record."""
    v = benchmarks.ingredients(cfg)
    s = HipSolver(v)
    if variant:
        s.set_variant(variant)
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    extra = (cfg.param.r,) if (cfg.formulation == "ellipMPC" and getattr(cfg, "submethod", "") == "soc") else ()
    s(x0[:256], xr[:256], ur[:256], *extra, want_sol=False)
    s(x0, xr, ur, *extra, want_sol=False)  # first full-size call: scratch allocation, rocBLAS kernel selection for this shape
    times = []
    for _ in range(7 if B * cfg.param.N < 2e6 else 1):  # short kernels: several timed calls, the median is reported (the first
        # launches after an idle gap run at ramping clocks: one or two of them take 3-7x longer)
        u, k, e, sol = s(x0, xr, ur, *extra, want_sol=False)
        times.append(sol.solve_time)
    ms = float(np.median(times))
    print(json.dumps(dict(config=name, formulation=cfg.formulation, method=cfg.method, n=cfg.sys.n, m=cfg.sys.m,
                          N=cfg.param.N, B=B, variant=s.variant, kernel_ms=round(ms, 3),
                          solves_per_s=round(B / ms * 1e3), k_unique=np.unique(k).tolist()[:4],
                          all_ms=[round(t, 2) for t in times])), flush=True)
    s.close()


if __name__ == "__main__":
    only = sys.argv[1:]
    jobs = [("C2", 65536, "mfma4"), ("C2", 65536, "mfma4g"), ("C2", 65536, "mfma"), ("C2", 65536, "stream"), ("C2_equ", 65536, "mfma4"), ("C2", 65536, "bsp"), ("C2_lax_gen", 65536, "bsp"), ("C2_lax_gen", 65536, "mfma4g"),
            ("C3", 262144, "mfma4g"), ("C3", 262144, "stream"), ("C2_lax_FISTA", 65536, "mfma4g"),
            ("C4", 131072, "mfma4g"), ("C4", 131072, "stream"), ("C5_soc", 65536, "bsp"), ("C5_soc", 65536, "tile"), ("C5_soc", 65536, "stream"),
            ("C5_HMPC_SADMM", 65536, "gemm"), ("C5_HMPC_SADMM_nosplit", 65536, "gemm"), ("C5_HMPC_SADMM_nosplit", 65536, "stream"), ("C2_cs", 65536, "tile"), ("C2_cs", 16384, "stream"), ("C2_ellip", 65536, "bsp"), ("C2_ellip", 65536, "stream"), ("C5_HMPC_SADMM", 65536, "tile"), ("C5_HMPC_SADMM", 16384, "stream")]
    for name, B, var in jobs:
        if only and name not in only:
            continue
        run(name, B, var)
