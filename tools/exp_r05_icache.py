"""Round-5 experiment (GPU box): does the unrolled iteration of the streamed-block kernels outgrow the instruction cache?
Time per stage and iteration of MPCT-EADMM (eadmm_r, C4 plant) and equMPC-FISTA (fista_r, C3 plant) against the horizon N: the
kernels are unrolled on the horizon, so their loop body grows with N (about 3.1 KB of code per stage for eadmm_r at n = 20, 2.0 KB per
stage for fista_r at n = 12) while the work per stage stays the same.  usage: python3 tools/exp_r05_icache.py [eadmm|fista] N [N ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver

fam = sys.argv[1]
for N in [int(a) for a in sys.argv[2:]]:
    cfg = benchmarks.config("C4" if fam == "eadmm" else "C3")
    cfg.param.N = N
    B = 131072 if fam == "eadmm" else 262144
    s = HipSolver(benchmarks.ingredients(cfg))
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    s(x0, xr, ur, want_sol=False)
    t = []
    for _ in range(4):
        u, k, e, sol = s(x0, xr, ur, want_sol=False)
        t.append(sol.solve_time)
    ms = float(np.median(t))
    it = int(k[0])
    print(f"{fam} N={N:3d} variant={s.variant} notes={s.notes[:60]!r} kernel_ms={ms:8.2f} k={it} us per (stage x iteration x 1e6 instances)={ms * 1e3 / N / it / (B / 1e6):.4f}", flush=True)
    s.close()
