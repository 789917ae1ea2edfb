"""A cold launch and RUN_ONE_REPS (default 10) warm launches of one configuration (profiling target).
usage: python3 tools/run_one.py <config> <B> <variant>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
name, B, variant = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cfg = benchmarks.config(name); s = HipSolver(benchmarks.ingredients(cfg)); s.set_variant(variant)
x0, xr, ur = benchmarks.sample_batch(cfg, B)
extra = (cfg.param.r,) if (cfg.formulation == "ellipMPC" and getattr(cfg, "submethod", "") == "soc") else ()
s(x0, xr, ur, *extra, want_sol=False)  # first launch: cold caches, first-touch of the scratch
times = []
for _ in range(int(os.environ.get("RUN_ONE_REPS", "10"))):
    u, k, e, sol = s(x0, xr, ur, *extra, want_sol=False)
    times.append(sol.solve_time)
print(name, B, variant, "kernel_ms (host timer around the launch) median", round(float(np.median(times)), 2), "of", len(times), "k", np.unique(k)[:3])
