"""A cold launch and RUN_ONE_REPS (default 10) warm launches of one configuration (profiling target).
usage: python3 tools/run_one.py <config> <B> <variant>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
name, B, variant = sys.argv[1], int(sys.argv[2]), sys.argv[3]
tv = name.endswith(":tv")  # "<config>:tv": the time-varying solver of the configuration, one (jittered) model per instance
name = name[:-3] if tv else name
cfg = benchmarks.config(name); s = HipSolver(benchmarks.ingredients(cfg, time_varying=True) if tv else benchmarks.ingredients(cfg)); s.set_variant(variant)
x0, xr, ur = benchmarks.sample_batch(cfg, B)
extra = (cfg.param.r,) if (cfg.formulation == "ellipMPC" and getattr(cfg, "submethod", "") == "soc") else ()
if tv:
    sysm, prm = cfg.sys, cfg.param
    rng = np.random.default_rng(cfg.seed + 7)
    LB = np.concatenate([np.ravel(sysm.LBx), np.ravel(sysm.LBu)]); UB = np.concatenate([np.ravel(sysm.UBx), np.ravel(sysm.UBu)])
    jit = lambda a, sc: np.asarray(a, float)[None] * (1.0 + sc * (2 * rng.random((B,) + np.shape(a)) - 1))
    extra = (jit(sysm.A, 0.02), jit(sysm.B, 0.02), jit(np.diag(prm.Q), 0.02), jit(np.diag(prm.R), 0.02), jit(LB, 0.05), jit(UB, 0.05))
s(x0, xr, ur, *extra, want_sol=False)  # first launch: cold caches, first-touch of the scratch
times = []
for _ in range(int(os.environ.get("RUN_ONE_REPS", "10"))):
    u, k, e, sol = s(x0, xr, ur, *extra, want_sol=False)
    times.append(sol.solve_time)
print(name, B, variant, "kernel_ms (host timer around the launch) median", round(float(np.median(times)), 2), "of", len(times), "k", np.unique(k)[:3])
