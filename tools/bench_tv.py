"""Time-varying laxMPC-ADMM at the C2 shape (one model per instance): host-buffer call, kernel time from the
timing record (update phase + iteration).  usage: python tools/bench_tv.py [B]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
cfg = benchmarks.config("C2_lax")
vt = benchmarks.ingredients(cfg, time_varying=True)
s = HipSolver(vt)
sysm, prm = cfg.sys, cfg.param
LB = np.concatenate([np.ravel(sysm.LBx), np.ravel(sysm.LBu)]); UB = np.concatenate([np.ravel(sysm.UBx), np.ravel(sysm.UBu)])
rng = np.random.default_rng(5)
j = lambda a, sc: np.asarray(a, float)[None] * (1.0 + sc * (2 * rng.random((B,) + np.shape(a)) - 1))
models = (j(sysm.A, 0.02), j(sysm.B, 0.02), j(np.diag(prm.Q), 0.02), j(np.diag(prm.R), 0.02), j(LB, 0.05), j(UB, 0.05))
x0, xr, ur = benchmarks.sample_batch(cfg, B)
s(x0[:256], xr[:256], ur[:256], *[a[:256] for a in models], want_sol=False)
u, k, e, sol = s(x0, xr, ur, *models, want_sol=False)
print(json.dumps(dict(config="C2_lax time-varying, one model per instance", B=B, variant=s.variant,
                      kernel_ms=round(sol.solve_time, 2), solves_per_s=round(B / sol.solve_time * 1e3), k_unique=np.unique(k).tolist()[:3])))
