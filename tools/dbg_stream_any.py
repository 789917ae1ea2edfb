import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
from _cases import random_cfg
n, m, N, form = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
cfg = random_cfg(n, m, N, seed=1700 + n); cfg.formulation = form
v = benchmarks.ingredients(cfg)
rng = np.random.default_rng(5 * n + m); B = 70
x0, xr, ur = 0.5 * rng.standard_normal((B, n)), 0.1 * rng.standard_normal((B, n)), 0.05 * rng.standard_normal((B, m))
O = oracle.admm_banded_batch(v, x0, xr, ur)
with HipSolver(v) as s:
    print("auto:", s.variant, s.notes[:200])
    s.set_variant("stream")
    for rep in range(2):
        u, k, e, sol = s(x0, xr, ur)
        bad = np.nonzero(k != O[1])[0]
        print("rep", rep, "bad e:", int((e != O[2]).sum()), "bad k:", len(bad), bad[:10], "k gpu", k[bad[:6]], "k ora", O[1][bad[:6]], "e gpu", e[bad[:6]], "du max", np.abs(u - O[0]).max(), "dz", np.abs(sol.z - O[3]).max())
    u, k, e, _ = s(x0, xr, ur, want_sol=False)
    print("nosol bad k:", (k != O[1]).sum())
