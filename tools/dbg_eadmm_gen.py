import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from oracle import oracle
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
name = sys.argv[1] if len(sys.argv) > 1 else "C1_MPCT_nd"
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = benchmarks.config(name)
cfg.solver_options.update(k_max=kmax)
v = benchmarks.ingredients(cfg)
s = HipSolver(v); s.set_variant("mfma4r")
x0, xr, ur = benchmarks.sample_batch(cfg, 8)
u, k, e, sol = s(x0, xr, ur)
O = oracle.eadmm_mpct_batch(v, x0, xr, ur)
n, m, N = cfg.sys.n, cfg.sys.m, cfg.param.N
nm = n + m
for nme, a, b in (("z1", sol.z1, O[3]), ("z2", sol.z2, O[4]), ("z3", sol.z3, O[5])):
    d = np.abs(a - b).max(axis=0)
    if nme != "z2":
        d = d.reshape(N + 1, nm)
        print(nme, "per stage max:", np.array2string(d.max(axis=1), precision=2))
        print(nme, "per row max:", np.array2string(d.max(axis=0), precision=2))
    else:
        print(nme, np.array2string(d, precision=2))
print("k", k, O[1])
