"""Per-stage differences of the ADMM MFMA4R variant (admm_r_kernel.inc) against the oracle: python tools/dbg_admm_r.py [config] [k_max] [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
name = sys.argv[1] if len(sys.argv) > 1 else "C1_lax"
kmax = int(sys.argv[2]) if len(sys.argv) > 2 else 1
B = int(sys.argv[3]) if len(sys.argv) > 3 else 20
cfg = benchmarks.config(name)
cfg.solver_options.update(k_max=kmax)
v = benchmarks.ingredients(cfg)
s = HipSolver(v); s.set_variant("mfma4r")
x0, xr, ur = benchmarks.sample_batch(cfg, B)
u, k, e, sol = s(x0, xr, ur)
O = oracle.admm_banded_batch(v, x0, xr, ur)
n, m, N = cfg.sys.n, cfg.sys.m, cfg.param.N
nm = n + m
print("k", k[:8], O[1][:8], "e", e[:4], O[2][:4], "du", np.abs(u - O[0]).max())
for nme, a, b in (("z", sol.z, O[3]), ("v", sol.v, O[4]), ("lam", sol.lam, O[5])):
    d = np.abs(a - b).max(axis=0)
    st0 = d[:m]; rest = d[m:]
    full = rest[: (len(rest) // nm) * nm].reshape(-1, nm)
    print(nme, "stage0 u:", np.array2string(st0, precision=2), "stages 1..:", np.array2string(full.max(axis=1), precision=2), "tail:", np.array2string(rest[full.size:], precision=2))
u2, k2, e2, _ = s(x0, xr, ur, want_sol=False)
print("nosol equal:", np.array_equal(u, u2), np.array_equal(k, k2))
