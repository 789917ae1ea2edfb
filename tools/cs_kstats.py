#!/usr/bin/env python3
"""MPCT-cs at the C2 shape, tol = 0: how many instances reach an exact floating-point fixed point before k_max (per variant)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
from oracle import oracle

cfg = benchmarks.config("C2_cs")
v = benchmarks.ingredients(cfg)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
x0, xr, ur = benchmarks.sample_batch(cfg, B)
with HipSolver(v) as s:
    res = {}
    for variant in ("fused", "tile"):
        s.set_variant(variant)
        u, k, e, sol = s(x0, xr, ur)
        res[variant] = (u, k, e, sol)
        print(variant, "k<200:", int((k < 200).sum()), "of", B, "min k", int(k.min()), "flags", np.unique(e, return_counts=True))
    early = np.nonzero(res["fused"][1] < 200)[0][:64]
    if len(early):
        O = oracle.mpct_cs_batch(v, x0[early], xr[early], ur[early])
        f = res["fused"]
        print("early exits vs oracle at k = 200: max|du| %.2e max|dz| %.2e max|dv| %.2e max|dlam| %.2e  (oracle k: %s)" % (
            np.abs(f[0][early] - O[0]).max(), np.abs(f[3].z[early] - O[3]).max(), np.abs(f[3].v[early] - O[4]).max(),
            np.abs(f[3].lam[early] - O[5]).max(), np.unique(O[1])))
