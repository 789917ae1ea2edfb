#!/usr/bin/env python3
"""Repeat the MPCT-cs TILE solve at the C4 shape (tol = 0): every instance must run its 200 iterations, every time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
cfg = benchmarks.config("C4_cs")
v = benchmarks.ingredients(cfg)
x0, xr, ur = benchmarks.sample_batch(cfg, 65)
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 30):
    junk = torch.full((96 * 1024 * 1024,), float('nan'), dtype=torch.float64, device='cuda')  # poison the memory the next hipMalloc may hand out
    del junk
    torch.cuda.synchronize(); torch.cuda.empty_cache()
    with HipSolver(v) as s:
        s.set_variant("tile")
        for inner in range(3):
            u, k, e, sol = s(x0, xr, ur)
            if not (k == 200).all():
                bad += 1
                i = np.nonzero(k != 200)[0]
                print("rep", rep, inner, "instances", i, "k", k[i], "nan in z:", np.isnan(sol.z[i]).any(), "e", e[i])
print("bad runs:", bad)
