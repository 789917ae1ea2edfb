"""Worst differences of the time-varying MFMA4R solvers against the oracle (GPU box): python3 tools/tv_margin.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from oracle import oracle
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
import test_time_varying as T
for name, fista in (("C2_lax", False), ("C2_equ", False), ("C1_lax", False), ("C2_lax_FISTA", True), ("C2_equ_FISTA", True)):
    cfg, v, vt, design = T._setup(name)
    s = HipSolver(vt)
    B = 160
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    models = T._perturbed_models(design, B)
    model, per = oracle.pack_tv_model(*models)
    O = (oracle.fista_tv_batch if fista else oracle.admm_tv_batch)(vt, x0, xr, ur, model, per)
    u, k, e, sol = s(x0, xr, ur, *models)
    print(f"{name}: variant {s.variant} k equal {np.array_equal(k, O[1])} max|du| {np.abs(u - O[0]).max():.2e} max|dz| {np.abs(sol.z - O[3]).max():.2e} max|lambda| {np.abs(O[-1]).max():.2e}")
    s.close()
