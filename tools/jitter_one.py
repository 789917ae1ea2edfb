"""Kernel time of N consecutive solves of one configuration (run-to-run spread).  usage: python tools/jitter_one.py <config> <B> <variant> [N] [--torch]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if "--torch" in sys.argv:
    import torch
    torch.zeros(8, device="cuda")
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
name, B, variant = sys.argv[1], int(sys.argv[2]), sys.argv[3]
N = int(sys.argv[4]) if len(sys.argv) > 4 and sys.argv[4].isdigit() else 30
cfg = benchmarks.config(name); s = HipSolver(benchmarks.ingredients(cfg)); s.set_variant(variant)
x0, xr, ur = benchmarks.sample_batch(cfg, B)
extra = (cfg.param.r,) if (cfg.formulation == "ellipMPC" and getattr(cfg, "submethod", "") == "soc") else ()
t = []
for i in range(N):
    u, k, e, sol = s(x0, xr, ur, *extra, want_sol=False)
    t.append(round(sol.solve_time, 2))
print(name, B, variant, "kernel_ms", t)
