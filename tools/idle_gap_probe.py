"""Does an idle gap before a launch slow it down?  (C5 soc on its default variant; run on the GPU box.)  Measured: +7 % on the first
launch after 50 ms - 2 s of idle; the one 60-90 ms launch among the first four of a process is a one-off, not gap related."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
cfg = benchmarks.config("C5_soc"); s = HipSolver(benchmarks.ingredients(cfg))
x0, xr, ur = benchmarks.sample_batch(cfg, 65536)
for gap in (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.5, 0.0, 0.0, 2.0, 0.0, 0.0, 0.05, 0.05, 0.05):
    time.sleep(gap)
    u, k, e, sol = s(x0, xr, ur, cfg.param.r, want_sol=False)
    print(f"gap {gap:4.2f} s -> {sol.solve_time:6.2f} ms")
