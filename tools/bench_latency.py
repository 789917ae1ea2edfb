"""Wall time per call of the host-buffer entry point (what a mex binds) against the batch size: python3 tools/bench_latency.py [config]
Run it twice - SPCIES_HIP_SPIN_WAIT_US=0 (block in hipStreamSynchronize) and the default (poll first) - to see the wake-up cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
cfg = benchmarks.config(sys.argv[1] if len(sys.argv) > 1 else "C2")
s = HipSolver(benchmarks.ingredients(cfg))
out = []
for B in (16, 256, 4096, 65536):
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    for _ in range(5):
        s(x0, xr, ur, want_sol=False)
    t = []
    for _ in range(40):
        t0 = time.perf_counter()
        u, k, e, sol = s(x0, xr, ur, want_sol=False)
        t.append((time.perf_counter() - t0) * 1e3)
    out.append(f"B={B}: median {np.median(t):.3f} ms (min {np.min(t):.3f}); library timing h2d {sol.update_time:.3f} solve {sol.solve_time:.3f} d2h {sol.polish_time:.3f}")
print("SPCIES_HIP_SPIN_WAIT_US=%s  " % os.environ.get("SPCIES_HIP_SPIN_WAIT_US", "(default)") + " | ".join(out))
