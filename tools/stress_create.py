"""Create / solve / destroy many solvers in one process (resource-leak and lifetime check of the run-time compiled paths)."""
import faulthandler, os, sys
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("STRESS_TORCH"):
    import torch  # its bundled ROCm user-space is then loaded before ours (mfma4_rtc.hpp: private hiprtc namespace)
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
names = sys.argv[2:] or ["C1_soc", "C5_soc"]
cfgs = [(benchmarks.config(n),) for n in names]
vs = [benchmarks.ingredients(c[0]) for c in cfgs]
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 40):
    cfg, v = cfgs[i % len(cfgs)][0], vs[i % len(cfgs)]
    x0, xr, ur = benchmarks.sample_batch(cfg, 40)
    extra = (cfg.param.r,) if getattr(cfg, "submethod", "") == "soc" else ()
    os.environ[f"STRESS_VAR_{i}"] = "x" * (50 + i)  # grows and reallocates the environment between creates
    s = HipSolver(v)
    u, k, e, sol = s(x0, xr, ur, *extra)
    print(i, cfg.name, s.variant, int(k[0]), flush=True)
    if i % 3 == 0:
        s.close()
