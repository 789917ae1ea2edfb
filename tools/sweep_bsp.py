"""Generator knobs of the BSP variant (prefetch ring depth, fence spacing) on one configuration.
usage (GPU box): python tools/sweep_bsp.py [config] [B]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver

name = sys.argv[1] if len(sys.argv) > 1 else "C5_soc"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
cfg = benchmarks.config(name)
v = benchmarks.ingredients(cfg)
x0, xr, ur = benchmarks.sample_batch(cfg, B)
extra = (cfg.param.r,) if cfg.formulation == "ellipMPC" else ()
ref = None
for pf, seg in ((16, 4), (8, 4), (24, 4), (32, 4), (16, 2), (16, 8), (16, 1000), (32, 1000)):
    os.environ["SPCIES_BSP_PF"], os.environ["SPCIES_BSP_SEG"] = str(pf), str(seg)
    try:
        s = HipSolver(v)
        s.set_variant("bsp")
        s(x0[:256], xr[:256], ur[:256], *extra, want_sol=False)
        s(x0, xr, ur, *extra, want_sol=False)
        u, k, e, sol = s(x0, xr, ur, *extra, want_sol=False)
        if ref is None:
            ref = u
        print(json.dumps(dict(config=name, pf=pf, seg=seg, kernel_ms=round(sol.solve_time, 3), solves_per_s=round(B / sol.solve_time * 1e3),
                              du=float(np.abs(u - ref).max()))), flush=True)
        s.close()
    except Exception as ex:
        print(json.dumps(dict(pf=pf, seg=seg, error=str(ex)[:200])), flush=True)
