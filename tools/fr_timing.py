"""Where a stage of fista_r_kernel spends its clocks (profiles/r05_C3_stage_timing.txt): the kernel is re-specialised with -DSPCIES_FR_TIMING=k (one clock
reading at the top of every stage and one at point k, wavefront 0 of workgroup 0; fista_r_kernel.inc) for each k in turn; the sums come back in u of
instance 0.  usage: python tools/fr_timing.py [config] (one tile per wavefront: B = 256 x 4 x 16)"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
name = sys.argv[1] if len(sys.argv) > 1 else "C3"
if len(sys.argv) > 2:  # child: one build
    import numpy as np
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver
    cfg = benchmarks.config(name)
    s = HipSolver(benchmarks.ingredients(cfg)); s.set_variant("mfma4r")
    B = 256 * 4 * 16
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    s(x0, xr, ur, want_sol=False)
    u, k, e, sol = s(x0, xr, ur, want_sol=False)
    N = cfg.param.N
    it = int(k[0])
    print(json.dumps(dict(point=int(sys.argv[2]), k=it, fwd_clk_per_stage=round(float(u[0, 0]) / (N * (it + 1)), 1), bwd_clk_per_stage=round(float(u[0, 1]) / (N * (it + 1)), 1),
                          kernel_ms=round(float(sol.solve_time), 3))))
    sys.exit(0)
for k in (0, 1, 2, 3, 4, 5, 6, 7):
    env = dict(os.environ, SPCIES_FR_PD=os.environ.get("SPCIES_FR_PD", "7"))
    env["SPCIES_FR_RTC_FLAGS"] = (os.environ.get("FR_EXTRA", "") + (" -DSPCIES_FR_TIMING=%d" % k if k else " -DSPCIES_FR_NOP=1")).strip()
    r = subprocess.run([sys.executable, os.path.abspath(__file__), name, str(k)], env=env, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
