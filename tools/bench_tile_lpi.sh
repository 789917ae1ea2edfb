#!/bin/bash
# TILE variant: kernel time against the lane split (run on the GPU box).  usage: tools/bench_tile_lpi.sh <config> <B> [lpi ...]
CFG=${1:-C5_soc}; B=${2:-65536}; shift 2
for L in ${@:-4 8 16 32 64}; do
  SPCIES_TILE_LPI=$L timeout 600 python - "$CFG" "$B" "$L" <<'PY'
import sys, json, numpy as np
sys.path.insert(0, ".")
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
name, B, L = sys.argv[1], int(sys.argv[2]), sys.argv[3]
cfg = benchmarks.config(name); s = HipSolver(benchmarks.ingredients(cfg)); s.set_variant("tile")
x0, xr, ur = benchmarks.sample_batch(cfg, B)
extra = (cfg.param.r,) if cfg.formulation == "ellipMPC" else ()
s(x0[:256], xr[:256], ur[:256], *extra, want_sol=False)
u, k, e, sol = s(x0, xr, ur, *extra, want_sol=False)
print(json.dumps(dict(config=name, B=B, lpi=L, kernel_ms=round(sol.solve_time, 2), solves_per_s=round(B / sol.solve_time * 1e3), k=np.unique(k).tolist()[:3])))
PY
done
