"""Child of tests/test_no_library_gemm.py (fresh process, no torch, SPCIES_HIP_RTC=0): HMPC controllers whose FUSED kernel cannot be
built resolve, under AUTO, to the library's own reference-order kernels - never to rocBLAS.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SPCIES_HIP_RTC"] = "0"
import numpy as np

from oracle import oracle
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver

out = {}
for name, N in (("C1_HMPC_SADMM", 7), ("C1_HMPC_nosplit", 7), ("C1_HMPCcc_nosplit", None)):
    cfg = benchmarks.config(name)
    if N:
        cfg.param.N = N  # not among the build-time shapes of hmpc_fused.hip: FUSED would need hiprtc
    v = benchmarks.ingredients(cfg)
    s = HipSolver(v)
    x0, xr, ur = benchmarks.sample_batch(cfg, 40)
    u, k, e, sol = s(x0, xr, ur)
    split = getattr(cfg, "submethod", "") == "split"
    O = oracle.admm_hmpc_batch(v, x0, xr, ur, sparse=True) if split else oracle.hmpc_dense_batch(v, x0, xr, ur)
    out[name] = {"variant": s.variant, "notes": s.notes[:200], "k_equal": bool(np.array_equal(k, O[1])), "du": float(np.abs(u - O[0]).max()),
                 "dz": float(np.abs(sol.z - O[3]).max())}
    s.close()
out["rocblas_mapped"] = any("rocblas" in line for line in open("/proc/self/maps"))
print(json.dumps(out), flush=True)
