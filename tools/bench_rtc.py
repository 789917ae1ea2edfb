"""MFMA4 specialised at run time (hiprtc) against MFMA4G on shapes without a build-time instantiation.
usage: python tools/bench_rtc.py   (on the GPU box)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from types import SimpleNamespace
from spcies_amd import benchmarks, sp_utils
from spcies_amd.solver import HipSolver

def cfg_for(p, N):
    sysm = sp_utils.oscillating_masses_sys(p)
    c = benchmarks.config("C2")
    Q = np.diag(np.concatenate([15 * np.ones(p), np.ones(p)])); R = 0.1 * np.eye(sysm.m)
    T = np.diag(3 * np.diag(Q))
    return SimpleNamespace(name=f"osc{p}_N{N}", sys=sysm, param=SimpleNamespace(Q=Q, R=R, T=T, N=N), formulation="laxMPC", method="ADMM",
                           solver_options=dict(rho=15, k_max=200, tol=0.0), B=65536, seed=5)

dev = torch.device("cuda", 0)
for p, N in ((3, 20), (3, 26), (4, 12), (5, 8), (6, 10)):
    cfg = cfg_for(p, N)
    v = benchmarks.ingredients(cfg)
    B = 65536
    rng = np.random.default_rng(1)
    n, m = cfg.sys.n, cfg.sys.m
    x0 = 0.3 * rng.standard_normal((B, n)); xr = np.tile(benchmarks.tester_status(cfg.sys).xr, (B, 1)); ur = 0.5 * np.ones((B, m))
    tx0, txr, tur = (torch.from_numpy(a).to(dev) for a in (x0, xr, ur))
    tu = torch.empty((B, m), dtype=torch.float64, device=dev); tk = torch.empty(B, dtype=torch.int32, device=dev); te = torch.empty(B, dtype=torch.int32, device=dev)
    res = {}
    for variant in ("mfma4g", "mfma4"):
        t0 = time.perf_counter(); s = HipSolver(v); t_set = time.perf_counter() - t0  # create: includes the hiprtc compilation
        s.set_variant(variant)
        s.reserve(B)
        st = torch.cuda.current_stream(dev).cuda_stream
        s.time_device(tx0, txr, tur, tu, tk, te, stream=st, reps=1)
        ms = s.time_device(tx0, txr, tur, tu, tk, te, stream=st, reps=3)
        res[variant] = (round(ms, 3), round(t_set, 2), tu[:4].cpu().numpy().copy())
        s.close()
    du = float(np.abs(res["mfma4"][2] - res["mfma4g"][2]).max())
    print(json.dumps(dict(shape=f"n={n} m={m} N={N}", B=B, mfma4g_ms=res["mfma4g"][0], mfma4_rtc_ms=res["mfma4"][0],
                          create_s=res["mfma4"][1], speedup=round(res["mfma4g"][0] / res["mfma4"][0], 2), du=du)), flush=True)
