"""A sweep of the LDS form of the time-varying MFMA4R path (admm_tvl_kernel.inc) over plant shapes against the oracle - one model per instance,
u, k, z at 1e-10 (scaled as tests/_cases.scaled_bar).  usage: python tools/tvl_sweep.py  -> one line per shape"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from oracle import oracle
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
from _cases import random_cfg, scaled_bar

shapes = [(17, 1, 4), (19, 2, 9), (24, 8, 5), (30, 2, 4), (16, 16, 5), (13, 7, 12), (9, 9, 20), (4, 14, 30), (21, 3, 16), (28, 4, 5), (20, 4, 20), (6, 2, 50)]
worst = 0.0
for (n, m, N) in shapes:
    for form, method in (("laxMPC", "ADMM"), ("equMPC", "ADMM"), ("laxMPC", "FISTA"), ("equMPC", "FISTA")):
        if form == "equMPC" and N * m < n:
            continue  # (the terminal equality makes W singular with fewer inputs than states over the horizon)
        cfg = random_cfg(n, m, N, seed=7000 + 31 * n + m)
        cfg.formulation, cfg.method = form, method
        cfg.param.T = np.diag(np.diag(cfg.param.T))
        if method == "FISTA":
            cfg.solver_options = dict(tol=1e-6, k_max=300)
        vt = benchmarks.ingredients(cfg, time_varying=True)
        sysm, prm = cfg.sys, cfg.param
        LB = np.concatenate([np.ravel(sysm.LBx), np.ravel(sysm.LBu)]); UB = np.concatenate([np.ravel(sysm.UBx), np.ravel(sysm.UBu)])
        rng = np.random.default_rng(n * 100 + m)
        B = 21
        j = lambda a, s: np.asarray(a, float)[None] * (1.0 + s * (2 * rng.random((B,) + np.shape(a)) - 1))
        models = (j(sysm.A, 0.02), j(sysm.B, 0.02), j(np.diag(prm.Q), 0.02), j(np.diag(prm.R), 0.02), j(LB, 0.05), j(UB, 0.05))
        x0 = 0.4 * rng.standard_normal((B, n)); xr = 0.1 * rng.standard_normal((B, n)); ur = 0.05 * rng.standard_normal((B, m))
        model, per = oracle.pack_tv_model(*models)
        try:
            with HipSolver(vt) as s:
                var = s.variant
                if var != "mfma4r":
                    print(json.dumps(dict(shape=[n, m, N], form=form, method=method, variant=var, note=s.notes[:120])), flush=True)
                    continue
                u, k, e, sol = s(x0, xr, ur, *models)
            O = (oracle.fista_tv_batch if method == "FISTA" else oracle.admm_tv_batch)(vt, x0, xr, ur, model, per)
            lam = O[4] if method == "FISTA" else O[5]
            bar = scaled_bar(np.maximum(1.0, np.abs(lam).max(axis=1, keepdims=True)))
            same = (k == O[1])
            du = float((np.abs(u - O[0]) / bar)[same].max()) if same.any() else -1.0
            dz = float((np.abs(sol.z - O[3]) / bar)[same].max()) if same.any() else -1.0
            worst = max(worst, du, dz)
            print(json.dumps(dict(shape=[n, m, N], form=form, method=method, variant=var, k_diff=int((~same).sum()), e_equal=bool(np.array_equal(e[same], O[2][same])),
                                  du_over_bar=round(du, 4), dz_over_bar=round(dz, 4), nan=bool(np.isnan(u).any()))), flush=True)
        except Exception as ex:
            print(json.dumps(dict(shape=[n, m, N], form=form, method=method, error=str(ex)[:200])), flush=True)
print(json.dumps(dict(worst_share_of_bar=worst)))
