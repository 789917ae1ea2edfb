"""GPU: the LDS form of the time-varying MFMA4R path on a few plant shapes against the oracle, under each of its switches - default (cooperative update
phase, S form), SPCIES_TVL_COOP=0 (one-lane update phase), -DSPCIES_TVL_SFORM=0 (triangles of Bi_l): max |du|, k differences, instances off by more than 1e-8.
The bisection that found the S form's first bug (the switch was defined below the kernels that test it).  usage: python tools/tvl_debug.py"""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch
torch.cuda.init()
from oracle import oracle
from spcies_amd import benchmarks
from spcies_amd.solver import HipSolver
from _cases import random_cfg
from test_time_varying import _perturbed_models
for (n, m, N, formulation, method) in [(20, 4, 6, "laxMPC", "ADMM"), (12, 6, 5, "equMPC", "ADMM"), (6, 2, 40, "laxMPC", "ADMM"), (8, 2, 30, "laxMPC", "ADMM"), (16, 2, 6, "laxMPC", "ADMM")]:
    cfg = random_cfg(n, m, N, seed=1400 + n)
    cfg.formulation, cfg.method = formulation, method
    cfg.param.T = np.diag(np.diag(cfg.param.T))
    vt = benchmarks.ingredients(cfg, time_varying=True)
    sysm, prm = cfg.sys, cfg.param
    LB = np.concatenate([np.ravel(sysm.LBx), np.ravel(sysm.LBu)]); UB = np.concatenate([np.ravel(sysm.UBx), np.ravel(sysm.UBu)])
    design = (np.asarray(sysm.A, float), np.asarray(sysm.B, float), np.diag(prm.Q).copy(), np.diag(prm.R).copy(), LB, UB)
    rng = np.random.default_rng(19 * n + m)
    B = 70
    x0 = 0.4 * rng.standard_normal((B, n)); xr = 0.1 * rng.standard_normal((B, n)); ur = 0.05 * rng.standard_normal((B, m))
    models = _perturbed_models(design, B)
    model, per = oracle.pack_tv_model(*models)
    O = oracle.admm_tv_batch(vt, x0, xr, ur, model, per)
    for env in [{}, {"SPCIES_TVL_COOP": "0"}, {"SPCIES_TVR_RTC_FLAGS": "-DSPCIES_TVL_SFORM=0"}]:
        for k in ("SPCIES_TVL_COOP", "SPCIES_TVR_RTC_FLAGS"): os.environ.pop(k, None)
        os.environ.update(env)
        with HipSolver(vt) as s:
            u, k, e, sol = s(x0, xr, ur, *models)
            var = s.variant
        print((n, m, N), env, var, "max|du|", np.abs(u - O[0]).max(), "k diff", np.abs(k.astype(int) - O[1].astype(int)).max(), "per-instance bad", int((np.abs(u - O[0]).max(axis=1) > 1e-8).sum()), flush=True)
