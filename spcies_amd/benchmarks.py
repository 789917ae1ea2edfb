"""Benchmark / test configurations (BASELINE.json ``configs``).

C1 is the reference's own test instance (``tests/spcies_tester.m:90-116`` with the solver settings
of ``tests/test_laxMPC_ADMM.m:6-16``); C2 is the headline shape the metric is quoted on.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from . import sp_utils


def _weights(sys, terminal="dlqr"):
    p = sys.p
    Q = np.diag(np.concatenate([15.0 * np.ones(p), np.ones(p)]))
    R = 0.1 * np.eye(sys.m)
    _, P = sp_utils.dlqr(sys.A, sys.B, Q, R)
    T = np.diag(P.sum(axis=1)) if terminal == "diag" else P
    return Q, R, T


def tester_status(sys):
    """``status`` of ``tests/spcies_tester.m:114-116``: x = 0.02, ur = 0.5, xr the matching steady state."""
    x = 0.02 * np.ones(sys.n)
    ur = 0.5 * np.ones(sys.m)
    xr = np.linalg.solve(sys.A - np.eye(sys.n), -sys.B @ ur)
    return SimpleNamespace(x=x, xr=xr, ur=ur)


def config(name):
    """Returns ``SimpleNamespace(sys, param, formulation, method, solver_options, B, seed)``."""
    if name in ("C1", "C1_lax"):  # reference test instance, laxMPC-ADMM (diag T as tests/test_laxMPC_ADMM.m:15)
        sys = sp_utils.oscillating_masses_sys(3)
        Q, R, T = _weights(sys, "diag")
        return SimpleNamespace(name=name, sys=sys, param=SimpleNamespace(Q=Q, R=R, T=T, N=10), formulation="laxMPC",
                               method="ADMM", solver_options=dict(rho=15, k_max=5000, tol=1e-7), B=1, seed=1201)
    if name == "C1_lax_denseT":  # example_OscMass.m:50-53 (dense T = P_dlqr)
        c = config("C1")
        c.name = name
        c.param.T = _weights(c.sys, "dlqr")[2]
        return c
    if name == "C1_equ":  # tests/test_equMPC_ADMM.m:6-14
        c = config("C1")
        c.name, c.formulation = name, "equMPC"
        return c
    if name in ("C2", "C2_lax"):  # headline: 12-state, N=15, 200 fixed iterations, B=65536
        sys = sp_utils.oscillating_masses_sys(6)
        Q, R, T = _weights(sys, "dlqr")
        return SimpleNamespace(name=name, sys=sys, param=SimpleNamespace(Q=Q, R=R, T=T, N=15), formulation="laxMPC",
                               method="ADMM", solver_options=dict(rho=15, k_max=200, tol=0.0), B=65536, seed=1202)
    if name == "C2_equ":
        c = config("C2")
        c.name, c.formulation = name, "equMPC"
        return c
    if name in ("C2_lax_N30", "C2_equ_N30", "C4_lax_ADMM", "C4_equ_ADMM"):
        # lax / equ ADMM past the register-resident MFMA4 kernel (more than 112 slab registers; n + m > 16): the 12-state plant at the
        # configs[2] horizon N = 30 and the 20-state plant of configs[3] at N = 20 - the shapes admm_r.hpp (MFMA4R) is for
        if name.startswith("C2"):
            c = config("C2" if "lax" in name else "C2_equ")
            c.param.N = 30
        else:
            sys = sp_utils.oscillating_masses_sys(10)
            Q, R, T = _weights(sys, "dlqr")
            c = SimpleNamespace(sys=sys, param=SimpleNamespace(Q=Q, R=R, T=T, N=20), formulation="laxMPC" if "lax" in name else "equMPC",
                                method="ADMM", solver_options=dict(rho=15, k_max=200, tol=0.0), B=65536, seed=1206)
        c.name = name
        return c
    if name.endswith("_gen"):  # C1_lax_gen, C1_equ_gen, C2_lax_gen: vector rho + one bound column per prediction step
        c = config(name[:-4])
        c.name = name
        rng = np.random.default_rng(77)
        n, m, N = c.sys.n, c.sys.m, c.param.N
        sysd = dict(vars(c.sys))
        wide = lambda a, sc: np.tile(np.ravel(a)[:, None], (1, N + 1)) * (1.0 + sc * rng.random((np.size(a), N + 1)))
        sysd.update(LBx=wide(c.sys.LBx, 0.2), UBx=wide(c.sys.UBx, 0.2), LBu=wide(c.sys.LBu, 0.3), UBu=wide(c.sys.UBu, 0.3))
        c.sys = SimpleNamespace(**sysd)
        dim = N * (n + m) - (0 if c.formulation == "laxMPC" else n)
        c.solver_options = dict(c.solver_options, rho=15.0 * (0.5 + rng.random(dim)))
        return c
    if name in ("C1_lax_FISTA", "C1_equ_FISTA"):  # tests/test_laxMPC_FISTA.m:6-15, tests/test_equMPC_FISTA.m:6-13
        c = config("C1")
        c.name, c.method, c.solver_options = name, "FISTA", dict(k_max=5000, tol=1e-7)
        c.formulation = "laxMPC" if "lax" in name else "equMPC"
        return c
    if name in ("C2_lax_FISTA", "C2_equ_FISTA", "C3"):  # C3: equMPC-FISTA, 12-state, N=30, 100 fixed iterations
        c = config("C2")
        c.name, c.method = name, "FISTA"
        c.formulation = "laxMPC" if "lax" in name else "equMPC"
        c.param.T = np.diag(np.diag(c.param.T))  # FISTA needs a diagonal T (compute_laxMPC_FISTA_ingredients.m:50-52)
        c.solver_options = dict(k_max=100, tol=0.0)
        if name == "C3":
            c.param.N, c.B, c.seed = 30, 262144, 1203
        return c
    if name in ("C4_lax_FISTA", "C4_equ_FISTA"):  # the 20-state plant of configs[3] at N = 20 under FISTA, 100 fixed iterations (time-varying runs past the register file)
        c = config("C4_lax_ADMM" if "lax" in name else "C4_equ_ADMM")
        c.name, c.method = name, "FISTA"
        c.param.T = np.diag(np.diag(c.param.T))
        c.solver_options = dict(k_max=100, tol=0.0)
        return c
    if name in ("C1_MPCT", "C4"):  # tests/test_MPCT_EADMM.m:6-17; C4: 20-state, N = 20, 200 fixed iterations
        sys = sp_utils.oscillating_masses_sys(3 if name == "C1_MPCT" else 10)
        Q, R, _ = _weights(sys, "diag")
        c = SimpleNamespace(name=name, sys=sys, param=SimpleNamespace(Q=Q, R=R, T=10 * Q, S=R, N=10),
                            formulation="MPCT", method="EADMM",
                            solver_options=dict(rho_base=2, rho_mult=20, k_max=5000, tol=1e-7), B=1, seed=1201)
        if name == "C4":
            c.param.N, c.B, c.seed = 20, 1048576, 1204
            c.solver_options.update(k_max=200, tol=0.0)
        return c
    if name in ("C1_MPCT_nd0", "C1_MPCT_nd", "C4_nd"):
        # the general-Q/R path of the generated solver (IS_DIAG == 0, compute_MPCT_EADMM_ingredients.m:142-154):
        # nd0: the tester's diagonal weights with force_diagonal off (same QP, so the reference test's z_opt still applies);
        # nd / C4_nd: coupled position / velocity weights and coupled inputs
        c = config("C4" if name == "C4_nd" else "C1_MPCT")
        c.name = name
        c.solver_options["force_diagonal"] = False
        if name != "C1_MPCT_nd0":
            n, m = c.sys.n, c.sys.m
            rng = np.random.default_rng(79)
            Mq, Mr = rng.standard_normal((n, n)), rng.standard_normal((m, m))
            c.param.Q = c.param.Q + 0.3 * (Mq @ Mq.T) / n
            c.param.R = c.param.R + 0.05 * (Mr @ Mr.T) / m
            c.param.T, c.param.S = 10 * c.param.Q, c.param.R
        if name == "C4_nd":
            c.B = 131072
        return c
    if name in ("C1_MPCT_cs", "C1_MPCT_cs_vec", "C2_cs", "C4_cs"):
        # tests/test_MPCT_ADMM.m:6-17 (rho: def_options_MPCT_ADMM_cs.m); C2 / C4 shapes with 200 fixed iterations.  At the C4
        # shape W = Aeq Hhat^-1 Aeq' has a condition number of 1e9: kept for the bit-exact STREAM variant only
        c = config("C4" if name == "C4_cs" else "C1_MPCT")
        c.name, c.method, c.submethod = name, "ADMM", "cs"
        c.solver_options = dict(k_max=5000, tol=1e-7)
        if name == "C4_cs":
            c.solver_options = dict(rho=0.05, k_max=200, tol=0.0)
            c.B = 131072
        if name == "C2_cs":
            c.sys = sp_utils.oscillating_masses_sys(6)
            Q, R, _ = _weights(c.sys, "diag")
            c.param = SimpleNamespace(Q=Q, R=R, T=10 * Q, S=R, N=15)
            c.solver_options = dict(rho=0.1, k_max=200, tol=0.0)
            c.B, c.seed = 65536, 1207
        if name.endswith("_vec"):
            dim = 2 * c.param.N * (c.sys.n + c.sys.m)
            c.solver_options["rho"] = 0.01 * (0.5 + np.random.default_rng(78).random(dim))
        return c
    if name in ("C1_ellip_vec", "C2_ellip_vec", "C1_ellip_inc", "C1_soc_inc"):
        # vector rho (compute_ellipMPC_ADMM_ingredients.m:67-77, 163-175): C1 keeps rho_N uniform (the reference's terminal Hessian
        # block T + diag(rho_N) P and the template's P diag(rho_N) then agree and the optimum is the QP's); C2 draws every entry,
        # 200 fixed iterations.  _inc: tightened constraints param.incBx / incBu (:101-128), growing along the horizon
        c = config(name[:-4])
        c.name = name
        n, m, N = c.sys.n, c.sys.m, c.param.N
        if name.endswith("_vec"):
            rho = 15.0 * (0.5 + np.random.default_rng(77).random(N * (n + m)))
            if name.startswith("C1"):
                rho[-n:] = 15.0
            c.solver_options["rho"] = rho
        else:
            c.param.incBx = 0.01 * np.outer(np.linspace(1.0, 2.0, n), np.arange(N + 1))
            c.param.incBu = 0.004 * np.outer(np.ones(m), np.arange(N + 1))
        return c
    if name in ("C1_ellip", "C2_ellip"):  # tests/test_ellipMPC_ADMM.m:6-21; C2: 12-state, N = 15, r = 0.5, 200 fixed iterations
        sys = sp_utils.oscillating_masses_sys(3 if name == "C1_ellip" else 6)
        Q, R, T = _weights(sys, "diag")
        st = tester_status(sys)
        c = SimpleNamespace(name=name, sys=sys, param=SimpleNamespace(Q=Q, R=R, T=T, N=10, P=np.eye(sys.n), c=st.xr, r=0.0),
                            formulation="ellipMPC", method="ADMM", submethod="",
                            solver_options=dict(rho=15, k_max=5000, tol=1e-7), B=1, seed=1206)
        if name == "C2_ellip":
            rng = np.random.default_rng(3)
            M = rng.standard_normal((sys.n, sys.n))
            c.param.N, c.param.r, c.B = 15, 0.5, 65536
            c.param.P = np.eye(sys.n) + 0.05 * (M @ M.T)  # a genuine (non-spherical) ellipsoid
            c.solver_options.update(k_max=200, tol=0.0)
        return c
    if name in ("C1_soc", "C5_soc"):  # tests/test_ellipMPC_ADMM_soc.m:8-25; C5: 12-state, N = 15, 200 fixed iterations
        sys = sp_utils.oscillating_masses_sys(3 if name == "C1_soc" else 6)
        Q, R, T = _weights(sys, "diag")
        st = tester_status(sys)
        c = SimpleNamespace(name=name, sys=sys, param=SimpleNamespace(Q=Q, R=R, T=T, N=10, P=np.eye(sys.n), c=st.xr, r=0.0),
                            formulation="ellipMPC", method="ADMM", submethod="soc",
                            solver_options=dict(rho=15, sigma=10, k_max=5000, tol_p=1e-7, tol_d=1e-7), B=1, seed=1205)
        if name == "C5_soc":
            c.param.N, c.param.r, c.B = 15, 0.5, 524288
            c.solver_options.update(k_max=200, tol_p=0.0, tol_d=0.0)
        return c
    if name.startswith("C1_HMPCcc"):
        # HMPC with coupled output constraints LBy <= E x + F u <= UBy (COUPLED_CONSTRAINTS, code_HMPC_ADMM_split_C.c:65, 233-283;
        # compute_HMPC_ADMM_split_ingredients.m:35-46, 147-175): positions, one output mixing a velocity into an input, the other input
        c = config(name.replace("HMPCcc", "HMPC"))
        c.name = name
        n, m = c.sys.n, c.sys.m
        E = np.zeros((5, n)); F = np.zeros((5, m))
        E[0, 0] = E[1, 1] = E[2, 2] = 1.0
        E[3, 3], F[3, 0], F[4, 1] = 0.2, 1.0, 1.0
        sysd = dict(vars(c.sys))
        sysd.update(E=E, F=F, LBy=np.array([-1.0, -1.0, -1.0, -0.8, -0.8]), UBy=np.array([0.3, 0.3, 0.3, 0.8, 0.8]))
        c.sys = SimpleNamespace(**sysd)
        c.solver_options = dict(c.solver_options, box_constraints=False)
        return c
    if name.startswith("C1_HMPC") or name.startswith("C5_HMPC"):
        # tests/test_HMPC_ADMM_s.m / test_HMPC_SADMM_s.m:6-22; C5: 12-state, N = 15, 200 fixed iterations
        sys = sp_utils.oscillating_masses_sys(3 if name.startswith("C1") else 6)
        Q, R, _ = _weights(sys, "diag")
        N = 10 if name.startswith("C1") else 15
        c = SimpleNamespace(name=name, sys=sys, formulation="HMPC", method="SADMM" if "SADMM" in name else "ADMM",
                            submethod="" if "nosplit" in name else "split",  # "": code_HMPC_ADMM_C.c (the default)
                            param=SimpleNamespace(N=N, w=3 * 1.627 * 0.2, Q=Q, R=R, Te=10 * N * Q, Th=10 * N * Q, Se=R,
                                                  Sh=0.5 * R),
                            solver_options=dict(rho=2, sigma=20, k_max=5000, tol_p=1e-7, tol_d=1e-7, sparse=True,
                                                use_soc="soc" in name, box_constraints=True),
                            B=1, seed=1205)
        if name.startswith("C5"):
            c.B = 524288
            c.solver_options.update(k_max=200, tol_p=0.0, tol_d=0.0)
        return c
    raise KeyError(name)


def hmpc_rotate_to_reference_phase(z, n, m, N, w):
    """The reference tests' HMPC ``z_opt`` (``tests/test_HMPC_ADMM_s.m:25``, ``test_HMPC_SADMM_s.m:25``,
    ``test_HMPC_ADMM.m:24``) counts the harmonic phase from the END of the horizon - it satisfies
    ``x_N = xe + xc`` - whereas the snapshot's code imposes ``x_N = xe + xs sin(wN) + xc cos(wN)``
    (``compute_HMPC_ADMM_split_ingredients.m:139``).  Same trajectory, harmonic coefficients rotated by ``wN``:
    ``xs' = xs cos(wN) - xc sin(wN)``, ``xc' = xs sin(wN) + xc cos(wN)`` (and the same for ``us, uc``)."""
    z = np.array(z, dtype=float)
    c, s = np.cos(w * N), np.sin(w * N)
    o = (N - 1) * (n + m) + m
    for a, b, wd in ((o + n, o + 2 * n, n), (o + 3 * n + m, o + 3 * n + 2 * m, m)):
        hs, hc = z[..., a:a + wd].copy(), z[..., b:b + wd].copy()
        z[..., a:a + wd] = hs * c - hc * s
        z[..., b:b + wd] = hs * s + hc * c
    return z


def sample_batch(cfg, B=None, seed=None, around_xr=None):
    """Seeded instances: ``x0 ~ U(-0.1, 0.1)^n``, ``ur = 0.5 + 0.1 U(-1, 1)^m``, ``xr`` the steady state of ``ur``.

    ``around_xr=s`` instead draws ``x0 = xr + U(-s, s)^n`` - starts close enough to the reference for the
    terminal equality of equMPC (``x_N = xr``) to be reachable inside the horizon."""
    B = cfg.B if B is None else B
    rng = np.random.default_rng(cfg.seed if seed is None else seed)
    sys = cfg.sys
    x0 = rng.uniform(-0.1, 0.1, size=(B, sys.n))
    ur = 0.5 + 0.1 * rng.uniform(-1.0, 1.0, size=(B, sys.m))
    xr = np.linalg.solve(sys.A - np.eye(sys.n), -(sys.B @ ur.T)).T.copy()
    if around_xr is not None:
        x0 = xr + x0 * (around_xr / 0.1)
    return x0, xr, ur


def ingredients(cfg, **solver_overrides):
    from .formulations import HMPC, MPCT, ellipMPC, laxMPC
    from .options import SpciesOptions
    so = dict(cfg.solver_options)
    so.update(solver_overrides)
    opt = SpciesOptions(formulation=cfg.formulation, method=cfg.method, submethod=getattr(cfg, "submethod", ""), options=so)
    ctrl = SimpleNamespace(sys=cfg.sys, param=cfg.param)
    hmpc = (HMPC.compute_HMPC_ADMM_split_ingredients if getattr(cfg, "submethod", "") == "split"
            else HMPC.compute_HMPC_ADMM_ingredients)
    fn = {("laxMPC", "ADMM"): laxMPC.compute_laxMPC_ADMM_ingredients,
          ("equMPC", "ADMM"): laxMPC.compute_equMPC_ADMM_ingredients,
          ("laxMPC", "FISTA"): laxMPC.compute_laxMPC_FISTA_ingredients,
          ("equMPC", "FISTA"): laxMPC.compute_equMPC_FISTA_ingredients,
          ("MPCT", "EADMM"): MPCT.compute_MPCT_EADMM_ingredients,
          ("MPCT", "ADMM"): MPCT.compute_MPCT_ADMM_cs_ingredients,
          ("ellipMPC", "ADMM"): (ellipMPC.compute_ellipMPC_ADMM_soc_ingredients if getattr(cfg, "submethod", "") == "soc"
                                 else ellipMPC.compute_ellipMPC_ADMM_ingredients),
          ("HMPC", "ADMM"): hmpc, ("HMPC", "SADMM"): hmpc}
    return laxMPC.add_engineering(fn[(cfg.formulation, cfg.method)](ctrl, opt), cfg.sys, opt)
