"""ellipMPC, ADMM with the terminal ellipsoid imposed as a second-order cone ('soc' submethod) -
host-side (offline) ingredients.

Reference: ``formulations/+ellipMPC/compute_ellipMPC_ADMM_soc_ingredients.m:22-215``.  Decision
vector ``z = (u0, x1, u1, ..., x_N, t)`` (``dim = N(n+m)+1``; the last entry is the cone's radius
slack, fixed to ``r`` by the last equality row), slack ``s in R^{n+1}`` with
``C z + s = d`` and ``s`` in the SOC.  ``W = Gh Hh^-1 Gh'`` is factorised ``L D L'`` (from its
Cholesky factor) and stored as CSC of ``L - I`` plus ``Dinv``; ``-Gh Hh^-1``, ``-Hh^-1 Gh'`` and
``-Hh^-1`` are stored CSR (0-based indices here; the reference subtracts 1 when printing,
``cons_ellipMPC_ADMM_soc_C.m:98-110``).
"""
from __future__ import annotations

import numpy as np
import scipy.linalg as sla

from .. import sp_utils
from .laxMPC import _get, _is_diag, build_G


def _inc(param, name, rows, N):
    """``param.incBx`` / ``param.incBu`` as a [rows][N+1] array (compute_ellipMPC_ADMM_ingredients.m:104-123)."""
    x = _get(param, name)
    if x is None:
        return np.zeros((rows, N + 1))
    x = np.asarray(x, dtype=float)
    return x.reshape((rows, N + 1), order="F") if min(x.shape or (1,)) == 1 or x.ndim == 1 else x.reshape(rows, N + 1)


def compute_ellipMPC_ADMM_soc_ingredients(controller, opt):
    sys, param = _get(controller, "sys"), _get(controller, "param")
    A = np.asarray(_get(sys, "A"), dtype=float)
    B = np.asarray(_get(sys, "B"), dtype=float)
    n, m = B.shape
    nm = n + m
    N = int(_get(param, "N"))
    Q = np.asarray(_get(param, "Q"), dtype=float)
    R = np.asarray(_get(param, "R"), dtype=float)
    T = np.asarray(_get(param, "T"), dtype=float)
    P = np.asarray(_get(param, "P"), dtype=float)
    r = float(_get(param, "r", 1.0))
    if not (_is_diag(Q) and _is_diag(R)):
        raise ValueError("Spcies:ellipMPC:ADMM:non_diagonal - matrices Q and R must be diagonal")
    rho, sigma = float(opt.solver["rho"]), float(opt.solver["sigma"])
    dim = N * nm + 1
    H = np.zeros((dim, dim))
    H[:m, :m] = R
    for l in range(N - 1):
        o = m + l * nm
        H[o:o + n, o:o + n] = Q
        H[o + n:o + nm, o + n:o + nm] = R
    H[dim - 1 - n:dim - 1, dim - 1 - n:dim - 1] = T
    G = build_G(A, B, N, terminal=True)
    G = np.block([[G, np.zeros((G.shape[0], 1))], [np.zeros((1, G.shape[1])), np.ones((1, 1))]])
    n_eq = G.shape[0]
    P_half = np.real(sla.sqrtm(P))
    Cm = np.hstack([np.zeros((n + 1, dim - n - 1)),
                    np.block([[np.zeros((1, n)), -np.ones((1, 1))], [-P_half, np.zeros((n, 1))]])])
    n_s = n + 1
    LBx, UBx = np.ravel(_get(sys, "LBx")).astype(float), np.ravel(_get(sys, "UBx")).astype(float)
    LBu, UBu = np.ravel(_get(sys, "LBu")).astype(float), np.ravel(_get(sys, "UBu")).astype(float)
    # tightened constraints (:101-128): param.incBx [n][N+1], param.incBu [m][N+1] (a vector is reshaped column-major, as
    # MATLAB's reshape does); columns 2..N tighten stages 1..N-1, u_0 keeps the plain input bounds
    incBx, incBu = _inc(param, "incBx", n, N), _inc(param, "incBu", m, N)
    LB = np.concatenate([LBu] + [np.concatenate([LBx + incBx[:, i], LBu + incBu[:, i]]) for i in range(1, N)])
    UB = np.concatenate([UBu] + [np.concatenate([UBx - incBx[:, i], UBu - incBu[:, i]]) for i in range(1, N)])
    Hh = np.block([[H + sigma * np.eye(dim), np.zeros((dim, n_s))], [np.zeros((n_s, dim)), rho * np.eye(n_s)]])
    Gh = np.block([[G, np.zeros((n_eq, n_s))], [Cm, np.eye(n_s)]])
    Hhi = np.linalg.inv(Hh)
    W = Gh @ Hhi @ Gh.T
    Wc = np.linalg.cholesky(W).T
    wd = np.diag(Wc)
    L = Wc.T / wd[None, :]
    Dinv = 1.0 / (wd * wd)
    Lv, Lr, Lc, *_ = sp_utils.full2CSC(L - np.eye(L.shape[0]))
    csr = lambda M: sp_utils.full2CSR(M)[:3]
    v = dict(n=n, m=m, N=N, formulation="ellipMPC", method="ADMM", submethod="soc", terminal=True,
             dim=dim, n_s=n_s, n_eq=n_eq)
    v["A"] = A.copy()
    v["Q"], v["R"], v["T"] = -Q, -R, -T
    v["LB"], v["UB"] = LB, UB
    v["PhiP"] = np.linalg.inv(P_half) @ P
    v["rho"], v["rho_i"], v["sigma"], v["sigma_i"] = rho, 1.0 / rho, sigma, 1.0 / sigma
    v["L_val"], v["L_row"], v["L_col"], v["Dinv"] = Lv, Lr, Lc, Dinv
    v["GhHhi_val"], v["GhHhi_col"], v["GhHhi_row"] = csr(-Gh @ Hhi)
    v["HhiGh_val"], v["HhiGh_col"], v["HhiGh_row"] = csr(-Hhi @ Gh.T)
    v["Hhi_val"], v["Hhi_col"], v["Hhi_row"] = csr(-Hhi)
    v["k_max"] = int(opt.solver["k_max"])
    v["tol_p"], v["tol_d"] = float(opt.solver["tol_p"]), float(opt.solver["tol_d"])
    v["tol"] = v["tol_p"]
    v["rho_is_scalar"] = True
    v["r_default"] = r
    return v


def compute_ellipMPC_ADMM_ingredients(controller, opt):
    """ellipMPC, ADMM with the P-projection onto the terminal ellipsoid (no submethod) - SURVEY section 8f
    rank 2.  Reference: ``formulations/+ellipMPC/compute_ellipMPC_ADMM_ingredients.m:60-247``: the lax
    ingredients with ``H = Hz + rho blkdiag(I, P)``, stage-wise bounds ``LBu0/UBu0, LBz/UBz`` and the terminal
    constants ``P, P_half = sqrtm(P), Pinv_half = P^-1 P_half, c, r``.  Scalar or vector ``rho`` (:67-77, 163-175): with a vector
    the reference forms ``H = Hz + rho .* blkdiag(I, P)`` - MATLAB broadcasts the column ``rho`` over the ROWS, so the terminal block
    of ``H`` is ``T + diag(rho_N) P`` - restated as written (the template's terminal ``q_hat`` uses ``P diag(rho_N)``, so the two
    agree when ``rho_N`` is uniform).  ``force_vector_rho`` with a scalar expands it (the reference's line :69 names an undefined
    ``options`` there; the intent is taken)."""
    sys, param = _get(controller, "sys"), _get(controller, "param")
    A = np.asarray(_get(sys, "A"), dtype=float)
    B = np.asarray(_get(sys, "B"), dtype=float)
    n, m = B.shape
    nm = n + m
    N = int(_get(param, "N"))
    Q = np.asarray(_get(param, "Q"), dtype=float)
    R = np.asarray(_get(param, "R"), dtype=float)
    T = np.asarray(_get(param, "T"), dtype=float)
    P = np.asarray(_get(param, "P"), dtype=float)
    c = np.ravel(np.asarray(_get(param, "c"), dtype=float))
    r = float(_get(param, "r", 1.0))
    if not (_is_diag(Q) and _is_diag(R)):
        raise ValueError("Spcies:ellipMPC:ADMM:non_diagonal - matrices Q and R must be diagonal")
    dim = N * nm
    rho = opt.solver["rho"]
    scalar = np.ndim(rho) == 0 and not opt.solver.get("force_vector_rho", False)
    rho = float(rho) if scalar else (np.full(dim, float(rho)) if np.ndim(rho) == 0 else np.ravel(np.asarray(rho, dtype=float)))
    if not scalar and rho.shape != (dim,):
        raise ValueError("ellipMPC ADMM: a vector rho has N (n + m) entries")
    Hz = np.zeros((dim, dim))
    Hz[:m, :m] = R
    for l in range(N - 1):
        o = m + l * nm
        Hz[o:o + n, o:o + n] = Q
        Hz[o + n:o + nm, o + n:o + nm] = R
    Hz[dim - n:, dim - n:] = T
    P_half = np.real(sla.sqrtm(P))
    E = np.eye(dim)
    E[dim - n:, dim - n:] = P
    H = Hz + (rho * E if scalar else rho[:, None] * E)
    G = build_G(A, B, N, terminal=True)
    Hinv = np.linalg.inv(H)
    W = G @ Hinv @ G.T
    if not scalar:  # H (hence W) is not symmetric with a non-uniform rho_N; MATLAB's chol (:99) reads the upper triangle only
        W = np.triu(W) + np.triu(W, 1).T
    Wc = np.linalg.cholesky(W).T
    incBx, incBu = _inc(param, "incBx", n, N), _inc(param, "incBu", m, N)
    LBx, UBx = np.ravel(_get(sys, "LBx")).astype(float), np.ravel(_get(sys, "UBx")).astype(float)
    LBu, UBu = np.ravel(_get(sys, "LBu")).astype(float), np.ravel(_get(sys, "UBu")).astype(float)
    LBz = np.array([np.concatenate([LBx + incBx[:, i], LBu + incBu[:, i]]) for i in range(1, N)])
    UBz = np.array([np.concatenate([UBx - incBx[:, i], UBu - incBu[:, i]]) for i in range(1, N)])
    v = dict(n=n, m=m, N=N, formulation="ellipMPC", method="ADMM", submethod="", terminal=True, dim=dim)
    v["Hi_0"] = np.diag(Hinv)[:m].copy()
    v["Hi"] = np.diag(Hinv)[m:m + (N - 1) * nm].reshape(N - 1, nm).copy()
    v["Hi_N"] = Hinv[dim - n:, dim - n:].copy()
    v["AB"] = np.hstack([A, B])
    v["LBu0"], v["UBu0"], v["LBz"], v["UBz"] = LBu.copy(), UBu.copy(), LBz, UBz
    v["P"], v["P_half"], v["Pinv_half"] = P.copy(), P_half, np.linalg.inv(P) @ P_half
    v["Q"], v["R"], v["T"] = -np.diag(Q).copy(), -np.diag(R).copy(), -T
    v["c"], v["r"] = c.copy(), r
    if scalar:
        v["rho"], v["rho_i"], v["rho_is_scalar"] = rho, 1.0 / rho, True
    else:  # (:167-174) the blob's names for the [N-1][n+m] pair are rho_v / rho_i_v
        v["rho"], v["rho_i"], v["rho_is_scalar"] = 0.0, 0.0, False
        v["rho_0"], v["rho_v"], v["rho_N"] = rho[:m].copy(), rho[m:dim - n].reshape(N - 1, nm).copy(), rho[dim - n:].copy()
        v["rho_i_0"], v["rho_i_v"], v["rho_i_N"] = 1.0 / v["rho_0"], 1.0 / v["rho_v"], 1.0 / v["rho_N"]
    Beta = np.zeros((N, n, n))
    Alpha = np.zeros((N - 1, n, n))
    for i in range(N):
        Beta[i] = Wc[i * n:(i + 1) * n, i * n:(i + 1) * n]
        Beta[i][np.diag_indices(n)] = 1.0 / np.diag(Beta[i])
    for i in range(N - 1):
        Alpha[i] = Wc[i * n:(i + 1) * n, (i + 1) * n:(i + 2) * n]
    v["Alpha"], v["Beta"] = Alpha, Beta
    v["k_max"] = int(opt.solver["k_max"])
    v["tol"] = float(opt.solver["tol"])
    return v
