"""MPCT (MPC for tracking) - EADMM ingredients, host-side (offline) restatement.

Reference: ``formulations/+MPCT/compute_MPCT_EADMM_ingredients.m:75-282``.  Three-block extended
ADMM on ``A1 z1 + A2 z2 + A3 z3 = b`` (``b[0:n] = x0``): ``z1 = (x_i, u_i)`` box-constrained,
``z2 = (x_s, u_s)`` the artificial reference (explicit solution through ``W2``), ``z3 = (x^_i, u^_i)``
the dynamics-constrained copy (banded Cholesky ``Alpha`` / ``Beta`` of ``W3 = Az3 H3^-1 Az3'``).
"""
from __future__ import annotations

import numpy as np

from .laxMPC import _get, _is_diag


def compute_MPCT_EADMM_ingredients(controller, opt):
    sys, param = _get(controller, "sys"), _get(controller, "param")
    A = np.asarray(_get(sys, "A"), dtype=float)
    B = np.asarray(_get(sys, "B"), dtype=float)
    n, m = B.shape
    nm = n + m
    N = int(_get(param, "N"))
    Q = np.asarray(_get(param, "Q"), dtype=float)
    R = np.asarray(_get(param, "R"), dtype=float)
    T = np.asarray(_get(param, "T"), dtype=float)
    S = np.asarray(_get(param, "S"), dtype=float)
    inf_value = float(opt.inf_value)
    bnd = lambda name, sign: np.asarray(_get(sys, name, sign * inf_value * np.ones(n if "x" in name else m)), dtype=float)
    LBx, UBx, LBu, UBu = bnd("LBx", -1), bnd("UBx", 1), bnd("LBu", -1), bnd("UBu", 1)
    so = dict(opt.solver)
    if "rho" in so:  # (:76-79)
        so["rho_base"], so["rho_mult"] = so["rho"], 1
    rho_base, rho_mult = float(so["rho_base"]), float(so["rho_mult"])
    is_diag = bool(opt.force_diagonal and _is_diag(Q) and _is_diag(R))  # (:142-148): IS_DIAG of the generated solver
    L = (N + 1) * nm + n + nm
    rho = rho_base * np.ones(L)
    rho[:2 * n] = rho_mult * rho_base          # x_0 = x (6b) and stage-0 state rows of z1 + z2 + z3 = 0
    rho[L - 2 * nm:] = rho_mult * rho_base     # stage-N rows and x_N = x_s, u_N = u_s
    # A1, A2, A3 (:94-101)
    A1 = -np.vstack([np.hstack([-np.eye(n), np.zeros((n, N * nm + m))]), np.eye((N + 1) * nm),
                     np.hstack([np.zeros((nm, N * nm)), np.eye(nm)])])
    A2 = np.vstack([np.zeros((n, nm)), np.kron(np.ones((N, 1)), np.eye(nm)), np.kron(np.ones((2, 1)), np.eye(nm))])
    A3 = np.vstack([np.zeros((n, (N + 1) * nm)), np.eye((N + 1) * nm), np.zeros((nm, (N + 1) * nm))])
    H1 = (rho[:, None] * A1).T @ A1
    H1i = 1.0 / np.diag(H1)
    H2 = np.block([[T, np.zeros((n, m))], [np.zeros((m, n)), S]]) + (rho[:, None] * A2).T @ A2
    Az2 = np.hstack([A - np.eye(n), B])
    H2i = np.linalg.inv(H2)
    W2 = H2i @ Az2.T @ np.linalg.inv(Az2 @ H2i @ Az2.T) @ Az2 @ H2i - H2i
    QR = np.block([[Q, np.zeros((n, m))], [np.zeros((m, n)), R]])
    H3 = np.kron(np.eye(N + 1), QR) + (rho[:, None] * A3).T @ A3
    Az3 = np.kron(np.eye(N), np.hstack([A, B]))
    for j in range(N - 1):  # -I at block (j, j+1) (:127-131)
        Az3[j * n:(j + 1) * n, (j + 1) * nm:(j + 1) * nm + n] = -np.eye(n)
    Az3 = np.hstack([Az3, np.vstack([np.zeros(((N - 1) * n, n)), -np.eye(n)]), np.zeros((N * n, m))])
    H3i = 1.0 / np.diag(H3)
    W3 = (Az3 * H3i[None, :]) @ Az3.T if is_diag else Az3 @ np.linalg.inv(H3) @ Az3.T
    W3c = np.linalg.cholesky(W3).T
    eps_x, eps_u = float(so["epsilon_x"]), float(so["epsilon_u"])
    fin = lambda a: np.where(np.isinf(a), np.sign(a) * inf_value, a)
    v = dict(n=n, m=m, N=N, formulation="MPCT", method="EADMM", terminal=True)
    v["H1i"] = H1i.reshape(N + 1, nm).copy()
    v["is_diag"] = is_diag
    if is_diag:
        v["H3i"] = H3i.reshape(N + 1, nm).copy()
    else:  # (:149-154, :204-211): dense inverses of the two distinct diagonal blocks of H3 and [A B] times them
        v["Q_mi"] = np.linalg.inv(Q + rho_mult * rho_base * np.eye(n))
        v["R_mi"] = np.linalg.inv(R + rho_mult * rho_base * np.eye(m))
        v["Q_bi"] = np.linalg.inv(Q + rho_base * np.eye(n))
        v["R_bi"] = np.linalg.inv(R + rho_base * np.eye(m))
        AB_ = np.hstack([A, B])
        v["AB_bi"] = AB_ @ np.block([[v["Q_bi"], np.zeros((n, m))], [np.zeros((m, n)), v["R_bi"]]])
        v["AB_mi"] = AB_ @ np.block([[v["Q_mi"], np.zeros((n, m))], [np.zeros((m, n)), v["R_bi"]]])
    v["AB"] = np.hstack([A, B])
    v["W2"] = W2
    v["T"] = -T
    v["S"] = -S
    v["LB"] = fin(np.concatenate([LBx, LBu]))
    v["UB"] = fin(np.concatenate([UBx, UBu]))
    v["LBs"] = fin(np.concatenate([LBx + eps_x, LBu + eps_u]))
    v["UBs"] = fin(np.concatenate([UBx - eps_x, UBu - eps_u]))
    v["LB0"] = np.concatenate([-inf_value * np.ones(n), LBu])
    v["UB0"] = np.concatenate([inf_value * np.ones(n), UBu])
    v["rho_mat"] = rho[n:L - nm].reshape(N + 1, nm).copy()
    v["rho_0"] = np.concatenate([rho[:n], np.zeros(m)])
    v["rho_s"] = rho[L - nm:].copy()
    Beta = np.zeros((N, n, n))
    Alpha = np.zeros((N - 1, n, n))
    for i in range(N):
        Beta[i] = W3c[i * n:(i + 1) * n, i * n:(i + 1) * n]
        Beta[i][np.diag_indices(n)] = 1.0 / np.diag(Beta[i])
    for i in range(N - 1):
        Alpha[i] = W3c[i * n:(i + 1) * n, (i + 1) * n:(i + 2) * n]
    v["Alpha"], v["Beta"] = Alpha, Beta
    v["k_max"] = int(so["k_max"])
    v["tol"] = float(so["tol"])
    v["rho"] = rho_base
    v["rho_i"] = 1.0 / rho_base
    v["rho_is_scalar"] = True
    v["dim"] = (N + 1) * nm
    return v


def compute_MPCT_ADMM_cs_ingredients(controller, opt):
    """MPCT ADMM on the extended state space ('cs' submethod):
    ``formulations/+MPCT/compute_MPCT_ADMM_cs_ingredients.m:69-141``.  Stage variable ``(x_j, x_s, u_j, u_s)`` of
    ``2(n+m)`` entries (the artificial reference is repeated in every stage and tied by equality rows),
    ``dim = 2N(n+m)``; ``W = Aeq Hhat^-1 Aeq'`` factorised ``L D L'`` from its Cholesky factor (no permutation);
    ``-Aeq Hhat^-1``, ``-Hhat^-1 Aeq'`` and ``-Hhat^-1`` in CSR."""
    from .. import sp_utils
    sys, param = _get(controller, "sys"), _get(controller, "param")
    A = np.asarray(_get(sys, "A"), dtype=float)
    B = np.asarray(_get(sys, "B"), dtype=float)
    n, m = B.shape
    nm, dnm = n + m, 2 * (n + m)
    N = int(_get(param, "N"))
    Q, R = np.asarray(_get(param, "Q"), float), np.asarray(_get(param, "R"), float)
    T, S = np.asarray(_get(param, "T"), float), np.asarray(_get(param, "S"), float)
    inf_value = float(opt.inf_value)
    bnd = lambda name, sign: np.ravel(np.asarray(_get(sys, name, sign * inf_value * np.ones(n if "x" in name else m)), dtype=float))
    LBx, UBx, LBu, UBu = bnd("LBx", -1), bnd("UBx", 1), bnd("LBu", -1), bnd("UBu", 1)
    so = opt.solver
    rho = so["rho"]
    if np.isscalar(rho) and so.get("force_vector_rho", False):  # (:71-75; the vector has one entry per decision variable)
        rho = float(rho) * np.ones(N * dnm)
    rho_is_scalar = bool(np.isscalar(rho) or np.size(rho) == 1)
    rho = float(np.ravel(rho)[0]) if rho_is_scalar else np.ravel(np.asarray(rho, dtype=float))
    if not rho_is_scalar and rho.size != N * dnm:
        raise ValueError("MPCT ADMM cs: a vector rho needs 2 N (n + m) entries")
    # Hessian (:83-91)
    Qz = np.block([[Q, -Q], [-Q, Q + T / N]])
    Rz = np.block([[R, -R], [-R, R + S / N]])
    Hs = np.block([[Qz, np.zeros((2 * n, 2 * m))], [np.zeros((2 * m, 2 * n)), Rz]])
    Hhat = np.kron(np.eye(N), Hs) + (rho * np.eye(N * dnm) if rho_is_scalar else np.diag(rho))
    # equality constraints (:96-110)
    Z = np.zeros
    AA = np.block([[A, Z((n, n))], [Z((n, n)), np.eye(n)], [Z((m, 2 * n))]])
    BB = np.block([[B, Z((n, m))], [Z((n, 2 * m))], [Z((m, m)), np.eye(m)]])
    II = np.block([[-np.eye(n), Z((n, n + 2 * m))], [Z((n, n)), -np.eye(n), Z((n, 2 * m))], [Z((m, 2 * n + m)), -np.eye(m)]])
    rs = 2 * n + m
    Aeq = Z(((N - 1) * rs + n, N * dnm))
    for j in range(N - 1):
        Aeq[j * rs:(j + 1) * rs, j * dnm:(j + 1) * dnm] = np.hstack([AA, BB])
        Aeq[j * rs:(j + 1) * rs, (j + 1) * dnm:(j + 2) * dnm] = II
    Aeq[(N - 1) * rs:, (N - 1) * dnm:] = np.hstack([A, -np.eye(n), B, Z((n, m))])
    init = np.block([[np.eye(n), Z((n, n + 2 * m))], [Z((n, n)), A - np.eye(n), Z((n, m)), B]])
    Aeq = np.vstack([np.hstack([init, Z((2 * n, N * dnm - dnm))]), Aeq])
    # bounds (:116-122)
    eps_x, eps_u = float(so["epsilon_x"]), float(so["epsilon_u"])
    LBs = np.concatenate([LBx, LBx + eps_x, LBu, LBu + eps_u])
    UBs = np.concatenate([UBx, UBx - eps_x, UBu, UBu - eps_u])
    # W = Aeq Hhat^-1 Aeq' and its L D L' (:125-134)
    Hinv = np.linalg.inv(Hhat)
    W = Aeq @ Hinv @ Aeq.T
    Wc = np.linalg.cholesky(W).T
    wd = np.diag(Wc)
    L = Wc.T / wd[None, :]  # (a division, so that the diagonal is exactly 1 and L - I strictly lower triangular)
    Lv, Lr, Lc, *_ = sp_utils.full2CSC(L - np.eye(L.shape[0]))
    csr = lambda M: sp_utils.full2CSR(M)[:3]
    v = dict(n=n, m=m, N=N, formulation="MPCT", method="ADMM", submethod="cs", terminal=True, dim=N * dnm,
             nrow_AHi=Aeq.shape[0], rho_is_scalar=rho_is_scalar)
    v["Tz"], v["Sz"] = -T / N, -S / N
    v["LB"], v["UB"] = np.tile(LBs, N), np.tile(UBs, N)
    v["rho"] = rho if rho_is_scalar else float(rho[0])
    v["rho_i"] = 1.0 / v["rho"]
    if not rho_is_scalar:
        v["rho_cs"], v["rho_i_cs"] = rho, 1.0 / rho
    v["L_val"], v["L_row"], v["L_col"], v["Dinv"] = Lv, Lr, Lc, 1.0 / (wd * wd)
    v["AHi_val"], v["AHi_col"], v["AHi_row"] = csr(-Aeq @ Hinv)
    v["HiA_val"], v["HiA_col"], v["HiA_row"] = csr(-Hinv @ Aeq.T)
    v["Hi_val"], v["Hi_col"], v["Hi_row"] = csr(-Hinv)
    v["k_max"], v["tol"] = int(so["k_max"]), float(so["tol"])
    v["Aeq"], v["Hhat"] = Aeq, Hhat  # dense forms, for tests
    return v
