"""laxMPC / equMPC ADMM ingredients - host-side (offline) restatement.

Reference: ``formulations/+laxMPC/compute_laxMPC_ADMM_ingredients.m:22-186`` and
``formulations/+equMPC/compute_equMPC_ADMM_ingredients.m`` (same structure without the terminal
block).  Decision vector order ``(u0, x1, u1, ..., x_{N-1}, u_{N-1}, x_N)`` (``:69``); equality
constraint ``G z = b`` with ``b = [-A x0; 0]``; ``W = G Hhat^{-1} G'`` is block tridiagonal and its
upper Cholesky factor block bidiagonal: ``Beta_l`` (diagonal blocks, diagonal entries stored
inverted) and ``Alpha_l`` (super-diagonal blocks) (``:170-183``).
"""
from __future__ import annotations

import numpy as np


def _get(obj, name, default=None):
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def _is_diag(M):
    return np.count_nonzero(M - np.diag(np.diag(M))) == 0


def add_engineering(v, sys, opt):
    """Scaling vectors and operating point the generated solver applies to its arguments when the option
    ``in_engineering`` is set (``compute_laxMPC_ADMM_ingredients.m:136-165``, printed by
    ``cons_laxMPC_ADMM_C.m:110-116``): ``x = scaling_x o (x_in - OpPoint_x)``, ``u_out = v_0 o scaling_i_u + OpPoint_u``
    (``code_laxMPC_ADMM_C.c:83-100, 642-646``).  Missing fields default to ones / zeros as there."""
    if not getattr(opt, "in_engineering", False):
        return v
    n, m = int(v["n"]), int(v["m"])
    get = lambda name, default: np.ravel(np.asarray(_get(sys, name, default), dtype=float))
    v["scaling_x"] = get("Nx", np.ones(n))
    v["scaling_u"] = get("Nu", np.ones(m))
    v["scaling_i_u"] = 1.0 / get("Nu", np.ones(m))
    v["OpPoint_x"] = get("x0", np.zeros(n))
    v["OpPoint_u"] = get("u0", np.zeros(m))
    v["in_engineering"] = True
    return v


def build_G(A, B, N, terminal=True):
    """Equality-constraint matrix of the lax (``terminal=True``) / equ (``False``) formulation.

    lax: ``N*n`` rows, ``N*(n+m)`` columns (``compute_laxMPC_ADMM_ingredients.m:78-86``; the
    reference relies on MATLAB growing ``Aeq`` when the last ``-I`` is written past its edge).
    equ: same rows, but ``x_N`` is not a variable (it is fixed to ``xr`` through ``b``), so the last
    ``n`` columns are absent.
    """
    n, m = B.shape
    nm = n + m
    ncol = N * nm if terminal else N * nm - n
    G = np.zeros((N * n, ncol))
    G[:n, :m] = B
    G[:n, m:m + n] = -np.eye(n)
    for l in range(1, N):
        r = slice(l * n, (l + 1) * n)
        c0 = m + (l - 1) * nm
        G[r, c0:c0 + n] = A
        G[r, c0 + n:c0 + nm] = B
        if terminal or l < N - 1:
            G[r, c0 + nm:c0 + nm + n] = -np.eye(n)
    return G


def compute_laxMPC_ADMM_ingredients(controller, opt, terminal=True):
    """Returns the ``vars`` dict the reference's ``cons_laxMPC_ADMM_C.m:72-130`` prints as C constants.

    ``controller`` needs ``.sys`` (A, B, LBx, UBx, LBu, UBu) and ``.param`` (Q, R, T, N);
    ``opt`` is a :class:`SpciesOptions`.  Scalar ``rho`` only on the HIP platform for now (the
    vector-``rho`` variant is SURVEY section 8f rank 3).
    """
    sys, param = _get(controller, "sys"), _get(controller, "param")
    A = np.asarray(_get(sys, "A"), dtype=float)
    B = np.asarray(_get(sys, "B"), dtype=float)
    n, m = B.shape
    N = int(_get(param, "N"))
    Q = np.asarray(_get(param, "Q"), dtype=float)
    R = np.asarray(_get(param, "R"), dtype=float)
    T = np.asarray(_get(param, "T"), dtype=float)
    if not (_is_diag(Q) and _is_diag(R)):
        raise ValueError("Spcies:laxMPC:ADMM:non_diagonal - matrices Q and R must be diagonal")
    rho = opt.solver["rho"]
    nm = n + m
    dim = N * nm if terminal else N * nm - n
    # vector rho (compute_laxMPC_ADMM_ingredients.m:53-64, 123-133): one penalty per decision variable
    rho_vec = None
    if np.ndim(rho) != 0 or opt.solver.get("force_vector_rho", False):
        rho_vec = np.ravel(np.asarray(rho, dtype=float)) * np.ones(dim)
        if opt.time_varying:
            raise ValueError("LaxMPC ADMM time varying solver only allows the use of a scalar rho")  # cons_laxMPC_ADMM_C.m:50-52
        rho = float("nan")
    else:
        rho = float(rho)
    if opt.time_varying:
        # compute_laxMPC_ADMM_ingredients.m:89-117, cons_laxMPC_ADMM_C.m:97-109: the generated solver receives
        # A, B, Q, R, LB, UB with every call and factorises on line; only T and inv(T + rho I) are constants
        v = dict(n=n, m=m, N=N, formulation="laxMPC" if terminal else "equMPC", method="ADMM", terminal=bool(terminal),
                 time_varying=True, rho=rho, rho_i=1.0 / rho, rho_is_scalar=True, k_max=int(opt.solver["k_max"]),
                 tol=float(opt.solver["tol"]), dim=dim)
        v["T"] = -T if terminal else np.zeros((n, n))
        v["T_rho_i"] = np.linalg.inv(T + rho * np.eye(n)) if terminal else np.zeros((n, n))
        return v

    H = np.zeros((dim, dim))
    H[:m, :m] = R
    for l in range(N - 1):
        o = m + l * nm
        H[o:o + n, o:o + n] = Q
        H[o + n:o + nm, o + n:o + nm] = R
    if terminal:
        H[dim - n:, dim - n:] = T
    Hhat = H + (np.diag(rho_vec) if rho_vec is not None else rho * np.eye(dim))
    G = build_G(A, B, N, terminal=terminal)
    Hinv = np.linalg.inv(Hhat)
    W = G @ Hinv @ G.T
    Wc = np.linalg.cholesky(W).T  # upper, as MATLAB chol()

    v = dict(n=n, m=m, N=N, formulation="laxMPC" if terminal else "equMPC", method="ADMM",
             terminal=bool(terminal))
    v["Hi_0"] = np.diag(Hinv)[:m].copy()
    v["Hi"] = np.diag(Hinv)[m:m + (N - 1) * nm].reshape(N - 1, nm).copy()
    if terminal:
        v["T"] = -T
        v["Hi_N"] = Hinv[dim - n:, dim - n:].copy()
    else:  # arrays the equMPC solver never reads; kept zero so the packed layout is uniform
        v["T"] = np.zeros((n, n))
        v["Hi_N"] = np.zeros((n, n))
    v["AB"] = np.hstack([A, B])
    v["Q"] = -np.diag(Q).copy()
    v["R"] = -np.diag(R).copy()
    LBx, UBx = np.asarray(_get(sys, "LBx"), dtype=float), np.asarray(_get(sys, "UBx"), dtype=float)
    LBu, UBu = np.asarray(_get(sys, "LBu"), dtype=float), np.asarray(_get(sys, "UBu"), dtype=float)
    if LBx.ndim == 2 and LBx.shape[1] > 1:
        # one column per prediction step 0..N (cons_laxMPC_ADMM_C.m:82-90, `VAR_BOUNDS`)
        LB, UB = np.vstack([LBx, LBu.reshape(m, -1)]), np.vstack([UBx, UBu.reshape(m, -1)])
        if LB.shape != (nm, N + 1) or UB.shape != (nm, N + 1):
            raise ValueError("stage-wise bounds must have one column per prediction step 0..N")
        v["var_bounds"] = True
        v["LB0"], v["UB0"] = LB[n:, 0].copy(), UB[n:, 0].copy()
        v["LB"], v["UB"] = LB[:, 1:N].T.copy(), UB[:, 1:N].T.copy()
        v["LBN"], v["UBN"] = LB[:n, N].copy(), UB[:n, N].copy()
    else:
        v["LB"] = np.concatenate([np.ravel(LBx), np.ravel(LBu)])
        v["UB"] = np.concatenate([np.ravel(UBx), np.ravel(UBu)])
    if rho_vec is None:
        v["rho"] = rho
        v["rho_i"] = 1.0 / rho
        v["rho_is_scalar"] = True
    else:
        tail = dim - (n if terminal else 0)
        v["rho"], v["rho_i"], v["rho_is_scalar"] = 0.0, 0.0, False
        v["rho_0"], v["rho_v"] = rho_vec[:m].copy(), rho_vec[m:tail].reshape(N - 1, nm).copy()
        v["rho_N"] = rho_vec[tail:].copy() if terminal else np.ones(n)
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
    v["dim"] = dim
    return v


def compute_equMPC_ADMM_ingredients(controller, opt):
    """equMPC-ADMM: the lax ingredients without the terminal block
    (``formulations/+equMPC/compute_equMPC_ADMM_ingredients.m:66-93``: ``H`` has no ``T`` block and
    ``Aeq = Aeq(:, 1:end-n)``); the solver then imposes ``x_N = xr`` through ``mu[N-1] -= xr``
    (``code_equMPC_ADMM_C.c:351``)."""
    return compute_laxMPC_ADMM_ingredients(controller, opt, terminal=False)


def compute_laxMPC_FISTA_ingredients(controller, opt, terminal=True):
    """laxMPC / equMPC FISTA ingredients (dual fast-gradient method; no ``rho``).

    Reference: ``formulations/+laxMPC/compute_laxMPC_FISTA_ingredients.m:50-165`` and
    ``formulations/+equMPC/compute_equMPC_FISTA_ingredients.m``.  ``W = G H^-1 G'`` with the *plain*
    Hessian; ``Q``, ``R`` and (lax) ``T`` must be diagonal (``:50-52``); ``QRi = -1/diag([Q, R])``,
    ``T = -diag(T)``, ``Ti = -1/diag(T)`` (``:111-119``); Alpha/Beta as in the ADMM solvers.
    """
    sys, param = _get(controller, "sys"), _get(controller, "param")
    A = np.asarray(_get(sys, "A"), dtype=float)
    B = np.asarray(_get(sys, "B"), dtype=float)
    n, m = B.shape
    N = int(_get(param, "N"))
    Q = np.asarray(_get(param, "Q"), dtype=float)
    R = np.asarray(_get(param, "R"), dtype=float)
    T = np.asarray(_get(param, "T"), dtype=float) if terminal else np.eye(n)
    if not (_is_diag(Q) and _is_diag(R) and _is_diag(T)):
        raise ValueError("Spcies:laxMPC:FISTA:non_diagonal - matrices Q, R and T must be diagonal")
    nm = n + m
    if opt.time_varying:  # (compute_laxMPC_FISTA_ingredients.m:71, 92, 138; cons_laxMPC_FISTA_C.m:94-108): only T, Ti are constants
        return dict(n=n, m=m, N=N, formulation="laxMPC" if terminal else "equMPC", method="FISTA", terminal=bool(terminal),
                    time_varying=True, Tdiag=-np.diag(T).copy() if terminal else np.zeros(n),
                    Ti=-1.0 / np.diag(T) if terminal else np.zeros(n), k_max=int(opt.solver["k_max"]), tol=float(opt.solver["tol"]),
                    rho=0.0, rho_i=0.0, rho_is_scalar=True, dim=N * nm if terminal else N * nm - n)
    dim = N * nm if terminal else N * nm - n
    hdiag = np.concatenate([np.diag(R)] + [np.concatenate([np.diag(Q), np.diag(R)])] * (N - 1)
                           + ([np.diag(T)] if terminal else []))
    G = build_G(A, B, N, terminal=terminal)
    W = (G / hdiag[None, :]) @ G.T
    Wc = np.linalg.cholesky(W).T
    v = dict(n=n, m=m, N=N, formulation="laxMPC" if terminal else "equMPC", method="FISTA", terminal=bool(terminal))
    v["AB"] = np.hstack([A, B])
    v["Q"] = -np.diag(Q).copy()
    v["R"] = -np.diag(R).copy()
    v["QRi"] = -np.concatenate([1.0 / np.diag(Q), 1.0 / np.diag(R)])
    v["Tdiag"] = -np.diag(T).copy() if terminal else np.zeros(n)
    v["Ti"] = -1.0 / np.diag(T) if terminal else np.zeros(n)
    v["LB"] = np.concatenate([np.ravel(_get(sys, "LBx")), np.ravel(_get(sys, "LBu"))]).astype(float)
    v["UB"] = np.concatenate([np.ravel(_get(sys, "UBx")), np.ravel(_get(sys, "UBu"))]).astype(float)
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
    v["rho"] = 0.0  # unused by FISTA; kept so the blob header is uniform
    v["rho_i"] = 0.0
    v["rho_is_scalar"] = True
    v["dim"] = dim
    return v


def compute_equMPC_FISTA_ingredients(controller, opt):
    return compute_laxMPC_FISTA_ingredients(controller, opt, terminal=False)
