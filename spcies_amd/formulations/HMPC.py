"""HMPC (harmonic MPC), ADMM / SADMM with the (z_hat, s_hat) = (z, s) splitting ('split' submethod),
sparse KKT path, box constraints - host-side (offline) ingredients.

Reference: ``formulations/+HMPC/compute_HMPC_ADMM_split_ingredients.m:23-323`` (the SADMM file is the
same computation).  Decision vector ``z = (u0, x1, u1, ..., x_{N-1}, u_{N-1}, xe, xs, xc, ue, us, uc)``
(``dim = (N-1)(n+m) + m + 3(n+m)``), slack ``s`` with ``C z + s = d`` in a product of 3-D cones
("diamond" = two shifted SOCs per constrained signal, or plain SOCs with ``use_soc``).  The KKT
matrix ``M = [Hh Gh'; Gh 0]`` (quasi-definite) is factorised ``L D L'`` with diagonal ``D``.

The reference takes whatever permutation MATLAB's ``ldl`` returns (``:228-234``) - the iterates do
not depend on it - so the factorisation order is ours to choose: primal block first in natural order
(the solver writes its right-hand side there, ``code_HMPC_ADMM_split_C.c:156-165``), constraint rows
after it in a reverse-Cuthill-McKee order of their Schur complement to limit fill.
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee

from types import SimpleNamespace

from .. import sp_utils
from .laxMPC import _get


def _ldl_nopivot(M):
    """Dense L D L' without pivoting (exists for every symmetric permutation of a quasi-definite matrix)."""
    n = M.shape[0]
    L = np.eye(n)
    D = np.zeros(n)
    A = M.copy()
    for j in range(n):
        D[j] = A[j, j]
        if D[j] == 0.0:
            raise np.linalg.LinAlgError("zero pivot in the KKT LDL factorisation")
        L[j + 1:, j] = A[j + 1:, j] / D[j]
        A[j + 1:, j + 1:] -= np.outer(L[j + 1:, j], A[j, j + 1:])
    return L, D


def _hmpc_problem(controller, opt):
    """H, G, the cone rows C_aux / dsoc and the bounds of the HMPC problem (box-constraint case) - common to the split and
    the non-split solver (``compute_HMPC_ADMM_split_ingredients.m:60-218`` = ``compute_HMPC_ADMM_ingredients.m:60-230``).

    Solver option ``stage0_cost`` (default true = the snapshot as coded).  False drops the stage-0 STATE term
    ``|x0 - xe - xc|_Q^2`` from the objective: ``H22`` then sums ``j = 1 .. N-1`` (``:114-116`` sum ``j = 0 .. N-1``:
    ``N*Q``, ``sum cos``, ``sum cos^2``) and the solver's ``q`` gets no ``-Q x0`` rows (``QQ`` is shipped as zeros;
    ``code_HMPC_ADMM_split_C.c:115-124``).  That is the formulation the reference tests' ``z_opt`` was computed for
    (``tests/test_HMPC_ADMM_s.m:25``, ``test_HMPC_ADMM.m:24``) - it is what pins the HMPC oracles."""
    sys, param = _get(controller, "sys"), _get(controller, "param")
    A = np.asarray(_get(sys, "A"), dtype=float)
    B = np.asarray(_get(sys, "B"), dtype=float)
    n, m = B.shape
    nm = n + m
    so = opt.solver
    use_soc = bool(so.get("use_soc", False))
    LBx, UBx = np.ravel(_get(sys, "LBx")).astype(float), np.ravel(_get(sys, "UBx")).astype(float)
    LBu, UBu = np.ravel(_get(sys, "LBu")).astype(float), np.ravel(_get(sys, "UBu")).astype(float)
    # box constraints on (x, u), or coupled output constraints LBy <= E x + F u <= UBy (:35-53; cons_HMPC_ADMM_split_C.m:51-56:
    # the option defaults to "coupled" when the system carries E / F)
    coupled = so.get("box_constraints") is False or (so.get("box_constraints") is None and _get(sys, "E") is not None)
    if coupled:
        E, F = np.atleast_2d(np.asarray(_get(sys, "E"), float)), np.atleast_2d(np.asarray(_get(sys, "F"), float))
        LBy, UBy = np.ravel(_get(sys, "LBy")).astype(float), np.ravel(_get(sys, "UBy")).astype(float)
    else:
        E = np.vstack([np.eye(n), np.zeros((m, n))])
        F = np.vstack([np.zeros((n, m)), np.eye(m)])
        LBy, UBy = np.concatenate([LBx, LBu]), np.concatenate([UBx, UBu])
    n_y = len(LBy)
    N = int(_get(param, "N"))
    w = float(_get(param, "w"))
    Q, R = np.asarray(_get(param, "Q"), float), np.asarray(_get(param, "R"), float)
    Te, Th = np.asarray(_get(param, "Te"), float), np.asarray(_get(param, "Th"), float)
    Se, Sh = np.asarray(_get(param, "Se"), float), np.asarray(_get(param, "Sh"), float)
    sj, cj = np.sin(w * np.arange(N)), np.cos(w * np.arange(N))
    stage0 = bool(so.get("stage0_cost", True))
    j0 = 0 if stage0 else 1  # first stage whose state enters the objective
    NQ = N - j0
    s_sum, c_sum, s2, c2, sc = sj[j0:].sum(), cj[j0:].sum(), (sj[j0:] ** 2).sum(), (cj[j0:] ** 2).sum(), (sj[j0:] * cj[j0:]).sum()
    sR, cR, s2R, c2R, scR = sj.sum(), cj.sum(), (sj ** 2).sum(), (cj ** 2).sum(), (sj * cj).sum()  # inputs: every stage
    # ---- Hessian (:98-127)
    d1 = (N - 1) * nm + m
    H11 = np.zeros((d1, d1))
    H11[:m, :m] = R
    for l in range(N - 1):
        o = m + l * nm
        H11[o:o + n, o:o + n] = Q
        H11[o + n:o + nm, o + n:o + nm] = R
    H12 = np.zeros((d1, 3 * n))
    for j in range(N - 1):
        H12[j * nm + m:(j + 1) * nm, :] = np.kron(np.array([[1.0, sj[j + 1], cj[j + 1]]]), -Q)
    H13 = np.zeros((d1, 3 * m))
    for j in range(N):
        H13[j * nm:j * nm + m, :] = np.kron(np.array([[1.0, sj[j], cj[j]]]), -R)
    H22 = np.block([[Te + NQ * Q, s_sum * Q, c_sum * Q], [s_sum * Q, Th + s2 * Q, sc * Q], [c_sum * Q, sc * Q, Th + c2 * Q]])
    H33 = np.block([[Se + N * R, sR * R, cR * R], [sR * R, Sh + s2R * R, scR * R], [cR * R, scR * R, Sh + c2R * R]])
    H = np.block([[H11, H12, H13], [H12.T, H22, np.zeros((3 * n, 3 * m))], [H13.T, np.zeros((3 * m, 3 * n)), H33]])
    dim = H.shape[0]
    # ---- equality constraints (:131-142): N-2 interior -I blocks, harmonic closure of the last dynamics row,
    #      three steady-state / harmonic-oscillator rows
    G = np.kron(np.eye(N - 1), np.hstack([A, B]))
    for j in range(N - 2):
        G[j * n:(j + 1) * n, j * nm + nm:j * nm + nm + n] = -np.eye(n)
    G = np.block([[B, -np.eye(n), np.zeros((n, G.shape[1] - n))], [np.zeros((G.shape[0], m)), G]])
    G = np.hstack([G, np.vstack([np.zeros((G.shape[0] - n, 3 * nm)),
                                 np.hstack([-np.eye(n), -np.eye(n) * np.sin(w * N), -np.eye(n) * np.cos(w * N),
                                            np.zeros((n, 3 * m))])])])
    I_n, Z_n = np.eye(n), np.zeros((n, n))
    tail = np.block([[A - I_n, Z_n, Z_n, B, np.zeros((n, 2 * m))],
                     [Z_n, A - np.cos(w) * I_n, np.sin(w) * I_n, np.zeros((n, m)), B, np.zeros((n, m))],
                     [Z_n, -np.sin(w) * I_n, A - np.cos(w) * I_n, np.zeros((n, 2 * m)), B]])
    G = np.vstack([G, np.hstack([np.zeros((3 * n, G.shape[1] - 3 * nm)), tail])])
    n_eq = G.shape[0]
    b = np.zeros(n_eq)
    # ---- cone constraints (:147-218): the coupled form, rows kron(I3, -E_j) | kron(I3, -F_j) per output j, IS the box form when
    #      E = [I; 0], F = [0; I] - except that the reference lists the box form's columns blockwise (C_n, C_m), kept below
    if use_soc:
        bd3 = lambda a, b_, c_: np.block([[a, np.zeros_like(a), np.zeros_like(a)], [np.zeros_like(a), b_, np.zeros_like(a)],
                                          [np.zeros_like(a), np.zeros_like(a), c_]])
        rows, dsoc = [], []
        for j in range(n_y):
            e, f = E[j:j + 1, :], F[j:j + 1, :]
            rows.append(np.hstack([bd3(e, -e, -e), bd3(f, -f, -f)]))
            rows.append(np.hstack([bd3(-e, -e, -e), bd3(-f, -f, -f)]))
            dsoc += [UBy[j], 0.0, 0.0, -LBy[j], 0.0, 0.0]
        C_aux, dsoc, n_soc = np.vstack(rows), np.array(dsoc), 2 * n_y
    elif coupled:
        C_aux = np.vstack([np.hstack([np.kron(np.eye(3), -E[j:j + 1, :]), np.kron(np.eye(3), -F[j:j + 1, :])]) for j in range(n_y)])
        dsoc, n_soc = np.zeros(3 * n_y), n_y
    else:
        C_n = np.vstack([np.kron(np.eye(3), -np.eye(n)[j:j + 1, :]) for j in range(n)])
        C_m = np.vstack([np.kron(np.eye(3), -np.eye(m)[j:j + 1, :]) for j in range(m)])
        C_aux = np.block([[C_n, np.zeros((3 * n, 3 * m))], [np.zeros((3 * m, 3 * n)), C_m]])
        dsoc, n_soc = np.zeros(3 * n_y), n_y
    LB = np.concatenate([LBu] + [np.concatenate([LBx, LBu])] * (N - 1))
    UB = np.concatenate([UBu] + [np.concatenate([UBx, UBu])] * (N - 1))
    if not stage0:
        Q = np.zeros_like(Q)  # QQ of the generated solver: only multiplies x0 (code_HMPC_ADMM_split_C.c:115-124)
    # coupled: one box slack per output and stage, s_box = -(E x_j + F u_j) ... the reference's C = blkdiag(-F, kron(I, [-E -F]), C_aux)
    Cbox = None
    if coupled:
        Cbox = np.zeros((N * n_y, dim - 3 * nm))
        Cbox[:n_y, :m] = -F
        for j in range(1, N):
            Cbox[j * n_y:(j + 1) * n_y, m + (j - 1) * nm:m + j * nm] = np.hstack([-E, -F])
    return SimpleNamespace(A=A, n=n, m=m, N=N, Q=Q, Te=Te, Se=Se, H=H, G=G, b=b, C_aux=C_aux, dsoc=dsoc, n_soc=n_soc,
                           use_soc=use_soc, LB=LB, UB=UB, LBy=LBy, UBy=UBy, coupled=coupled, n_y=n_y, Cbox=Cbox)


def compute_HMPC_ADMM_split_ingredients(controller, opt, reorder=True):
    P = _hmpc_problem(controller, opt)
    A, n, m, N, Q, Te, Se, H, G, b = P.A, P.n, P.m, P.N, P.Q, P.Te, P.Se, P.H, P.G, P.b
    C_aux, dsoc, n_soc, use_soc, LBy, UBy = P.C_aux, P.dsoc, P.n_soc, P.use_soc, P.LBy, P.UBy
    nm, dim, n_eq = n + m, H.shape[0], G.shape[0]
    so = opt.solver
    rho, sigma = float(so["rho"]), float(so["sigma"])
    C = np.hstack([np.zeros((3 * n_soc, dim - 3 * nm)), C_aux])
    if P.coupled:  # (:174-175) box slacks of the outputs first, then the cone rows
        C = np.block([[P.Cbox, np.zeros((P.Cbox.shape[0], 3 * nm))], [C]])
        dsoc = np.concatenate([np.zeros(P.Cbox.shape[0]), dsoc])
    n_s = C.shape[0]
    # ---- KKT matrix and its L D L' (:221-234)
    Hh = np.block([[H + sigma * np.eye(dim), np.zeros((dim, n_s))], [np.zeros((n_s, dim)), rho * np.eye(n_s)]])
    Gh = np.block([[G, np.zeros((n_eq, n_s))], [C, np.eye(n_s)]])
    bh = np.concatenate([b, dsoc])
    nc = n_eq + n_s
    # order of the constraint rows behind the primal block: the reference takes whatever MATLAB's ldl returns there (:228-234,
    # idx_x0 tracks the x0 rows); solver option kkt_order = 'rcm' (default, least fill), 'natural', or 'random:<seed>' - any
    # permutation of a quasi-definite matrix has an L D L' with diagonal D, and the iterates do not depend on it
    perm2 = np.arange(nc)
    order = str(so.get("kkt_order", "rcm" if reorder else "natural"))
    if order == "rcm":
        S = Gh @ np.linalg.solve(Hh, Gh.T)
        perm2 = np.asarray(reverse_cuthill_mckee(sp.csr_matrix(np.abs(S) > 1e-14), symmetric_mode=True))
    elif order.startswith("random"):
        perm2 = np.random.default_rng(int(order.split(":")[1]) if ":" in order else 0).permutation(nc)
    elif order != "natural":
        raise ValueError("kkt_order: 'rcm', 'natural' or 'random:<seed>'")
    Ghp = Gh[perm2]
    M = np.block([[Hh, Ghp.T], [Ghp, np.zeros((nc, nc))]])
    L, D = _ldl_nopivot(M)
    if not np.all(np.isfinite(L)) or np.abs(L @ np.diag(D) @ L.T - M).max() > 1e-8 * max(1.0, np.abs(M).max()):
        raise np.linalg.LinAlgError("KKT LDL factorisation failed")
    Lv, Lr, Lc, *_ = sp_utils.full2CSC(L - np.eye(L.shape[0]), threshold=0.0)
    inv = np.empty(nc, dtype=int)
    inv[perm2] = np.arange(nc)
    v = dict(n=n, m=m, N=N, formulation="HMPC", method=opt.method or "ADMM", submethod="split", terminal=True,
             dim=dim, n_s=n_s, n_eq=n_eq, n_soc=n_soc, use_soc=use_soc, nrow_M=dim + n_s + nc, coupled=P.coupled, n_y=P.n_y)
    v["A"], v["Q"], v["Te"], v["Se"] = A.copy(), Q.copy(), Te.copy(), Se.copy()
    v["LB"], v["UB"] = P.LB, P.UB
    v["LBy"], v["UBy"] = LBy, UBy
    v["L_val"], v["L_row"], v["L_col"], v["Dinv"] = Lv, Lr, Lc, 1.0 / D
    v["idx_x0"] = inv[:n].astype(np.int32)       # where the first n equality rows (x0 rows) sit in the permuted tail
    v["bh"] = bh[perm2]
    # NON_SPARSE path (option sparse = false, the reference's default): primal_hat = M2 bh - M1 q_hat (:236-239)
    Hhi = np.linalg.inv(Hh)
    Wi = np.linalg.inv(Gh @ Hhi @ Gh.T)
    v["M1"] = Hhi @ Gh.T @ Wi @ Gh @ Hhi - Hhi
    v["M2"] = Hhi @ Gh.T @ Wi
    v["bh_nat"] = bh.copy()
    v["sparse"] = bool(so.get("sparse", False))
    v["rho"], v["rho_i"], v["sigma"], v["sigma_i"] = rho, 1.0 / rho, sigma, 1.0 / sigma
    v["alpha"] = float(so.get("alpha", 0.95))
    v["k_max"] = int(so["k_max"])
    v["tol_p"], v["tol_d"] = float(so["tol_p"]), float(so["tol_d"])
    v["tol"] = v["tol_p"]
    v["rho_is_scalar"] = True
    v["H"], v["G"], v["C"], v["d"] = H, G, C, dsoc  # dense forms, for tests (KKT residual of the solution)
    return v


def compute_HMPC_ADMM_ingredients(controller, opt):
    """HMPC ADMM / SADMM WITHOUT the splitting (the reference's default HMPC solver):
    ``formulations/+HMPC/compute_HMPC_ADMM_ingredients.m:60-300``, box-constraint case.  Same ``H``, ``G`` and cone rows
    as the split formulation; here every bounded decision variable gets a slack too, ``C = blkdiag(-I, C_aux)``,
    ``s`` has ``n_box = dim - 3(n+m)`` box rows followed by the cone rows, and the z-update is the dense
    ``z = M2 b + M1 q_hat`` with ``Hh = H + rho C'C``, ``W = G Hh^-1 G'``, ``M1 = Hh^-1 G' W^-1 G Hh^-1 - Hh^-1``,
    ``M2 = (Hh^-1 G' W^-1)(:, 1:n)``  (``:236-254``)."""
    P = _hmpc_problem(controller, opt)
    n, m, N, H, G, C_aux = P.n, P.m, P.N, P.H, P.G, P.C_aux
    nm, dim = n + m, H.shape[0]
    so = opt.solver
    n_box = dim - 3 * nm
    C = np.block([[-np.eye(n_box), np.zeros((n_box, 3 * nm))], [np.zeros((C_aux.shape[0], n_box)), C_aux]])
    LBbox, UBbox = P.LB, P.UB
    if P.coupled:  # (compute_HMPC_ADMM_ingredients.m:155-180): N n_y box slacks of the outputs, bounds LBy / UBy per stage
        n_box = N * P.n_y
        C = np.block([[P.Cbox, np.zeros((n_box, 3 * nm))], [np.zeros((C_aux.shape[0], dim - 3 * nm)), C_aux]])
        LBbox, UBbox = np.tile(P.LBy, N), np.tile(P.UBy, N)
    d = np.concatenate([np.zeros(n_box), P.dsoc])
    rho = float(so["rho"])
    Hhi = np.linalg.inv(H + rho * (C.T @ C))
    Wi = np.linalg.inv(G @ Hhi @ G.T)
    M1 = Hhi @ G.T @ Wi @ G @ Hhi - Hhi
    M2 = (Hhi @ G.T @ Wi)[:, :n]
    csr = lambda M: sp_utils.full2CSR(M)[:3]
    v = dict(n=n, m=m, N=N, formulation="HMPC", method=opt.method or "ADMM", submethod="", terminal=True, dim=dim,
             n_s=C.shape[0], n_eq=G.shape[0], n_soc=P.n_soc, n_box=n_box, use_soc=P.use_soc)
    v["A"], v["Q"], v["Te"], v["Se"] = P.A.copy(), P.Q.copy(), P.Te.copy(), P.Se.copy()
    v["LB"], v["UB"], v["LBy"], v["UBy"] = LBbox, UBbox, P.LBy, P.UBy
    v["coupled"], v["n_y"] = P.coupled, P.n_y
    v["rho"], v["rho_i"] = rho, 1.0 / rho
    v["alpha"] = float(so.get("alpha", 0.95)) if v["method"] == "SADMM" else 1.0  # :256-258
    v["k_max"] = int(so["k_max"])
    v["tol_p"], v["tol_d"] = float(so["tol_p"]), float(so["tol_d"])
    v["tol"] = v["tol_p"]
    v["C_val"], v["C_col"], v["C_row"] = csr(C)
    v["Ct_val"], v["Ct_col"], v["Ct_row"] = csr(C.T)
    v["d"], v["M1"], v["M2"] = d, M1, M2
    v["rho_is_scalar"] = True
    v["sigma"], v["sigma_i"] = 0.0, 0.0
    v["H"], v["G"], v["C"] = H, G, C  # dense forms, for tests
    return v
