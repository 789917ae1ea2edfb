"""ORACLE (test infrastructure only) - dense numpy restatement of the reference's *non-sparse*
MATLAB solver, used exactly as the reference's tests use it: as the independent comparator of the
sparse solver (``tests/test_laxMPC_ADMM.m:31-41``, threshold 1e-10, ``tests/spcies_tester.m:260``).

Follows ``platforms/Matlab/spcies_laxMPC_ADMM_solver.m:170-321`` (and the equMPC sibling):
builds ``H``, ``G``, ``W`` densely, solves ``W mu = -G H^{-1} q - b`` with a general solver
(``solve_eqQP.m:16-28``) and clamps (``solve_boxQP.m:16-35``).  It shares nothing with the banded
Alpha/Beta path, so it also validates the host-side ingredient computation.
"""
from __future__ import annotations

import numpy as np


def dense_admm(A, B, Q, R, T, N, LBx, UBx, LBu, UBu, rho, tol, k_max, x0, xr, ur, terminal=True):
    A = np.asarray(A, float); B = np.asarray(B, float)
    n, m = B.shape
    nm = n + m
    dim = N * nm - (0 if terminal else n)
    H = np.zeros((dim, dim))
    H[:m, :m] = R
    for l in range(N - 1):
        o = m + l * nm
        H[o:o + n, o:o + n] = Q
        H[o + n:o + nm, o + n:o + nm] = R
    if terminal:
        H[dim - n:, dim - n:] = T
    Hrho = H + rho * np.eye(dim)
    G = np.zeros((N * n, dim))
    G[:n, :m] = B
    G[:n, m:m + n] = -np.eye(n)
    for l in range(1, N):
        c0 = m + (l - 1) * nm
        G[l * n:(l + 1) * n, c0:c0 + nm] = np.hstack([A, B])
        if c0 + nm + n <= dim:
            G[l * n:(l + 1) * n, c0 + nm:c0 + nm + n] = -np.eye(n)
    Hinv = np.linalg.inv(Hrho)
    W = G @ Hinv @ G.T
    LB = np.concatenate([LBu] + [np.concatenate([LBx, LBu])] * (N - 1) + ([LBx] if terminal else []))
    UB = np.concatenate([UBu] + [np.concatenate([UBx, UBu])] * (N - 1) + ([UBx] if terminal else []))
    b = np.zeros(N * n)
    b[:n] = -A @ x0
    if not terminal:
        b[-n:] = xr  # x_N = xr moved to the right-hand side
    q = -np.concatenate([R @ ur] + [np.concatenate([Q @ xr, R @ ur])] * (N - 1) + ([T @ xr] if terminal else []))
    z = np.zeros(dim); v = np.zeros(dim); lam = np.zeros(dim); v1 = v.copy()
    k = 0
    while True:
        k += 1
        qk = q + lam - rho * v
        mu = np.linalg.solve(W, -G @ (Hinv @ qk) - b)
        z = -Hinv @ (G.T @ mu) - Hinv @ qk
        v = np.clip(z + lam / rho, LB, UB)
        lam = lam + rho * (z - v)
        rp = np.max(np.abs(z - v)); rd = np.max(np.abs(v - v1))
        v1 = v.copy()
        if rp <= tol and rd <= tol:
            e = 1
            break
        if k >= k_max:
            e = -1
            break
    return v[:m].copy(), k, e, z, v, lam
