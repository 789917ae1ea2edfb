"""Host-side utilities mirroring the reference's ``+sp_utils`` package.

Only what the batched hot path needs is restated here (numpy/scipy, no MATLAB):

* plant helpers (``gen_oscillating_masses``, ``c2d``, ``dlqr``, ``example_OscMass``) -
  reference ``+sp_utils/gen_oscillating_masses.m:28-59``, ``+sp_utils/example_OscMass.m:14-57``;
  MATLAB's Control-Toolbox ``c2d`` (ZOH) and ``dlqr`` are closed source, their published
  definitions (matrix exponential of the augmented matrix, stabilising DARE solution) are used.
* sparse-format helpers (``full2CSR``, ``full2CSC``, ``full2LDL``, ``LDLsolve``, ``smv``) and
  projections (``proj_SOC``, ``proj_SSOC``, ``proj_D``) - reference ``+sp_utils/*.m``
  (line numbers in each docstring).  Indices are 0-based here (the reference is 1-based and
  subtracts one when it prints the C constants, e.g. ``cons_ellipMPC_ADMM_soc_C.m:98-110``).
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import scipy.linalg as sla


# --------------------------------------------------------------------------- plants
def gen_oscillating_masses(M, K, F):
    """Continuous-time chain of ``p`` masses joined by ``p+1`` springs.

    State = [positions; velocities].  Mirrors ``+sp_utils/gen_oscillating_masses.m:28-59``
    (returns the (A, B) pair instead of a Control-Toolbox ``ss`` object).
    """
    M = np.asarray(M, dtype=float).ravel()
    K = np.asarray(K, dtype=float).ravel()
    F = np.asarray(F).ravel().astype(bool)
    p = M.size
    if K.size != p + 1 or F.size != p:
        raise ValueError("gen_oscillating_masses: need len(K) == len(M)+1 and len(F) == len(M)")
    Av = np.zeros((p, p))
    for i in range(p):
        Av[i, i] = -(K[i] + K[i + 1])
        if i > 0:
            Av[i, i - 1] = K[i]
        if i < p - 1:
            Av[i, i + 1] = K[i + 1]
    Av /= M[:, None]
    A = np.block([[np.zeros((p, p)), np.eye(p)], [Av, np.zeros((p, p))]])
    B = np.vstack([np.zeros((p, p)), np.diag(1.0 / M)])[:, F]
    return A, B


def c2d(A, B, Ts):
    """Zero-order-hold discretisation: ``expm([[A, B], [0, 0]] * Ts)`` (what ``c2d(ss, Ts)`` does)."""
    A = np.asarray(A, dtype=float)
    B = np.asarray(B, dtype=float)
    n, m = B.shape
    aug = np.zeros((n + m, n + m))
    aug[:n, :n] = A
    aug[:n, n:] = B
    E = sla.expm(aug * Ts)
    return E[:n, :n].copy(), E[:n, n:].copy()


def dlqr(A, B, Q, R):
    """Discrete LQR: returns ``(K, P)`` with ``P`` the stabilising DARE solution (MATLAB ``[K, P] = dlqr``)."""
    P = sla.solve_discrete_are(A, B, Q, R)
    K = np.linalg.solve(R + B.T @ P @ B, B.T @ P @ A)
    return K, P


def oscillating_masses_sys(p=3, Ts=0.2):
    """Discrete-time benchmark plant with the bound pattern of ``tests/spcies_tester.m:90-111``.

    ``p=3`` is the reference's test/tutorial plant; larger ``p`` follows the same pattern
    (masses alternating 1 / 0.5, all springs 2, forces on the first and last mass).
    """
    M = np.array([1.0 if i % 2 == 0 else 0.5 for i in range(p)])
    K = 2.0 * np.ones(p + 1)
    F = np.zeros(p, dtype=bool)
    F[0] = True
    F[-1] = True
    Ac, Bc = gen_oscillating_masses(M, K, F)
    A, B = c2d(Ac, Bc, Ts)
    n, m = B.shape
    return SimpleNamespace(
        A=A, B=B,
        LBx=-np.concatenate([np.ones(p), 1000.0 * np.ones(p)]),
        UBx=np.concatenate([0.3 * np.ones(p), 1000.0 * np.ones(p)]),
        LBu=-0.8 * np.ones(m), UBu=0.8 * np.ones(m),
        p=p, n=n, m=m,
    )


def example_OscMass():
    """``[sys, param] = sp_utils.example_OscMass()`` (``+sp_utils/example_OscMass.m:14-57``)."""
    sys = oscillating_masses_sys(3, 0.2)
    sys.x0 = np.zeros(sys.n)
    sys.u0 = np.zeros(sys.m)
    sys.Nx = np.ones(sys.n)
    sys.Nu = np.ones(sys.m)
    Q = np.diag(np.concatenate([15.0 * np.ones(3), np.ones(3)]))
    R = 0.1 * np.eye(sys.m)
    _, T = dlqr(sys.A, sys.B, Q, R)
    return sys, SimpleNamespace(Q=Q, R=R, T=T, N=10)


# --------------------------------------------------------------------------- sparse formats
def full2CSR(M, threshold=0.0):
    """Dense -> CSR ``(val, col, row_ptr, nnz, nrow, ncol)``, 0-based (``+sp_utils/full2CSR.m:28-62``).

    Entries with ``abs(x) <= threshold`` are dropped; an empty row simply repeats the row pointer.
    """
    M = np.asarray(M, dtype=float)
    nrow, ncol = M.shape
    val, col, row = [], [], [0]
    for i in range(nrow):
        nz = np.nonzero(np.abs(M[i]) > threshold)[0]
        val.extend(M[i, nz].tolist())
        col.extend(nz.tolist())
        row.append(len(val))
    return (np.asarray(val), np.asarray(col, dtype=np.int32), np.asarray(row, dtype=np.int32),
            len(val), nrow, ncol)


def full2CSC(M, threshold=0.0):
    """Dense -> CSC ``(val, row, col_ptr, nnz, nrow, ncol)``, 0-based (``+sp_utils/full2CSC.m:25-44``)."""
    M = np.asarray(M, dtype=float)
    nrow, ncol = M.shape
    val, row, col = [], [], [0]
    for j in range(ncol):
        nz = np.nonzero(np.abs(M[:, j]) > threshold)[0]
        val.extend(M[nz, j].tolist())
        row.extend(nz.tolist())
        col.append(len(val))
    return (np.asarray(val), np.asarray(row, dtype=np.int32), np.asarray(col, dtype=np.int32),
            len(val), nrow, ncol)


def full2LDL(M, threshold=0.0):
    """LDL' of a positive-definite matrix via Cholesky (``+sp_utils/full2LDL.m:16-57``).

    Returns ``(L_val, L_row, L_col, Dinv)``: CSC of ``L - I`` and the inverted diagonal of ``D``.
    """
    M = np.asarray(M, dtype=float)
    C = np.linalg.cholesky(M)  # lower
    d = np.diag(C)
    L = C / d[None, :]
    Dinv = 1.0 / (d * d)
    Lv, Lr, Lc, *_ = full2CSC(L - np.eye(M.shape[0]), threshold)
    return Lv, Lr, Lc, Dinv


def LDLsolve(L_val, L_row, L_col, Dinv, b):
    """Solve ``L D L' x = b`` with the CSC factor (``+sp_utils/LDLsolve.m:22-49``)."""
    x = np.array(b, dtype=float).copy()
    n = x.size
    for j in range(n):
        for p in range(L_col[j], L_col[j + 1]):
            x[L_row[p]] -= L_val[p] * x[j]
    x *= Dinv
    for j in range(n - 1, -1, -1):
        for p in range(L_col[j], L_col[j + 1]):
            x[j] -= L_val[p] * x[L_row[p]]
    return x


def smv(val, col, row, x):
    """CSR sparse matrix-vector product (``+sp_utils/smv.m:23-36``)."""
    y = np.zeros(len(row) - 1)
    for i in range(len(row) - 1):
        for p in range(row[i], row[i + 1]):
            y[i] += val[p] * x[col[p]]
    return y


# --------------------------------------------------------------------------- projections
def proj_SOC(x):
    """Projection onto ``{(x0, x1): ||x1|| <= x0}`` (``+sp_utils/proj_SOC.m:12-27``)."""
    x = np.asarray(x, dtype=float)
    x0, x1 = x[0], x[1:]
    nx = np.linalg.norm(x1)
    if nx <= x0:
        return x.copy()
    if nx <= -x0:
        return np.zeros_like(x)
    return 0.5 * (x0 + nx) * np.concatenate([[1.0], x1 / nx])


def proj_SSOC(x, alpha, d):
    """Projection onto the shifted cone ``||x1|| <= alpha (x0 - d)`` (``+sp_utils/proj_SSOC.m:14-29``)."""
    x = np.asarray(x, dtype=float)
    x0, x1 = x[0], x[1:]
    nx = np.linalg.norm(x1)
    s = alpha * (x0 - d)
    shift = np.concatenate([[d], np.zeros(x1.size)])
    if nx <= s:
        return x.copy()
    if nx <= -s:
        return shift
    return 0.5 * (s + nx) * np.concatenate([[alpha], x1 / nx]) + shift


def proj_D(x, LB, UB):
    """Projection onto the 'diamond' ``LB + ||x1|| <= x0 <= UB - ||x1||`` (``+sp_utils/proj_D.m:19-23``)."""
    return proj_SSOC(proj_SSOC(x, 1.0, LB), -1.0, UB)
