"""Shared by the CPU and the GPU tests: random controllers and the parity bars of the non-bit-exact variants.

The bars.  SURVEY 8c: 1e-10 on u, z, v against the oracle (the reference's `tol_spcies`, spcies_tester.m:260).  Two families of
instances cannot meet a FLAT 1e-10 in ANY operation order other than the oracle's own, and the number that says so is measured
on the oracle itself (tests/test_oracle_conditioning.py, CPU, runs every round):

* equMPC instances whose terminal equality is unreachable (ADMM / FISTA on an infeasible QP: e_flag = -1, multipliers |lambda| up to
  4e7).  Feeding the ORACLE the constants the reference's generator would print for it (`%1.15f`, platforms/+C_code/dec_var.m)
  instead of full doubles moves its z by up to 8e-14 |lambda| (9.7e-9 absolute on the C2_equ batch); one ulp on Beta and Hi alone
  moves it by 3e-15 |lambda| (5e-10).  The bar for such instances is LAMBDA_BAR_COEFF |lambda| = 2.5e-13 |lambda| - about three
  times what the reference's own generated solver differs from its own exact-constant self - and the flat 1e-10 below |lambda| = 400.
* MPCT ADMM cs at the C4 shape (cond(W) = 1e9): the oracle moves by 8e-9 under the same print quantisation; bar CS_ILL_BAR = 1e-7.
"""
from types import SimpleNamespace

import numpy as np

TOL_SPCIES = 1e-10
LAMBDA_BAR_COEFF = 2.5e-13       # bar on u, z, v of an instance = max(TOL_SPCIES, LAMBDA_BAR_COEFF * max|lambda| of the instance)
LAMBDA_FLAT_BELOW = TOL_SPCIES / LAMBDA_BAR_COEFF  # = 400: below this |lambda| the flat bar applies
CS_ILL_BAR = 1e-7                # MPCT ADMM cs at the C4 shape (cond(W) = 1e9), re-ordered sums


def scaled_bar(lscale):
    """Per-instance bar on u, z, v: `lscale` = max |lambda| of the instance (any array shape)."""
    return TOL_SPCIES * np.maximum(1.0, np.asarray(lscale, dtype=float) / LAMBDA_FLAT_BELOW)


def random_cfg(n, m, N, seed):
    """A random stable plant with box constraints that contain the origin and a dense terminal weight."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, n))
    A *= 0.95 / max(abs(np.linalg.eigvals(A)))  # stable, well inside the unit circle
    Bm = rng.standard_normal((n, m))
    sys = SimpleNamespace(A=A, B=Bm, n=n, m=m, LBx=-1.0 - rng.random(n), UBx=1.0 + rng.random(n), LBu=-0.5 - rng.random(m),
                          UBu=0.5 + rng.random(m))
    param = SimpleNamespace(Q=np.diag(1.0 + 4 * rng.random(n)), R=np.diag(0.1 + rng.random(m)), N=N)
    M = rng.standard_normal((n, n))
    param.T = np.diag(np.diag(param.Q)) * 3 + 0.1 * (M @ M.T)  # dense terminal weight
    return SimpleNamespace(name=f"rand_{n}_{m}_{N}", sys=sys, param=param, formulation="laxMPC", method="ADMM",
                           solver_options=dict(rho=8.0, tol=1e-6, k_max=400), B=1, seed=seed)


def one_ulp(a, rng):
    """Every entry of `a` moved to a neighbouring double, direction at random."""
    a = np.array(a, dtype=float, copy=True)
    s = rng.integers(0, 2, a.shape) * 2 - 1
    return np.nextafter(a, np.where(s > 0, np.inf, -np.inf))


def assert_k(k, ko, rerun=None, scale=1.0, max_share=1e-3, what=""):
    """`k` must equal the oracle's `ko`.  SURVEY 8c tolerates a difference of ONE iteration on at most 0.1 % of the instances "when
    the deciding residual is within 1e-12 of tol" - checked, not assumed: `rerun(idx, dtol)` re-runs the ORACLE on the instances
    `idx` with its exit tolerance(s) shifted by `dtol` and returns its `k`; an instance whose GPU `k` differs must be reproduced by
    the oracle with the tolerance moved by +-1e-12 * scale (`scale`: the instance's bar scale, 1 for well-conditioned ones) - i.e. its
    exit test was decided within that margin.  Without `rerun` no difference is accepted at all.  Returns the mask of equal k."""
    k, ko = np.asarray(k).astype(int), np.asarray(ko).astype(int)
    dk = np.abs(k - ko)
    if not (dk > 0).any():
        return dk == 0
    assert dk.max() <= 1, f"{what}: k differs by {dk.max()}"
    assert (dk > 0).mean() <= max_share + (1.0 / len(k) if len(k) < 1000 else 0.0), f"{what}: k differs on {(dk > 0).sum()} of {len(k)} instances"
    assert rerun is not None, f"{what}: k differs on {(dk > 0).sum()} instance(s) and the test has no residual witness"
    idx = np.nonzero(dk)[0]
    sc = np.broadcast_to(np.asarray(scale, dtype=float).reshape(-1) if np.ndim(scale) else float(scale), (len(k),))[idx]
    for i, s_i in zip(idx, sc):
        eps = 1e-12 * float(s_i)
        # GPU left one iteration EARLIER: its residual was <= tol where the oracle's was just above -> the oracle with tol + eps agrees;
        # one iteration LATER: the oracle with tol - eps agrees
        dtol = eps if k[i] < ko[i] else -eps
        k_shift = int(np.asarray(rerun(np.array([i]), dtol)).reshape(-1)[0])
        assert k_shift == k[i], (f"{what}: instance {i}: k = {k[i]} against the oracle's {ko[i]}, and moving the oracle's tolerance by "
                                 f"{dtol:+.1e} gives {k_shift}: the exit test was not decided within rounding")
    return dk == 0
