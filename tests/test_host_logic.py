"""CPU: host-side logic - options precedence, dispatch, blob round trip, sp_utils numerics."""
import numpy as np
import pytest
import scipy.sparse as sp

from spcies_amd import benchmarks, blob, sp_utils
from spcies_amd.gen_controller import spcies_gen_controller
from spcies_amd.options import SpciesOptions


def test_options_defaults_and_precedence():
    o = SpciesOptions(formulation="laxMPC")
    assert o.method == "ADMM" and o.submethod == "" and o.platform == "HIP"
    assert o.solver["rho"] == 1e-2 and o.solver["tol"] == 1e-4 and o.solver["k_max"] == 1000  # def_options_laxMPC_ADMM.m
    o = SpciesOptions(formulation="laxMPC", options=dict(rho=15, tol=1e-7, debug=False), debug=True)
    assert o.solver["rho"] == 15 and o.solver["k_max"] == 1000
    assert o.debug is True  # explicit name-value beats the options struct
    assert SpciesOptions(formulation="MPCT").method == "EADMM"
    assert SpciesOptions(formulation="HMPC", method="SADMM").submethod == "split"
    with pytest.warns(DeprecationWarning):
        o = SpciesOptions(type="equMPC", solver_options=dict(rho=3.0))  # deprecated aliases the reference tests use
    assert o.formulation == "equMPC" and o.solver["rho"] == 3.0
    with pytest.raises(ValueError):
        SpciesOptions(formulation="nope")
    with pytest.raises(ValueError):
        SpciesOptions(formulation="laxMPC", platform="Arduino")
    d = SpciesOptions(formulation="laxMPC").default_defines()
    assert d == {"DEBUG": 1, "MEASURE_TIME": 1, "in_engineering": 0, "TIME_VARYING": 0, "IS_DIAG": 1}


def test_gen_controller_dispatch_errors():
    cfg = benchmarks.config("C1")
    with pytest.raises(ValueError):
        spcies_gen_controller(param=cfg.param, formulation="laxMPC")
    with pytest.raises(ValueError):
        spcies_gen_controller(sys=cfg.sys, param=cfg.param)  # no formulation
    with pytest.raises(NotImplementedError):
        spcies_gen_controller(sys=cfg.sys, param=cfg.param, formulation="laxMPC", platform="C")
    with pytest.raises(ValueError):
        spcies_gen_controller(sys=cfg.sys, param=cfg.param, formulation="laxMPC", method="EADMM")


def test_non_diagonal_weights_rejected():
    cfg = benchmarks.config("C1")
    cfg.param.Q = cfg.param.Q + 0.1
    with pytest.raises(ValueError, match="non_diagonal"):
        benchmarks.ingredients(cfg)


@pytest.mark.parametrize("name", ["C1_lax", "C1_equ", "C2_lax"])
def test_ingredients_factor_W(name):
    """Alpha/Beta must be the block-bidiagonal Cholesky factor of W = G Hhat^-1 G'."""
    from spcies_amd.formulations.laxMPC import build_G
    cfg = benchmarks.config(name)
    v = benchmarks.ingredients(cfg)
    n, m, N = v["n"], v["m"], v["N"]
    Wc = np.zeros((N * n, N * n))
    for l in range(N):
        blk = np.triu(v["Beta"][l]).copy()
        blk[np.diag_indices(n)] = 1.0 / np.diag(blk)
        Wc[l * n:(l + 1) * n, l * n:(l + 1) * n] = blk
        if l < N - 1:
            Wc[l * n:(l + 1) * n, (l + 1) * n:(l + 2) * n] = v["Alpha"][l]
    G = build_G(cfg.sys.A, cfg.sys.B, N, terminal=v["terminal"])
    hd = np.concatenate([v["Hi_0"], v["Hi"].ravel()])
    Hinv = np.diag(hd)
    if v["terminal"]:
        Hinv = np.block([[Hinv, np.zeros((hd.size, n))], [np.zeros((n, hd.size)), v["Hi_N"]]])
    W = G @ Hinv @ G.T
    assert np.abs(Wc.T @ Wc - W).max() < 1e-12


def test_blob_round_trip_and_inf_bounds():
    cfg = benchmarks.config("C2")
    cfg.sys.UBx = cfg.sys.UBx.copy()
    cfg.sys.UBx[-1] = np.inf
    v = benchmarks.ingredients(cfg)
    b = blob.pack(v)
    assert len(b) % 64 == 0
    w = blob.unpack(b)
    assert (w["n"], w["m"], w["N"], w["k_max"]) == (12, 2, 15, 200) and w["tol"] == 0.0
    for key in ("AB", "Alpha", "Beta", "Hi", "Hi_0", "Hi_N", "Q", "R", "T", "LB"):
        assert np.array_equal(w[key], v[key])
    assert w["UB"][11] == 1e20  # +inf -> 1e20 as dec_var.m:245-248
    with pytest.raises(ValueError):
        blob.unpack(b[:-1])


def test_sparse_helpers_match_scipy():
    rng = np.random.default_rng(0)
    M = rng.standard_normal((7, 5)) * (rng.random((7, 5)) < 0.4)
    val, col, row, nnz, nr, nc = sp_utils.full2CSR(M)
    ref = sp.csr_matrix(M)
    assert np.array_equal(val, ref.data) and np.array_equal(col, ref.indices) and np.array_equal(row, ref.indptr)
    val, rw, cp, *_ = sp_utils.full2CSC(M)
    ref = sp.csc_matrix(M)
    assert np.array_equal(val, ref.data) and np.array_equal(rw, ref.indices) and np.array_equal(cp, ref.indptr)
    x = rng.standard_normal(5)
    v_, c_, r_, *_ = sp_utils.full2CSR(M)
    assert np.allclose(sp_utils.smv(v_, c_, r_, x), M @ x)
    A = rng.standard_normal((6, 6))
    S = A @ A.T + 6 * np.eye(6)
    Lv, Lr, Lc, Dinv = sp_utils.full2LDL(S)
    b = rng.standard_normal(6)
    assert np.allclose(sp_utils.LDLsolve(Lv, Lr, Lc, Dinv, b), np.linalg.solve(S, b))


def test_projections():
    rng = np.random.default_rng(1)
    for _ in range(200):
        x = rng.standard_normal(4) * 2
        z = sp_utils.proj_SOC(x)
        assert np.linalg.norm(z[1:]) <= z[0] + 1e-12
        assert np.allclose(sp_utils.proj_SOC(z), z)  # idempotent
        y = rng.standard_normal(4)
        y[0] = np.linalg.norm(y[1:]) + abs(y[0])  # a point of the cone
        assert (x - z) @ (y - z) <= 1e-9  # projection inequality
        zs = sp_utils.proj_SSOC(x, -1.0, 0.7)
        assert np.linalg.norm(zs[1:]) <= -(zs[0] - 0.7) + 1e-12
    d = sp_utils.proj_D(np.array([0.5, 3.0, 0.0]), 0.0, 1.0)
    assert d[0] - 0.0 >= np.linalg.norm(d[1:]) - 1e-12 and 1.0 - d[0] >= np.linalg.norm(d[1:]) - 1e-12
