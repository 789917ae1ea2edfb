"""Time-varying lax/equ MPC ADMM and FISTA (TIME_VARYING == 1 in
formulations/+laxMPC/code_laxMPC_ADMM_C.c:117-279, code_equMPC_ADMM_C.c:117-265, tutorial
examples/t01_time_varying_MPC.m): the model (A, B, Q, R, LB, UB) arrives with every call and the banded
Cholesky factors are computed on line.

Oracle pin (CPU): the on-line factors equal the off-line ones of compute_laxMPC_ADMM_ingredients.m (chol of
W) to rounding, and a time-varying solve handed the design model reproduces the ordinary solver - hence the
reference tests' z_opt.  GPU: the HIP path against the oracle, bit for bit (STREAM operation order), with one
model per instance.
"""
import json
import os

import numpy as np
import pytest


def _setup(name):
    from spcies_amd import benchmarks
    cfg = benchmarks.config(name)
    v = benchmarks.ingredients(cfg)
    vt = benchmarks.ingredients(cfg, time_varying=True)
    sys, prm = cfg.sys, cfg.param
    LB = np.concatenate([np.ravel(sys.LBx), np.ravel(sys.LBu)])
    UB = np.concatenate([np.ravel(sys.UBx), np.ravel(sys.UBu)])
    return cfg, v, vt, (np.asarray(sys.A, float), np.asarray(sys.B, float), np.diag(prm.Q).copy(), np.diag(prm.R).copy(), LB, UB)


def _perturbed_models(design, B, seed=5):
    """One model per instance: the design model with every entry of A, B and the weights moved by up to 2 %,
    bounds by up to 5 %."""
    rng = np.random.default_rng(seed)
    A, Bm, Q, R, LB, UB = design
    j = lambda a, s: a[None] * (1.0 + s * (2 * rng.random((B,) + a.shape) - 1))
    return j(A, 0.02), j(Bm, 0.02), j(Q, 0.02), j(R, 0.02), j(LB, 0.05), j(UB, 0.05)


@pytest.mark.parametrize("name,test_name", [("C1_lax", "test_laxMPC_ADMM"), ("C1_equ", "test_equMPC_ADMM"), ("C2_lax", None)])
def test_oracle_time_varying_matches_offline(name, test_name, golden_dir):
    from oracle import oracle
    from spcies_amd import benchmarks
    cfg, v, vt, design = _setup(name)
    assert vt["time_varying"] and "Alpha" not in vt and vt["T_rho_i"].shape == (cfg.sys.n, cfg.sys.n)
    model, per = oracle.pack_tv_model(*design)
    x0, xr, ur = benchmarks.sample_batch(cfg, 12)
    if test_name:
        st = benchmarks.tester_status(cfg.sys)
        x0[0], xr[0], ur[0] = st.x, st.xr, st.ur
    out = oracle.admm_tv_batch(vt, x0, xr, ur, model, per, want_factors=True)
    ref = oracle.admm_banded_batch(v, x0, xr, ur)
    assert np.abs(out[6] - v["Alpha"]).max() <= 1e-12
    assert np.abs(np.triu(out[7]) - np.triu(v["Beta"])).max() <= 1e-11
    assert np.array_equal(out[1], ref[1]) and np.array_equal(out[2], ref[2])
    from _cases import TOL_SPCIES, scaled_bar  # (the measured |lambda| allowance: tests/test_oracle_conditioning.py)
    lscale = scaled_bar(np.abs(ref[5]).max(axis=1, keepdims=True)) / TOL_SPCIES
    assert (np.abs(out[0] - ref[0]) / lscale).max() <= 1e-10 and (np.abs(out[3] - ref[3]) / lscale).max() <= 1e-9
    if test_name:
        with open(os.path.join(golden_dir, "reference_z_opt.json")) as f:
            z_opt = np.array(json.load(f)[test_name])
        assert out[2][0] == 1 and np.abs(out[3][0] - z_opt).max() <= 1e-4


def test_blob_roundtrip_time_varying():
    from spcies_amd import blob
    cfg, v, vt, design = _setup("C1_lax")
    b = blob.pack(vt)
    w = blob.unpack(b)
    assert w["time_varying"] and np.array_equal(w["T_rho_i"], vt["T_rho_i"]) and "AB" not in w
    assert not blob.unpack(blob.pack(v))["time_varying"]


def _compare_tv(variant, got, O, vt, x0, xr, ur, model, per):
    """STREAM and TILE run the reference's operation order: bit for bit.  MFMA4R (factors in registers, explicit Beta^-1, w-form) re-associates
    sums: 1e-10 on u, z, v (the measured |lambda| allowance of tests/_cases.py for instances whose multipliers blow up), k equal unless the
    oracle itself flips when its tolerance moves by 1e-12."""
    from oracle import oracle
    from _cases import TOL_SPCIES, assert_k, scaled_bar
    u, k, e, sol = got
    if variant != "mfma4r":
        assert np.array_equal(k, O[1]) and np.array_equal(e, O[2]) and np.array_equal(u, O[0])
        if sol.z is not None:
            assert np.array_equal(sol.z, O[3]) and np.array_equal(sol.v, O[4]) and np.array_equal(sol.lam, O[5])
        return

    def rerun(idx, dtol):
        v2 = dict(vt)
        v2["tol"] = float(vt["tol"]) + dtol
        pr = np.ndim(xr) == 2
        return oracle.admm_tv_batch(v2, x0[idx], xr[idx] if pr else xr, ur[idx] if pr else ur, model[idx] if per else model, per, want_sol=False)[1]
    lscale = np.abs(O[5]).max(axis=1, keepdims=True)
    bar = scaled_bar(lscale)
    same = assert_k(k, O[1], rerun, scale=bar / TOL_SPCIES, what="time-varying MFMA4R")
    assert np.array_equal(e[same], O[2][same])
    assert (np.abs(u - O[0]) / bar)[same].max() <= 1.0
    if sol.z is not None:
        assert (np.abs(sol.z - O[3]) / bar)[same].max() <= 1.0 and (np.abs(sol.v - O[4]) / bar)[same].max() <= 1.0
        assert (np.abs(sol.lam - O[5]) / (bar * (1.0 + lscale)))[same].max() <= 1.0


@pytest.mark.gpu
# mfma4r: the factors in REGISTERS, explicit inverses on the matrix pipe (admm_tvr.hpp) - 1e-10, AUTO's choice where its kernel is built
@pytest.mark.parametrize("variant", ["stream", "mfma4r"])
@pytest.mark.parametrize("name,B,overrides", [("C1_lax", 70, {}), ("C1_equ", 40, {}), ("C2_lax", 130, {}),
                                              ("C2_lax", 50, dict(tol=1e-6, k_max=3000)), ("C2_equ", 45, {})])
def test_hip_time_varying_vs_oracle(name, B, overrides, variant):
    from oracle import oracle
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver, SpciesArgError
    cfg, v, vt, design = _setup(name)
    vt = benchmarks.ingredients(cfg, time_varying=True, **overrides)
    s = HipSolver(vt)
    assert s.time_varying and s.variant == "mfma4r"  # AUTO
    s.set_variant(variant)
    assert s.variant == variant
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    with pytest.raises(SpciesArgError):
        s(x0, xr, ur)  # nine inputs are required (struct_laxMPC_ADMM_C_Matlab.c:29-31)
    # one shared model (the design model)
    model, per = oracle.pack_tv_model(*design)
    O = oracle.admm_tv_batch(vt, x0, xr, ur, model, per)
    _compare_tv(variant, s(x0, xr, ur, *design), O, vt, x0, xr, ur, model, per)
    # one model per instance
    models = _perturbed_models(design, B)
    model, per = oracle.pack_tv_model(*models)
    O = oracle.admm_tv_batch(vt, x0, xr, ur, model, per)
    full = s(x0, xr, ur, *models)
    _compare_tv(variant, full, O, vt, x0, xr, ur, model, per)
    nosol = s(x0[:9], xr[:9], ur[:9], *[a[:9] for a in models], want_sol=False)  # the no-record kernel: the same bits
    assert np.array_equal(nosol[0], full[0][:9]) and np.array_equal(nosol[1], full[1][:9])
    # ragged batch sizes around the four instances of a workgroup, and one reference for the whole batch
    for Bs in (1, 3, 5):
        part = s(x0[:Bs], xr[0], ur[0], *[a[:Bs] for a in models])
        Os = oracle.admm_tv_batch(vt, x0[:Bs], xr[0], ur[0], model[:Bs], per)
        _compare_tv(variant, part, Os, vt, x0[:Bs], xr[0], ur[0], model[:Bs], per)
    s.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,N", [("C1_lax", 4), ("C1_equ", 13), ("C1_lax", 23), ("C2_lax", 8), ("C2_equ", 9)])  # (equMPC needs N m >= n: with fewer inputs than states the terminal equality makes W singular - the oracle itself returns NaN)
def test_time_varying_mfma4r_other_horizons(name, N):
    """MFMA4R for a horizon without a build-time kernel: specialised with hiprtc at create time (the register arrays are indexed by the
    unrolled horizon), AUTO's choice; 1e-10 against the oracle with one model per instance."""
    import copy
    from types import SimpleNamespace
    from oracle import oracle
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver
    cfg = copy.copy(benchmarks.config(name))
    cfg.param = SimpleNamespace(**dict(vars(cfg.param), N=N))
    vt = benchmarks.ingredients(cfg, time_varying=True, tol=1e-6, k_max=600)
    sysm, prm = cfg.sys, cfg.param
    LB = np.concatenate([np.ravel(sysm.LBx), np.ravel(sysm.LBu)])
    UB = np.concatenate([np.ravel(sysm.UBx), np.ravel(sysm.UBu)])
    design = (np.asarray(sysm.A, float), np.asarray(sysm.B, float), np.diag(prm.Q).copy(), np.diag(prm.R).copy(), LB, UB)
    with HipSolver(vt) as s:
        assert s.variant == "mfma4r", s.notes
        B = 21
        x0, xr, ur = benchmarks.sample_batch(cfg, B)
        models = _perturbed_models(design, B)
        model, per = oracle.pack_tv_model(*models)
        O = oracle.admm_tv_batch(vt, x0, xr, ur, model, per)
        _compare_tv("mfma4r", s(x0, xr, ur, *models), O, vt, x0, xr, ur, model, per)
        s.set_variant("stream")  # the bit-exact variant on the same handle
        _compare_tv("stream", s(x0, xr, ur, *models), O, vt, x0, xr, ur, model, per)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,N,formulation,method", [(5, 3, 6, "laxMPC", "ADMM"), (9, 2, 8, "laxMPC", "ADMM"), (10, 4, 7, "equMPC", "ADMM"),
                                                      (13, 3, 5, "laxMPC", "ADMM"), (4, 2, 9, "laxMPC", "FISTA"), (10, 4, 7, "laxMPC", "FISTA"),
                                                      (8, 3, 6, "equMPC", "FISTA")])
def test_time_varying_any_plant_size(n, m, N, formulation, method):
    """The 9-input solvers of a plant whose (n, m) has no build-time kernel: the whole MFMA4R path - update phase (tv_update_kernel.inc,
    the text the build-time instantiations compile), inverses, solve - is specialised with hiprtc at create time; one model per
    instance against the oracle; STREAM by name runs the bit-exact pair specialised for the same (n, m)."""
    from oracle import oracle
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver
    from _cases import random_cfg
    cfg = random_cfg(n, m, N, seed=1300 + n)
    cfg.formulation, cfg.method = formulation, method
    cfg.param.T = np.diag(np.diag(cfg.param.T))  # (FISTA takes a diagonal terminal weight)
    if method == "FISTA":
        cfg.solver_options = dict(tol=1e-6, k_max=400)
    vt = benchmarks.ingredients(cfg, time_varying=True)
    sysm, prm = cfg.sys, cfg.param
    LB = np.concatenate([np.ravel(sysm.LBx), np.ravel(sysm.LBu)])
    UB = np.concatenate([np.ravel(sysm.UBx), np.ravel(sysm.UBu)])
    design = (np.asarray(sysm.A, float), np.asarray(sysm.B, float), np.diag(prm.Q).copy(), np.diag(prm.R).copy(), LB, UB)
    rng = np.random.default_rng(17 * n + m)
    B = 23
    x0 = (0.6 if method == "FISTA" else 0.4) * rng.standard_normal((B, n))
    xr = 0.1 * rng.standard_normal((B, n))
    ur = 0.05 * rng.standard_normal((B, m))
    models = _perturbed_models(design, B)
    model, per = oracle.pack_tv_model(*models)
    with HipSolver(vt) as s:
        assert s.time_varying and s.variant == "mfma4r", s.notes
        if method == "FISTA":
            O = oracle.fista_tv_batch(vt, x0, xr, ur, model, per)
            _compare_tv_fista("mfma4r", s(x0, xr, ur, *models), O, vt, x0, xr, ur, model, per)
        else:
            O = oracle.admm_tv_batch(vt, x0, xr, ur, model, per)
            _compare_tv("mfma4r", s(x0, xr, ur, *models), O, vt, x0, xr, ur, model, per)
        nosol = s(x0[:7], xr[:7], ur[:7], *[a[:7] for a in models], want_sol=False)
        assert np.abs(nosol[0] - O[0][:7]).max() <= 1e-9 and np.abs(nosol[1].astype(int) - O[1][:7].astype(int)).max() <= 1
        # STREAM by name: the bit-exact pair (update phase + iteration) specialised for this (n, m) as well (round 5; refused before)
        s.set_variant("stream")
        assert s.variant == "stream"
        if method == "FISTA":
            _compare_tv_fista("stream", s(x0, xr, ur, *models), O, vt, x0, xr, ur, model, per)
        else:
            _compare_tv("stream", s(x0, xr, ur, *models), O, vt, x0, xr, ur, model, per)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,N,formulation,method", [(20, 4, 6, "laxMPC", "ADMM"), (12, 6, 5, "equMPC", "ADMM"), (17, 3, 6, "equMPC", "FISTA"),
                                                      (20, 2, 5, "laxMPC", "FISTA"),
                                                      (6, 2, 40, "laxMPC", "ADMM"),    # n + m <= 16 but 40 blocks of factors: past the registers, 35 KB of LDS
                                                      (18, 3, 7, "equMPC", "ADMM"),    # n, n + m not multiples of 4: the masked last k-slab, two ragged row groups
                                                      (21, 3, 4, "laxMPC", "ADMM")])   # 21 lanes per instance in the cooperative update phase, all of them columns; n past the merged-tile form
def test_time_varying_plants_past_the_register_file(n, m, N, formulation, method, monkeypatch):
    """The 9-input solvers of plants the register-resident solver does not hold (n + m > 16: the 20-state plant of configs[3] among them).  Rounds 2-4
    answered ENOSUP here.  STREAM - update phase and iteration specialised with hiprtc for this (n, m), the update phase in its rolled form past
    n = 16 (tv_band_factor_rolled) - bit for bit against the oracle, one model per instance and one shared model.  AUTO is MFMA4R in its LDS form
    (admm_tvl_kernel.inc: one wavefront per instance, the instance's factors in the LDS, stage vectors of two registers; ADMM and FISTA), 1e-10."""
    from oracle import oracle
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver
    from _cases import random_cfg
    cfg = random_cfg(n, m, N, seed=1400 + n)
    cfg.formulation, cfg.method = formulation, method
    cfg.param.T = np.diag(np.diag(cfg.param.T))
    if method == "FISTA":
        cfg.solver_options = dict(tol=1e-6, k_max=400)
    vt = benchmarks.ingredients(cfg, time_varying=True)
    sysm, prm = cfg.sys, cfg.param
    LB = np.concatenate([np.ravel(sysm.LBx), np.ravel(sysm.LBu)])
    UB = np.concatenate([np.ravel(sysm.UBx), np.ravel(sysm.UBu)])
    design = (np.asarray(sysm.A, float), np.asarray(sysm.B, float), np.diag(prm.Q).copy(), np.diag(prm.R).copy(), LB, UB)
    rng = np.random.default_rng(19 * n + m)
    B = 70  # (two wavefronts, the second partly filled)
    x0 = (0.6 if method == "FISTA" else 0.4) * rng.standard_normal((B, n))
    xr = 0.1 * rng.standard_normal((B, n))
    ur = 0.05 * rng.standard_normal((B, m))
    models = _perturbed_models(design, B)
    model, per = oracle.pack_tv_model(*models)
    fn, cmp = (oracle.fista_tv_batch, _compare_tv_fista) if method == "FISTA" else (oracle.admm_tv_batch, _compare_tv)
    with HipSolver(vt) as s:
        auto = "mfma4r"  # the LDS form, ADMM and FISTA
        assert s.time_varying and s.variant == auto, (s.variant, s.notes)
        O = fn(vt, x0, xr, ur, model, per)
        shared, per1 = oracle.pack_tv_model(*design)
        O1 = fn(vt, x0, xr, ur, shared, per1)
        full = s(x0, xr, ur, *models)
        cmp("mfma4r", full, O, vt, x0, xr, ur, model, per)
        nosol = s(x0[:9], xr[:9], ur[:9], *[a[:9] for a in models], want_sol=False)  # the record-free instantiation: the same bits (no contraction in either)
        assert np.array_equal(nosol[0], full[0][:9]) and np.array_equal(nosol[1], full[1][:9])
        cmp("mfma4r", s(x0, xr, ur, *design), O1, vt, x0, xr, ur, shared, per1)
        s.set_variant("stream")
        cmp("stream", s(x0, xr, ur, *models), O, vt, x0, xr, ur, model, per)
        nosol = s(x0[:9], xr[:9], ur[:9], *[a[:9] for a in models], want_sol=False)
        assert np.array_equal(nosol[0], O[0][:9]) and np.array_equal(nosol[1], O[1][:9])
        cmp("stream", s(x0, xr, ur, *design), O1, vt, x0, xr, ur, shared, per1)
    # the update phase of the LDS form is cooperative (tv_update_coop_kernel: a lane per column, several instances per wavefront); every entry of its
    # factors is the sum the one-lane-per-instance kernels form, in their order - SPCIES_TVL_COOP=0 runs those instead: the same bits out of the solve
    monkeypatch.setenv("SPCIES_TVL_COOP", "0")
    with HipSolver(vt) as s1:
        assert s1.variant == "mfma4r", (s1.variant, s1.notes)
        one = s1(x0, xr, ur, *models)
    assert all(np.array_equal(a, b) for a, b in zip(full[:3], one[:3]))
    assert np.array_equal(full[3].z, one[3].z) and np.array_equal(full[3].lam, one[3].lam)
    if (n, m, N) == (20, 4, 6):  # the triangles of Bi_l instead of S_l (the chain with six products per stage pair): the documented switch stays alive
        monkeypatch.delenv("SPCIES_TVL_COOP")
        monkeypatch.setenv("SPCIES_TVR_RTC_FLAGS", "-DSPCIES_TVL_SFORM=0")
        with HipSolver(vt) as s2:
            assert s2.variant == "mfma4r", (s2.variant, s2.notes)
            cmp("mfma4r", s2(x0, xr, ur, *models), O, vt, x0, xr, ur, model, per)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["C2_lax", "C1_equ", "C2_lax_FISTA"])
def test_time_varying_register_form_takes_its_images_from_the_cooperative_update_phase(name, monkeypatch):
    """The register-resident MFMA4R kernels read S_l and M_l = S_l E from an instance-major scratch written by tv_update_coop_kernel (FORM 1: several lanes
    per instance, the reference's sums for Beta / Alpha, S and M formed in the LDS).  SPCIES_TVR_COOP=0 is the path of rounds 4-5 - one lane per instance,
    then tv_ms_kernel, [row][Bp] scratch: the same factors with M and S summed in another order, so the solves agree to rounding (and both sit inside the
    oracle's bar: test_hip_time_varying_vs_oracle runs the default)."""
    from spcies_amd.solver import HipSolver
    cfg, v, vt, design = _setup(name)
    B = 150
    rng = np.random.default_rng(77)
    n, m = cfg.sys.n, cfg.sys.m
    x0 = 0.4 * rng.standard_normal((B, n)); xr = 0.1 * rng.standard_normal((B, n)); ur = 0.05 * rng.standard_normal((B, m))
    models = _perturbed_models(design, B)
    with HipSolver(vt) as s:
        assert s.variant == "mfma4r", (s.variant, s.notes)
        coop = s(x0, xr, ur, *models)
    monkeypatch.setenv("SPCIES_TVR_COOP", "0")
    with HipSolver(vt) as s:
        assert s.variant == "mfma4r", (s.variant, s.notes)
        lane = s(x0, xr, ur, *models)
    assert np.abs(coop[0] - lane[0]).max() <= 1e-10 and np.abs(coop[3].z - lane[3].z).max() <= 1e-9  # (equMPC: multipliers of 1e3 - the scaled bar of _cases.py)
    assert (np.abs(coop[1].astype(int) - lane[1].astype(int)) <= 1).all() and (coop[1] != lane[1]).mean() <= 0.01


@pytest.mark.gpu
def test_time_varying_mfma4r_is_switched_off_by_the_environment(monkeypatch):
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver
    cfg, v, vt, design = _setup("C1_lax")
    monkeypatch.setenv("SPCIES_HIP_TVR", "0")
    with HipSolver(vt) as s:
        assert s.variant == "stream"
        with pytest.raises(Exception, match="MFMA4R"):
            s.set_variant("mfma4r")


# ----------------------------------------------------------------------------------------------
# Time-varying lax/equ MPC FISTA (TIME_VARYING == 1 in code_laxMPC_FISTA_C.c:18, 42-56, 83-271, code_equMPC_FISTA_C.c)
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name,test_name", [("C1_lax_FISTA", "test_laxMPC_FISTA"), ("C1_equ_FISTA", "test_equMPC_FISTA"),
                                            ("C2_lax_FISTA", None)])
def test_oracle_time_varying_fista_matches_offline(name, test_name, golden_dir):
    """The on-line factors equal the off-line ones (chol of W = G H^-1 G') to rounding, and a time-varying solve handed the
    design model reproduces the ordinary solver - hence the reference tests' z_opt."""
    from oracle import oracle
    from spcies_amd import benchmarks
    cfg, v, vt, design = _setup(name)
    assert vt["time_varying"] and "Alpha" not in vt and vt["Ti"].shape == (cfg.sys.n,)
    model, per = oracle.pack_tv_model(*design)
    x0, xr, ur = benchmarks.sample_batch(cfg, 12)
    if test_name:
        st = benchmarks.tester_status(cfg.sys)
        x0[0], xr[0], ur[0] = st.x, st.xr, st.ur
    out = oracle.fista_tv_batch(vt, x0, xr, ur, model, per, want_factors=True)
    ref = oracle.fista_banded_batch(v, x0, xr, ur)
    assert np.abs(out[5] - v["Alpha"]).max() <= 1e-12
    assert np.abs(np.triu(out[6]) - np.triu(v["Beta"])).max() <= 1e-11
    assert np.array_equal(out[1], ref[1]) and np.array_equal(out[2], ref[2])
    assert np.abs(out[0] - ref[0]).max() <= 1e-10 and np.abs(out[3] - ref[3]).max() <= 1e-8
    if test_name:
        with open(os.path.join(golden_dir, "reference_z_opt.json")) as f:
            z_opt = np.array(json.load(f)[test_name])
        assert out[2][0] == 1 and np.abs(out[3][0] - z_opt).max() <= 1e-4


def test_blob_roundtrip_time_varying_fista():
    from spcies_amd import blob
    cfg, v, vt, design = _setup("C1_lax_FISTA")
    w = blob.unpack(blob.pack(vt))
    assert w["time_varying"] and np.array_equal(w["Ti"], vt["Ti"]) and "AB" not in w and w["method"] == "FISTA"


def _compare_tv_fista(variant, got, O, vt, x0, xr, ur, model, per):
    """STREAM: the reference's operation order, bit for bit.  MFMA4R (factors in registers, explicit Beta^-1): 1e-10 on u and z (the measured
    |lambda| allowance of tests/_cases.py where the dual blows up), the dual itself relative to its size; k equal unless the oracle flips when
    its tolerance moves by 1e-12."""
    from oracle import oracle
    from _cases import TOL_SPCIES, assert_k, scaled_bar
    u, k, e, sol = got
    if variant != "mfma4r":
        assert np.array_equal(k, O[1]) and np.array_equal(e, O[2]) and np.array_equal(u, O[0])
        if sol.z is not None:
            assert np.array_equal(sol.z, O[3]) and np.array_equal(sol.lam, O[4])
        return

    def rerun(idx, dtol):
        v2 = dict(vt)
        v2["tol"] = float(vt["tol"]) + dtol
        pr = np.ndim(xr) == 2
        return oracle.fista_tv_batch(v2, x0[idx], xr[idx] if pr else xr, ur[idx] if pr else ur, model[idx] if per else model, per, want_sol=False)[1]
    lscale = np.maximum(1.0, np.abs(O[4]).max(axis=1, keepdims=True))
    bar = scaled_bar(lscale)
    same = assert_k(k, O[1], rerun, scale=bar / TOL_SPCIES, what="time-varying FISTA MFMA4R")
    assert np.array_equal(e[same], O[2][same])
    assert (np.abs(u - O[0]) / bar)[same].max() <= 1.0
    if sol.z is not None:
        assert (np.abs(sol.z - O[3]) / bar)[same].max() <= 1.0
        assert (np.abs(sol.lam - O[4]) / (bar * lscale))[same].max() <= 1.0


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["stream", "mfma4r"])  # mfma4r: one wavefront per instance, the factors in registers (admm_tvr_kernel.inc) - AUTO
@pytest.mark.parametrize("name,B,overrides", [("C1_lax_FISTA", 70, {}), ("C1_equ_FISTA", 40, dict(k_max=400)), ("C2_lax_FISTA", 130, {}),
                                              ("C2_equ_FISTA", 50, dict(tol=1e-6, k_max=2000))])
def test_hip_time_varying_fista_vs_oracle(name, B, overrides, variant):
    """The HIP path (update phase + iteration on the instance's own factors) against the oracle: STREAM bit for bit, MFMA4R to 1e-10."""
    from oracle import oracle
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver, SpciesArgError
    cfg, v, vt, design = _setup(name)
    vt = benchmarks.ingredients(cfg, time_varying=True, **overrides)
    s = HipSolver(vt)
    assert s.time_varying and s.variant == "mfma4r" and [f for f, _ in s.sol_fields] == ["z", "lambda"], s.notes
    s.set_variant(variant)
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    with pytest.raises(SpciesArgError):
        s(x0, xr, ur)  # nine inputs are required
    model, per = oracle.pack_tv_model(*design)  # one shared model (the design model)
    O = oracle.fista_tv_batch(vt, x0, xr, ur, model, per)
    _compare_tv_fista(variant, s(x0, xr, ur, *design), O, vt, x0, xr, ur, model, per)
    models = _perturbed_models(design, B)  # one model per instance
    model, per = oracle.pack_tv_model(*models)
    O = oracle.fista_tv_batch(vt, x0, xr, ur, model, per)
    full = s(x0, xr, ur, *models)
    _compare_tv_fista(variant, full, O, vt, x0, xr, ur, model, per)
    nosol = s(x0[:9], xr[:9], ur[:9], *[a[:9] for a in models], want_sol=False)
    assert np.array_equal(nosol[0], full[0][:9]) and np.array_equal(nosol[1], full[1][:9])
    for Bs in (1, 3, 5):  # ragged batches around the four instances of a workgroup, one reference for the whole batch
        Os = oracle.fista_tv_batch(vt, x0[:Bs], xr[0], ur[0], model[:Bs], per)
        _compare_tv_fista(variant, s(x0[:Bs], xr[0], ur[0], *[a[:Bs] for a in models]), Os, vt, x0[:Bs], xr[0], ur[0], model[:Bs], per)
    s.close()


# ----------------------------------------------------------------------------------------------
# option in_engineering (first switch): arguments in engineering units
# ----------------------------------------------------------------------------------------------
def _eng_cfg(name, seed=11):
    import copy
    from types import SimpleNamespace
    from spcies_amd import benchmarks
    cfg = copy.copy(benchmarks.config(name))
    rng = np.random.default_rng(seed)
    n, m = cfg.sys.n, cfg.sys.m
    sysd = dict(vars(cfg.sys))
    sysd.update(Nx=0.5 + rng.random(n), Nu=0.5 + rng.random(m), x0=0.1 * rng.standard_normal(n), u0=0.1 * rng.standard_normal(m))
    cfg.sys = SimpleNamespace(**sysd)
    return cfg


def _oracle_eng(fn, v, x0, xr, ur, **kw):
    """The oracle in scaled units wrapped by the reference's argument scaling (code_laxMPC_ADMM_C.c:83-100, 642-646)."""
    sx, ox, su, ou, siu = (v[k] for k in ("scaling_x", "OpPoint_x", "scaling_u", "OpPoint_u", "scaling_i_u"))
    O = list(fn({k: val for k, val in v.items()}, sx * (x0 - ox), sx * (xr - ox), su * (ur - ou), **kw))
    O[0] = O[0] * siu + ou
    return O


def test_blob_roundtrip_in_engineering():
    from spcies_amd import benchmarks, blob
    cfg = _eng_cfg("C1_lax")
    v = benchmarks.ingredients(cfg, in_engineering=True)
    w = blob.unpack(blob.pack(v))
    assert w["in_engineering"] and np.array_equal(w["scaling_i_u"], 1.0 / cfg.sys.Nu) and np.array_equal(w["OpPoint_x"], cfg.sys.x0)
    assert not blob.unpack(blob.pack(benchmarks.ingredients(cfg)))["in_engineering"]


@pytest.mark.gpu
@pytest.mark.parametrize("name,B", [("C1_soc", 40), ("C1_HMPC_SADMM", 24), ("C1_HMPC_nosplit", 24), ("C1_MPCT_cs", 30), ("C1_ellip", 30)])
def test_hip_in_engineering_sparse_solvers_vs_oracle(name, B):
    """in_engineering of the sparse-KKT / dense-M1 solvers and ellipMPC ADMM (code_ellipMPC_ADMM_soc_C.c:63-72, 292-297;
    code_HMPC_ADMM_split_C.c:78-86, 356-360; code_HMPC_ADMM_C.c:64-72, 265-269; code_MPCT_ADMM_cs_C.c:56-64, 226-230): the STREAM
    variant (reference operation order) equals the oracle wrapped in the same scaling bit for bit, the default variant to 1e-10."""
    from oracle import oracle
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver
    cfg = _eng_cfg(name)
    v = benchmarks.ingredients(cfg, in_engineering=True)
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    x0, xr, ur = x0 / v["scaling_x"] + v["OpPoint_x"], xr / v["scaling_x"] + v["OpPoint_x"], ur / v["scaling_u"] + v["OpPoint_u"]
    extra, kw = (), {}
    if name == "C1_soc":
        extra, kw, fn = (cfg.param.r,), dict(r=cfg.param.r), oracle.admm_soc_batch
    elif name == "C1_HMPC_SADMM":
        fn = oracle.admm_hmpc_batch
    elif name == "C1_HMPC_nosplit":
        fn = oracle.hmpc_dense_batch
    elif name == "C1_MPCT_cs":
        fn = oracle.mpct_cs_batch
    else:
        fn = oracle.admm_banded_batch
    if kw:
        O = _oracle_eng(lambda vv, a, b, c: fn(vv, a, b, c, kw["r"]), v, x0, xr, ur)
    else:
        O = _oracle_eng(fn, v, x0, xr, ur)
    with HipSolver(v) as s:
        u, k, e, sol = s(x0, xr, ur, *extra)  # the default variant
        same = k == O[1]
        assert same.mean() >= 0.9 and np.array_equal(e[same], O[2][same])
        assert np.abs(u - O[0])[same].max() <= 1e-9 and np.abs(sol.z - O[3])[same].max() <= 1e-9
        s.set_variant("stream")
        u, k, e, sol = s(x0, xr, ur, *extra)
        assert np.array_equal(k, O[1]) and np.array_equal(e, O[2]) and np.array_equal(u, O[0]) and np.array_equal(sol.z, O[3])


@pytest.mark.gpu
@pytest.mark.parametrize("name,variant", [("C1_lax", "stream"), ("C1_lax", "mfma4r"), ("C1_equ", "stream"), ("C1_lax_FISTA", "stream"), ("C1_lax_FISTA", "mfma4r")])
def test_hip_time_varying_in_engineering_vs_oracle(name, variant):
    """Both options at once (code_laxMPC_ADMM_C.c:83-100: with TIME_VARYING == 1 the model's LB / UB arrive in engineering units too and are
    scaled like x0 / xr / ur; A, B, Q, R pass as they are): the oracle in scaled units wrapped by that scaling, one model per instance.
    (Refused with ENOSUP before round 5.)"""
    from oracle import oracle
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver
    cfg = _eng_cfg(name)
    vt = benchmarks.ingredients(cfg, time_varying=True, in_engineering=True)
    assert vt["time_varying"] and vt["in_engineering"]
    n, m = cfg.sys.n, cfg.sys.m
    sysm, prm = cfg.sys, cfg.param
    LB = np.concatenate([np.ravel(sysm.LBx), np.ravel(sysm.LBu)])
    UB = np.concatenate([np.ravel(sysm.UBx), np.ravel(sysm.UBu)])
    design = (np.asarray(sysm.A, float), np.asarray(sysm.B, float), np.diag(prm.Q).copy(), np.diag(prm.R).copy(), LB, UB)
    B = 45
    A, Bm, Q, R, LBs, UBs = _perturbed_models(design, B)
    sc = np.concatenate([vt["scaling_x"], vt["scaling_u"]])
    op = np.concatenate([vt["OpPoint_x"], vt["OpPoint_u"]])
    LB_in, UB_in = LBs / sc + op, UBs / sc + op          # what the caller hands over: engineering units
    model, per = oracle.pack_tv_model(A, Bm, Q, R, sc * (LB_in - op), sc * (UB_in - op))  # what the solver iterates on
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    x0, xr, ur = x0 / vt["scaling_x"] + vt["OpPoint_x"], xr / vt["scaling_x"] + vt["OpPoint_x"], ur / vt["scaling_u"] + vt["OpPoint_u"]
    fista = cfg.method == "FISTA"
    fn = oracle.fista_tv_batch if fista else oracle.admm_tv_batch
    O = _oracle_eng(fn, vt, x0, xr, ur, model=model, per_instance=per)
    with HipSolver(vt) as s:
        s.set_variant(variant)
        u, k, e, sol = s(x0, xr, ur, A, Bm, Q, R, LB_in, UB_in)
        if variant == "stream":
            assert np.array_equal(k, O[1]) and np.array_equal(e, O[2]) and np.array_equal(u, O[0])
            assert np.array_equal(sol.z, O[3])  # the record stays in scaled units, as the reference's
        else:
            assert np.array_equal(e, O[2]) and np.abs(k.astype(int) - O[1]).max() <= 1
            assert np.abs(u - O[0]).max() <= 1e-10 and np.abs(sol.z - O[3]).max() <= 1e-10


@pytest.mark.gpu
@pytest.mark.parametrize("name,variant,B", [("C1_lax", "stream", 60), ("C2_lax", "mfma4", 100), ("C1_equ_FISTA", "stream", 40),
                                            ("C1_MPCT", "stream", 30)])
def test_hip_in_engineering_vs_oracle(name, variant, B):
    from oracle import oracle
    from spcies_amd import benchmarks
    from spcies_amd.solver import HipSolver
    cfg = _eng_cfg(name)
    v = benchmarks.ingredients(cfg, in_engineering=True)
    s = HipSolver(v)
    s.set_variant(variant)
    x0, xr, ur = benchmarks.sample_batch(cfg, B)
    x0, xr, ur = x0 / v["scaling_x"] + v["OpPoint_x"], xr / v["scaling_x"] + v["OpPoint_x"], ur / v["scaling_u"] + v["OpPoint_u"]
    fn = {"ADMM": oracle.admm_banded_batch, "FISTA": oracle.fista_banded_batch, "EADMM": oracle.eadmm_mpct_batch}[cfg.method]
    O = _oracle_eng(fn, v, x0, xr, ur)
    u, k, e, sol = s(x0, xr, ur)
    if variant == "stream":
        assert np.array_equal(k, O[1]) and np.array_equal(e, O[2]) and np.array_equal(u, O[0])
        z = sol.z1 if cfg.method == "EADMM" else sol.z
        assert np.array_equal(z, O[3])  # the record stays in scaled units, as the reference's
    else:
        assert np.array_equal(e, O[2]) and np.abs(k.astype(int) - O[1]).max() <= 1
        assert np.abs(u - O[0]).max() <= 1e-10 and np.abs(sol.z - O[3]).max() <= 1e-10
    s.close()
