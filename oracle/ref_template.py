"""ORACLE cross-check (test infrastructure, THIS CONTAINER ONLY - needs /root/reference).

Instantiates the reference's own C solver template in a temporary directory, compiles it with
gcc -O3 (as ``+sp_utils/get_generic_mex_exec.m:27`` does for the mex) into ``oracle/_ref/*.so`` and
calls it through ctypes, so that ``oracle/admm_banded_oracle.c`` can be compared with the
reference's hot loop itself.

What is the reference's and what is ours - stated precisely because it limits the claim:

* the solver text (``formulations/+X/code_X_ADMM_C.c``, ``header_X_ADMM_C.h``,
  ``platforms/+C_code/generic_solver_struct.c``, ``snippets/*.c|h``) is read where it lies under
  ``/root/reference`` and never copied into this repository (only the compiled ``.so`` is kept,
  under the git-ignored ``oracle/_ref/``);
* the ``$INSERT_DEFINES$`` / ``$INSERT_CONSTANTS$`` blocks, which the reference's MATLAB generator
  would print (``classes/Spcies_constructor.m:95-271``, ``platforms/+C_code/dec_var.m``), are printed
  HERE from our numpy ingredients with the reference's format rules (``%1.15f``, ``[k][i][j]``
  3-D order, +-inf -> +-1e20).  MATLAB is not available, so this is a stand-in for generated
  code: the result is NOT claimed as a "reference build" (bench.py's cpu_baseline stays
  ``kind: "port"``) and the oracle's formal pin is the reference tests' golden optimum.  It is
  recorded as additional evidence that the restated loop nests match the reference's.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np

REF = "/root/reference"
_HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(_HERE, "_ref")


def available():
    return os.path.isdir(os.path.join(REF, "formulations"))


def _fmt(x):
    x = float(x)
    if x == np.inf:
        x = 1e20
    if x == -np.inf:
        x = -1e20
    return "%1.15f" % x


def _decl(name, a):
    """``const static double`` declaration as dec_var.m prints it ('array' option: scalars -> [1])."""
    a = np.asarray(a, dtype=float)
    if a.ndim == 0:
        a = a.reshape(1)
    if a.ndim == 1:
        body = "{ " + ", ".join(_fmt(x) for x in a) + " }"
        dims = f"[{a.size}]"
    elif a.ndim == 2:
        body = "{ " + ", ".join("{" + ", ".join(_fmt(x) for x in r) + "}" for r in a) + " }"
        dims = f"[{a.shape[0]}][{a.shape[1]}]"
    else:  # our 3-D arrays are already [k][i][j]
        body = "{ " + ", ".join("{" + ", ".join("{" + ", ".join(_fmt(x) for x in r) + "}" for r in blk) + "}"
                                for blk in a) + " }"
        dims = f"[{a.shape[0]}][{a.shape[1]}][{a.shape[2]}]"
    return f"const static double {name}{dims} = {body};\n"


def _decl_scalar(name, x):
    return f"const static double {name} = {_fmt(x)};\n"


def _decl_int(name, a):
    a = np.asarray(a).ravel()
    return f"const static int {name}[{a.size}] = " + "{ " + ", ".join("%d" % int(x) for x in a) + " };\n"


def _unescape(text):
    """``fprintf(fid, text)`` semantics of Spcies_constructor.m:222."""
    return text.replace("%%", "%").replace("\\\\", "\\")


def _snippets(text, ext):
    for name in set(re.findall(r"spcies_snippet_(\w+)\(\);", text)):
        with open(os.path.join(REF, "snippets", f"{name}.{ext}")) as f:
            text = text.replace(f"spcies_snippet_{name}();", f.read())
    return text


def build_admm(v, name):
    """Instantiate + compile the lax/equ ADMM or FISTA template for ingredients ``v``; returns the .so path."""
    form, method = v["formulation"], v.get("method", "ADMM")
    if v.get("submethod") == "soc":
        return _build_soc(v, name)
    if v.get("submethod") == "split":
        return _build_hmpc(v, name, sparse=v.get("_template_sparse", True))
    if form == "HMPC":
        return _build_hmpc_nosplit(v, name)
    if v.get("submethod") == "cs":
        return _build_mpct_cs(v, name)
    fdir = os.path.join(REF, "formulations", f"+{form}")
    n, m, N = v["n"], v["m"], v["N"]
    defs = ["#define DEBUG 1", "#define MEASURE_TIME 1", "#define in_engineering 0", "#define TIME_VARYING 0",
            "#define IS_DIAG 1", f"#define nn_ {n}", f"#define mm_ {m}", f"#define nm_ {n + m}", f"#define NN_ {N}",
            f"#define k_max {int(v['k_max'])}", f"#define tol {_fmt(v['tol'])}"]
    variables = ""
    if method == "ADMM" and form == "ellipMPC":  # cons_ellipMPC_ADMM_C.m:74-118
        order = [(k, k) for k in ["LBu0", "UBu0", "LBz", "UBz", "Hi", "Hi_0", "Hi_N", "AB", "P", "P_half", "Pinv_half",
                                  "Alpha", "Beta", "Q", "R", "T"]]
        if v.get("rho_is_scalar", True):
            defs += ["#define SCALAR_RHO", f"#define rho {_fmt(v['rho'])}", f"#define rho_i {_fmt(v['rho_i'])}"]
        else:  # :111-117
            order += [("rho", "rho_v"), ("rho_0", "rho_0"), ("rho_N", "rho_N"), ("rho_i", "rho_i_v"), ("rho_i_0", "rho_i_0"),
                      ("rho_i_N", "rho_i_N")]
        variables = _decl("c", np.asarray(v["c"], float)).replace("const static ", "") \
            + f"double r = {_fmt(v['r'])};\n"
    elif method == "ADMM":  # cons_laxMPC_ADMM_C.m:72-130
        order = [(k, k) for k in ["LB", "UB", "Hi", "Hi_0"] + (["Hi_N"] if v["terminal"] else [])
                 + ["Q", "R", "AB", "Alpha", "Beta"] + (["T"] if v["terminal"] else [])]
        if v.get("var_bounds", False):  # :82-90
            defs += ["#define VAR_BOUNDS 1"]
            order = [("LB0", "LB0"), ("UB0", "UB0")] + order + ([("LBN", "LBN"), ("UBN", "UBN")] if v["terminal"] else [])
        if v.get("rho_is_scalar", True):
            defs += ["#define SCALAR_RHO", f"#define rho {_fmt(v['rho'])}", f"#define rho_i {_fmt(v['rho_i'])}"]
        else:  # :123-129
            order += [("rho", "rho_v"), ("rho_0", "rho_0"), ("rho_i", "rho_i_v"), ("rho_i_0", "rho_i_0")] \
                + ([("rho_N", "rho_N"), ("rho_i_N", "rho_i_N")] if v["terminal"] else [])
    elif method == "FISTA":  # cons_laxMPC_FISTA_C.m:94-107 / cons_equMPC_FISTA_C.m
        order = [(k, k) for k in ["LB", "UB", "AB", "Alpha", "Beta", "Q", "R", "QRi"]] \
            + ([("T", "Tdiag"), ("Ti", "Ti")] if v["terminal"] else [])
    else:  # EADMM, cons_MPCT_EADMM_C.m:82-108 (force_diagonal path: H3i; general Q, R: no IS_DIAG define, six dense blocks)
        order = [("rho", "rho_mat"), ("rho_0", "rho_0"), ("rho_s", "rho_s"), ("LB", "LB"), ("UB", "UB"), ("LB_0", "LB0"),
                 ("UB_0", "UB0"), ("LB_s", "LBs"), ("UB_s", "UBs"), ("AB", "AB"), ("T", "T"), ("S", "S"),
                 ("Alpha", "Alpha"), ("Beta", "Beta"), ("H1i", "H1i"), ("W2", "W2")]
        if v.get("is_diag", True):
            order += [("H3i", "H3i")]
        else:
            defs.remove("#define IS_DIAG 1")  # Spcies_options.m:669: the define is only printed with force_diagonal
            order += [(k, k) for k in ("Q_bi", "Q_mi", "R_bi", "R_mi", "AB_bi", "AB_mi")]
    consts = "".join(_decl(cn, v[k]) for cn, k in order)
    with open(os.path.join(REF, "platforms", "+C_code", "generic_solver_struct.c")) as f:
        code = f.read()
    with open(os.path.join(fdir, f"code_{form}_{method}_C.c")) as f:
        code = code.replace("$INSERT_SOLVER$", f.read())
    with open(os.path.join(fdir, f"header_{form}_{method}_C.h")) as f:
        header = f.read()
    code = code.replace("$INSERT_CONSTANTS$", consts).replace("$INSERT_VARIABLES$", variables)
    header = header.replace("$INSERT_DEFINES$", "\n".join(defs))
    code, header = _snippets(code, "c"), _snippets(header, "h")
    code = _unescape(code.replace("$INSERT_NAME$", name))
    header = _unescape(header.replace("$INSERT_NAME$", name))
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, f"lib{name}.so")
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, f"{name}.c"), "w") as f:
            f.write(code)
        with open(os.path.join(td, f"{name}.h"), "w") as f:
            f.write(header)
        subprocess.check_call(["gcc", "-O3", "-fPIC", "-shared", "-w", "-o", so, os.path.join(td, f"{name}.c"), "-lm"])
    return so


def run_admm(so, v, x0, xr, ur):
    """Call ``<formulation>_<method>`` of the compiled template once per instance.  Returns
    ``u, k, e, z, v, lam`` (``v`` is ``None`` for FISTA, whose record holds ``z`` and ``lambda`` only)."""
    n, m, N = v["n"], v["m"], v["N"]
    method = v.get("method", "ADMM")
    if v.get("submethod") == "soc":
        raise ValueError("use run_soc for the ellipMPC soc solver (extra input r)")
    dim = N * (n + m) - (0 if v["terminal"] else n)
    cs = v.get("submethod") == "cs"
    if cs:
        dim = int(v["dim"])
    lib = C.CDLL(so)
    fn = getattr(lib, f"{v['formulation']}_{method}" + ("_cs" if cs else ""))
    if method == "EADMM":
        return _run_eadmm(lib, fn, v, x0, xr, ur)
    fista = method == "FISTA"
    # header_equMPC_FISTA_C.h declares z[(NN_-1)*nm_+mm_]; the lax one z[NN_*nm_]; lambda[NN_*nn_] in both
    ldim = N * n if fista else dim

    class Sol(C.Structure):
        _fields_ = [("z", C.c_double * dim)] + ([] if fista else [("v", C.c_double * dim)]) + \
                   [("lam", C.c_double * ldim), ("t", C.c_double * 4)]
    x0 = np.atleast_2d(np.asarray(x0, float))
    B = x0.shape[0]
    per = np.ndim(xr) == 2
    u = np.zeros((B, m)); k = np.zeros(B, np.int32); e = np.zeros(B, np.int32)
    z = np.zeros((B, dim)); vv = None if fista else np.zeros((B, dim)); lam = np.zeros((B, ldim))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for i in range(B):
        sol = Sol()
        xi = np.ascontiguousarray(x0[i]); xri = np.ascontiguousarray(xr[i] if per else xr)
        uri = np.ascontiguousarray(ur[i] if per else ur)
        ui = np.zeros(m); ki = C.c_int(0); ei = C.c_int(0)
        fn(dp(xi), dp(xri), dp(uri), dp(ui), C.byref(ki), C.byref(ei), C.byref(sol))
        u[i] = ui; k[i] = ki.value; e[i] = ei.value
        z[i] = np.frombuffer(sol.z); lam[i] = np.frombuffer(sol.lam)
        if not fista:
            vv[i] = np.frombuffer(sol.v)
    return u, k, e, z, vv, lam


def _run_eadmm(lib, fn, v, x0, xr, ur):
    """MPCT_EADMM: record z1, z2, z3, lambda (header_MPCT_EADMM_C.h:14-24).  Returns u, k, e, z1, z2, z3, lam."""
    n, m, N = v["n"], v["m"], v["N"]
    nm = n + m

    class Sol(C.Structure):
        _fields_ = [("z1", C.c_double * ((N + 1) * nm)), ("z2", C.c_double * nm), ("z3", C.c_double * ((N + 1) * nm)),
                    ("lam", C.c_double * ((N + 3) * nm)), ("t", C.c_double * 4)]
    x0 = np.atleast_2d(np.asarray(x0, float))
    B = x0.shape[0]
    per = np.ndim(xr) == 2
    u = np.zeros((B, m)); k = np.zeros(B, np.int32); e = np.zeros(B, np.int32)
    z1 = np.zeros((B, (N + 1) * nm)); z3 = np.zeros((B, (N + 1) * nm)); z2 = np.zeros((B, nm)); lam = np.zeros((B, (N + 3) * nm))
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for i in range(B):
        sol = Sol()
        xi = np.ascontiguousarray(x0[i]); xri = np.ascontiguousarray(xr[i] if per else xr)
        uri = np.ascontiguousarray(ur[i] if per else ur)
        ui = np.zeros(m); ki = C.c_int(0); ei = C.c_int(0)
        fn(dp(xi), dp(xri), dp(uri), dp(ui), C.byref(ki), C.byref(ei), C.byref(sol))
        u[i] = ui; k[i] = ki.value; e[i] = ei.value
        z1[i] = np.frombuffer(sol.z1); z2[i] = np.frombuffer(sol.z2); z3[i] = np.frombuffer(sol.z3)
        lam[i] = np.frombuffer(sol.lam)
    return u, k, e, z1, z2, z3, lam


def _build_soc(v, name):
    """ellipMPC ADMM soc: cons_ellipMPC_ADMM_soc_C.m:66-117 (scalars are plain `const static double`: the
    var_options there carry no 'array' flag)."""
    fdir = os.path.join(REF, "formulations", "+ellipMPC")
    n, m, N = v["n"], v["m"], v["N"]
    defs = ["#define DEBUG 1", "#define MEASURE_TIME 1", "#define in_engineering 0", "#define TIME_VARYING 0",
            "#define IS_DIAG 1", f"#define dim {v['dim']}", f"#define n_s {v['n_s']}", f"#define n_eq {v['n_eq']}",
            f"#define nn_ {n}", f"#define mm_ {m}", f"#define nm_ {n + m}", f"#define NN_ {N}",
            f"#define nrow_GhHhi {len(v['GhHhi_row']) - 1}", f"#define nrow_HhiGh {len(v['HhiGh_row']) - 1}",
            f"#define nrow_Hhi {len(v['Hhi_row']) - 1}", f"#define k_max {int(v['k_max'])}",
            f"#define tol_p {_fmt(v['tol_p'])}", f"#define tol_d {_fmt(v['tol_d'])}"]
    consts = "".join(_decl_scalar(k, v[k]) for k in ("rho", "rho_i", "sigma", "sigma_i"))
    consts += "".join(_decl(k, v[k]) for k in ("Q", "R", "T", "A", "LB", "UB", "PhiP"))
    consts += _decl("L_val", v["L_val"]) + _decl_int("L_col", v["L_col"]) + _decl_int("L_row", v["L_row"]) + _decl("Dinv", v["Dinv"])
    for pfx in ("GhHhi", "HhiGh", "Hhi"):
        consts += _decl(pfx + "_val", v[pfx + "_val"]) + _decl_int(pfx + "_col", v[pfx + "_col"]) + _decl_int(pfx + "_row", v[pfx + "_row"])
    with open(os.path.join(REF, "platforms", "+C_code", "generic_solver_struct.c")) as f:
        code = f.read()
    with open(os.path.join(fdir, "code_ellipMPC_ADMM_soc_C.c")) as f:
        code = code.replace("$INSERT_SOLVER$", f.read())
    with open(os.path.join(fdir, "header_ellipMPC_ADMM_soc_C.h")) as f:
        header = f.read()
    code = code.replace("$INSERT_CONSTANTS$", consts).replace("$INSERT_VARIABLES$", "")  # r is an input of this solver
    header = header.replace("$INSERT_DEFINES$", "\n".join(defs))
    code, header = _snippets(code, "c"), _snippets(header, "h")
    code = _unescape(code.replace("$INSERT_NAME$", name))
    header = _unescape(header.replace("$INSERT_NAME$", name))
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, f"lib{name}.so")
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, f"{name}.c"), "w") as f:
            f.write(code)
        with open(os.path.join(td, f"{name}.h"), "w") as f:
            f.write(header)
        subprocess.check_call(["gcc", "-O3", "-fPIC", "-shared", "-w", "-o", so, os.path.join(td, f"{name}.c"), "-lm"])
    return so


def run_soc(so, v, x0, xr, ur, r):
    """``ellipMPC_ADMM_soc(x0, xr, ur, &r, u, &k, &e, &sol)``; returns u, k, e, z, s, z_hat, s_hat, lam, mu."""
    n, m, dim, n_s = v["n"], v["m"], v["dim"], v["n_s"]
    lib = C.CDLL(so)
    fn = lib.ellipMPC_ADMM_soc

    class Sol(C.Structure):
        _fields_ = [("z", C.c_double * dim), ("s", C.c_double * n_s), ("z_hat", C.c_double * dim),
                    ("s_hat", C.c_double * n_s), ("lam", C.c_double * dim), ("mu", C.c_double * n_s), ("t", C.c_double * 4)]
    x0 = np.atleast_2d(np.asarray(x0, float))
    B = x0.shape[0]
    per = np.ndim(xr) == 2
    r = np.atleast_1d(np.asarray(r, float))
    u = np.zeros((B, m)); k = np.zeros(B, np.int32); e = np.zeros(B, np.int32)
    out = {f: np.zeros((B, w)) for f, w in (("z", dim), ("s", n_s), ("z_hat", dim), ("s_hat", n_s), ("lam", dim), ("mu", n_s))}
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for i in range(B):
        sol = Sol()
        xi = np.ascontiguousarray(x0[i]); xri = np.ascontiguousarray(xr[i] if per else xr)
        uri = np.ascontiguousarray(ur[i] if per else ur)
        ri = C.c_double(r[i] if r.size == B and B > 1 else r[0])
        ui = np.zeros(m); ki = C.c_int(0); ei = C.c_int(0)
        fn(dp(xi), dp(xri), dp(uri), C.byref(ri), dp(ui), C.byref(ki), C.byref(ei), C.byref(sol))
        u[i] = ui; k[i] = ki.value; e[i] = ei.value
        for f in out:
            out[f][i] = np.frombuffer(getattr(sol, f))
    return (u, k, e, out["z"], out["s"], out["z_hat"], out["s_hat"], out["lam"], out["mu"])


def _build_hmpc(v, name, sparse=True):
    """HMPC ADMM / SADMM split, box constraints, sparse (L D L') or NON_SPARSE (dense M1, M2: the reference's
    default option): cons_HMPC_ADMM_split_C.m:88-181.  (`bh` goes to $INSERT_VARIABLES$ as a non-const array;
    scalars rho.. carry no 'array' flag.)"""
    fdir = os.path.join(REF, "formulations", "+HMPC")
    n, m, N = v["n"], v["m"], v["N"]
    defs = ["#define DEBUG 1", "#define MEASURE_TIME 1", "#define in_engineering 0", "#define TIME_VARYING 0",
            "#define IS_DIAG 1", f"#define nn_ {n}", f"#define mm_ {m}", f"#define nm_ {n + m}", f"#define NN_ {N}",
            f"#define dim {v['dim']}", f"#define n_s {v['n_s']}", f"#define n_eq {v['n_eq']}", f"#define n_soc {v['n_soc']}",
            f"#define k_max {int(v['k_max'])}", f"#define tol_p {_fmt(v['tol_p'])}", f"#define tol_d {_fmt(v['tol_d'])}"]
    dim_M2 = (v["n_eq"] + v["n_s"]) if v["use_soc"] else n
    defs += [f"#define nrow_M {v['nrow_M']}"] if sparse else ["#define NON_SPARSE 1", f"#define dim_M2 {dim_M2}"]
    if v["method"] == "SADMM":
        defs += [f"#define alpha_SADMM {_fmt(v['alpha'])}", "#define IS_SYMMETRIC 1"]
    if v["use_soc"]:
        defs += ["#define USE_SOC 1"]
    if v.get("coupled", False):  # cons_HMPC_ADMM_split_C.m:95-97, 108-110
        defs += [f"#define n_y {int(v['n_y'])}", "#define COUPLED_CONSTRAINTS 1"]
    consts = "".join(_decl_scalar(k, v[k]) for k in ("rho", "rho_i", "sigma", "sigma_i"))
    consts += "".join(_decl(cn, v[k]) for cn, k in (("A", "A"), ("QQ", "Q"), ("Te", "Te"), ("Se", "Se"), ("LB", "LB"),
                                                    ("UB", "UB"), ("LBy", "LBy"), ("UBy", "UBy")))
    if sparse:
        consts += _decl("L_val", v["L_val"]) + _decl_int("L_col", v["L_col"]) + _decl_int("L_row", v["L_row"]) \
            + _decl("Dinv", v["Dinv"]) + _decl_int("idx_x0", v["idx_x0"])
        variables = _decl("bh", v["bh"]).replace("const static ", "")
    else:
        consts += _decl("M1", v["M1"]) + _decl("M2", np.asarray(v["M2"])[:, :dim_M2])
        variables = _decl("bh", v["bh_nat"]).replace("const static ", "")
    with open(os.path.join(REF, "platforms", "+C_code", "generic_solver_struct.c")) as f:
        code = f.read()
    with open(os.path.join(fdir, "code_HMPC_ADMM_split_C.c")) as f:
        code = code.replace("$INSERT_SOLVER$", f.read())
    with open(os.path.join(fdir, "header_HMPC_ADMM_split_C.h")) as f:
        header = f.read()
    code = code.replace("$INSERT_CONSTANTS$", consts).replace("$INSERT_VARIABLES$", variables).replace("$INSERT_VARIABLES$", variables)
    header = header.replace("$INSERT_DEFINES$", "\n".join(defs))
    code, header = _snippets(code, "c"), _snippets(header, "h")
    code = _unescape(code.replace("$INSERT_NAME$", name))
    header = _unescape(header.replace("$INSERT_NAME$", name))
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, f"lib{name}.so")
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, f"{name}.c"), "w") as f:
            f.write(code)
        with open(os.path.join(td, f"{name}.h"), "w") as f:
            f.write(header)
        subprocess.check_call(["gcc", "-O3", "-fPIC", "-shared", "-w", "-o", so, os.path.join(td, f"{name}.c"), "-lm"])
    return so


def run_hmpc(so, v, x0, xr, ur):
    """``HMPC_ADMM(x0, xr, ur, u, &k, &e, &sol)``; returns u, k, e, z, s, z_hat, s_hat, lam, mu."""
    n, m, dim, n_s = v["n"], v["m"], v["dim"], v["n_s"]
    lib = C.CDLL(so)
    fn = lib.HMPC_ADMM

    class Sol(C.Structure):
        _fields_ = [("z", C.c_double * dim), ("s", C.c_double * n_s), ("z_hat", C.c_double * dim),
                    ("s_hat", C.c_double * n_s), ("lam", C.c_double * dim), ("mu", C.c_double * n_s), ("t", C.c_double * 4)]
    x0 = np.atleast_2d(np.asarray(x0, float))
    B = x0.shape[0]
    per = np.ndim(xr) == 2
    u = np.zeros((B, m)); k = np.zeros(B, np.int32); e = np.zeros(B, np.int32)
    out = {f: np.zeros((B, w)) for f, w in (("z", dim), ("s", n_s), ("z_hat", dim), ("s_hat", n_s), ("lam", dim), ("mu", n_s))}
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for i in range(B):
        sol = Sol()
        xi = np.ascontiguousarray(x0[i]); xri = np.ascontiguousarray(xr[i] if per else xr)
        uri = np.ascontiguousarray(ur[i] if per else ur)
        ui = np.zeros(m); ki = C.c_int(0); ei = C.c_int(0)
        fn(dp(xi), dp(xri), dp(uri), dp(ui), C.byref(ki), C.byref(ei), C.byref(sol))
        u[i] = ui; k[i] = ki.value; e[i] = ei.value
        for f in out:
            out[f][i] = np.frombuffer(getattr(sol, f))
    return (u, k, e, out["z"], out["s"], out["z_hat"], out["s_hat"], out["lam"], out["mu"])


def _build_hmpc_nosplit(v, name):
    """HMPC ADMM / SADMM without the splitting (the reference's default HMPC solver): cons_HMPC_ADMM_C.m:88-151."""
    fdir = os.path.join(REF, "formulations", "+HMPC")
    n, m, N = v["n"], v["m"], v["N"]
    defs = ["#define DEBUG 1", "#define MEASURE_TIME 1", "#define in_engineering 0", "#define TIME_VARYING 0",
            "#define IS_DIAG 1", f"#define nn_ {n}", f"#define mm_ {m}", f"#define nm_ {n + m}", f"#define NN_ {N}",
            f"#define dim {v['dim']}", f"#define n_s {v['n_s']}", f"#define n_eq {v['n_eq']}", f"#define n_soc {v['n_soc']}",
            f"#define n_y {int(v.get('n_y', n + m))}", f"#define n_box {v['n_box']}", f"#define nrow_C {v['n_s']}", f"#define nrow_Ct {v['dim']}",
            f"#define k_max {int(v['k_max'])}", f"#define tol_p {_fmt(v['tol_p'])}", f"#define tol_d {_fmt(v['tol_d'])}"]
    if v["method"] == "SADMM":
        defs += [f"#define alpha_SADMM {_fmt(v['alpha'])}", "#define IS_SYMMETRIC 1"]
    if v["use_soc"]:
        defs += ["#define USE_SOC 1"]
    consts = "".join(_decl_scalar(k, v[k]) for k in ("rho", "rho_i")) + _decl("A", v["A"])
    consts += _decl("C_val", v["C_val"]) + _decl_int("C_col", v["C_col"]) + _decl_int("C_row", v["C_row"])
    consts += _decl("Ct_val", v["Ct_val"]) + _decl_int("Ct_col", v["Ct_col"]) + _decl_int("Ct_row", v["Ct_row"])
    consts += "".join(_decl(cn, v[k]) for cn, k in (("QQ", "Q"), ("Te", "Te"), ("Se", "Se"), ("LB", "LB"), ("UB", "UB"),
                                                    ("LBy", "LBy"), ("UBy", "UBy")))
    if v["use_soc"]:
        consts += _decl("d", v["d"])
    consts += _decl("M1", v["M1"]) + _decl("M2", v["M2"])
    with open(os.path.join(REF, "platforms", "+C_code", "generic_solver_struct.c")) as f:
        code = f.read()
    with open(os.path.join(fdir, "code_HMPC_ADMM_C.c")) as f:
        code = code.replace("$INSERT_SOLVER$", f.read())
    with open(os.path.join(fdir, "header_HMPC_ADMM_C.h")) as f:
        header = f.read()
    code = code.replace("$INSERT_CONSTANTS$", consts).replace("$INSERT_VARIABLES$", "")
    header = header.replace("$INSERT_DEFINES$", "\n".join(defs))
    code, header = _snippets(code, "c"), _snippets(header, "h")
    code = _unescape(code.replace("$INSERT_NAME$", name))
    header = _unescape(header.replace("$INSERT_NAME$", name))
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, f"lib{name}.so")
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, f"{name}.c"), "w") as f:
            f.write(code)
        with open(os.path.join(td, f"{name}.h"), "w") as f:
            f.write(header)
        subprocess.check_call(["gcc", "-O3", "-fPIC", "-shared", "-w", "-o", so, os.path.join(td, f"{name}.c"), "-lm"])
    return so


def run_hmpc_nosplit(so, v, x0, xr, ur):
    """``HMPC_ADMM(x0, xr, ur, u, &k, &e, &sol)`` of the non-split solver; returns u, k, e, z, s, lam."""
    n, m, dim, n_s = v["n"], v["m"], v["dim"], v["n_s"]
    lib = C.CDLL(so)
    fn = lib.HMPC_ADMM

    class Sol(C.Structure):
        _fields_ = [("z", C.c_double * dim), ("s", C.c_double * n_s), ("lam", C.c_double * n_s), ("t", C.c_double * 4)]
    x0 = np.atleast_2d(np.asarray(x0, float))
    B = x0.shape[0]
    per = np.ndim(xr) == 2
    u = np.zeros((B, m)); k = np.zeros(B, np.int32); e = np.zeros(B, np.int32)
    out = {f: np.zeros((B, w)) for f, w in (("z", dim), ("s", n_s), ("lam", n_s))}
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    for i in range(B):
        sol = Sol()
        xi = np.ascontiguousarray(x0[i]); xri = np.ascontiguousarray(xr[i] if per else xr)
        uri = np.ascontiguousarray(ur[i] if per else ur)
        ui = np.zeros(m); ki = C.c_int(0); ei = C.c_int(0)
        fn(dp(xi), dp(xri), dp(uri), dp(ui), C.byref(ki), C.byref(ei), C.byref(sol))
        u[i] = ui; k[i] = ki.value; e[i] = ei.value
        for f in out:
            out[f][i] = np.frombuffer(getattr(sol, f))
    return (u, k, e, out["z"], out["s"], out["lam"])


def _build_mpct_cs(v, name):
    """MPCT ADMM on the extended state space ('cs'): cons_MPCT_ADMM_cs_C.m:66-112."""
    fdir = os.path.join(REF, "formulations", "+MPCT")
    n, m, N = v["n"], v["m"], v["N"]
    defs = ["#define DEBUG 1", "#define MEASURE_TIME 1", "#define in_engineering 0", "#define TIME_VARYING 0",
            "#define IS_DIAG 1", f"#define nn_ {n}", f"#define mm_ {m}", f"#define nm_ {n + m}", f"#define dnm_ {2 * (n + m)}",
            f"#define nrow_AHi {v['nrow_AHi']}", f"#define nrow_HiA {v['dim']}", f"#define NN_ {N}",
            f"#define k_max {int(v['k_max'])}", f"#define tol {_fmt(v['tol'])}"]
    if v["rho_is_scalar"]:
        defs.append("#define SCALAR_RHO 1")
        consts = _decl_scalar("rho", v["rho"]) + _decl_scalar("rho_i", v["rho_i"])
    else:
        consts = _decl("rho", v["rho_cs"]) + _decl("rho_i", v["rho_i_cs"])
    consts += "".join(_decl(k, v[k]) for k in ("Tz", "Sz", "LB", "UB", "L_val"))
    consts += _decl_int("L_col", v["L_col"]) + _decl_int("L_row", v["L_row"]) + _decl("Dinv", v["Dinv"])
    for pfx in ("AHi", "HiA", "Hi"):
        consts += _decl(f"{pfx}_val", v[f"{pfx}_val"]) + _decl_int(f"{pfx}_col", v[f"{pfx}_col"]) + _decl_int(f"{pfx}_row", v[f"{pfx}_row"])
    with open(os.path.join(REF, "platforms", "+C_code", "generic_solver_struct.c")) as f:
        code = f.read()
    with open(os.path.join(fdir, "code_MPCT_ADMM_cs_C.c")) as f:
        code = code.replace("$INSERT_SOLVER$", f.read())
    with open(os.path.join(fdir, "header_MPCT_ADMM_cs_C.h")) as f:
        header = f.read()
    code = code.replace("$INSERT_CONSTANTS$", consts).replace("$INSERT_VARIABLES$", "")
    header = header.replace("$INSERT_DEFINES$", "\n".join(defs))
    code, header = _snippets(code, "c"), _snippets(header, "h")
    code = _unescape(code.replace("$INSERT_NAME$", name))
    header = _unescape(header.replace("$INSERT_NAME$", name))
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, f"lib{name}.so")
    with tempfile.TemporaryDirectory() as td:
        with open(os.path.join(td, f"{name}.c"), "w") as f:
            f.write(code)
        with open(os.path.join(td, f"{name}.h"), "w") as f:
            f.write(header)
        subprocess.check_call(["gcc", "-O3", "-fPIC", "-shared", "-w", "-o", so, os.path.join(td, f"{name}.c"), "-lm"])
    return so
