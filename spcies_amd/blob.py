"""Problem blob: the controller's constants in the layout ``include/spcies_hip.h`` documents.

The reference prints these arrays as ``const static double`` initialisers into the generated C file
(``formulations/+laxMPC/cons_laxMPC_ADMM_C.m:82-118`` through ``platforms/+C_code/dec_var.m``);
the HIP platform ships them as one little-endian blob instead (full doubles, no ``%1.15f``
quantisation).
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"SPCSBLB1"
VERSION = 1
HEADER_BYTES = 128
ENTRY_BYTES = 48

FORMULATION = {"laxMPC": 1, "equMPC": 2, "MPCT": 3, "ellipMPC": 4, "HMPC": 5}
METHOD = {"ADMM": 1, "FISTA": 2, "EADMM": 3, "SADMM": 4}
ARRAY_ID = {"AB": 1, "Alpha": 2, "Beta": 3, "Hi": 4, "Hi_0": 5, "Hi_N": 6, "Q": 7, "R": 8, "T": 9, "LB": 10, "UB": 11,
            "QRi": 12, "Tdiag": 13, "Ti": 14,
            "S": 15, "rho_mat": 16, "rho_0": 17, "rho_s": 18, "LB0": 19, "UB0": 20, "LBs": 21, "UBs": 22,
            "H1i": 23, "W2": 24, "H3i": 25,
            "A": 26, "PhiP": 27, "L_val": 28, "L_col": 29, "L_row": 30, "Dinv": 31, "GhHhi_val": 32, "GhHhi_col": 33,
            "GhHhi_row": 34, "HhiGh_val": 35, "HhiGh_col": 36, "HhiGh_row": 37, "Hhi_val": 38, "Hhi_col": 39,
            "Hhi_row": 40, "Te": 41, "Se": 42, "LBy": 43, "UBy": 44, "idx_x0": 45, "bh": 46, "T_rho_i": 47,
            "scaling_x": 48, "scaling_u": 49, "scaling_i_u": 50, "OpPoint_x": 51, "OpPoint_u": 52,
            "P": 53, "P_half": 54, "Pinv_half": 55, "c": 56, "LBz": 57, "UBz": 58, "LBu0": 59, "UBu0": 60,
            "rho_v": 61, "rho_N": 62, "rho_i_v": 63, "rho_i_0": 64, "rho_i_N": 65, "LBN": 66, "UBN": 67, "M1": 68, "M2": 69, "bh_nat": 70,
            "C_val": 71, "C_col": 72, "C_row": 73, "Ct_val": 74, "Ct_col": 75, "Ct_row": 76, "d": 77,
            "Tz": 78, "Sz": 79, "AHi_val": 80, "AHi_col": 81, "AHi_row": 82, "HiA_val": 83, "HiA_col": 84, "HiA_row": 85,
            "Hi_val": 86, "Hi_col": 87, "Hi_row": 88, "rho_cs": 89, "rho_i_cs": 90,
            "Q_bi": 91, "Q_mi": 92, "R_bi": 93, "R_mi": 94, "AB_bi": 95, "AB_mi": 96}
INT_ARRAYS = {"L_col", "L_row", "GhHhi_col", "GhHhi_row", "HhiGh_col", "HhiGh_row", "Hhi_col", "Hhi_row", "idx_x0",
              "C_col", "C_row", "Ct_col", "Ct_row",
              "AHi_col", "AHi_row", "HiA_col", "HiA_row", "Hi_col", "Hi_row"}
SUBMETHOD = {"": 0, "soc": 1, "split": 2, "cs": 3, "semiband": 4}
_ID_NAME = {v: k for k, v in ARRAY_ID.items()}
_HDR = "<8sIIIIIIIIIIIIQddd5d"
assert struct.calcsize(_HDR) == HEADER_BYTES
_ENT = "<IIQQ4I2I"
assert struct.calcsize(_ENT) == ENTRY_BYTES

INF_VALUE = 1e20  # +-inf bounds are stored as +-1e20, as dec_var.m:245-248 prints them


def _align(x, a=64):
    return (x + a - 1) // a * a


def pack(v):
    """Pack an ingredients dict (``compute_*_ingredients``) into blob bytes."""
    names = [k for k in ARRAY_ID if k in v and not (k == "d" and v.get("submethod") != "")]  # d: HMPC without the splitting
    arrays = []
    for k in names:
        if k in INT_ARRAYS:
            arrays.append((k, np.ascontiguousarray(np.asarray(v[k], dtype="<i4"))))
            continue
        a = np.ascontiguousarray(np.asarray(v[k], dtype="<f8"))
        if k in ("LB", "UB", "LB0", "UB0", "LBs", "UBs", "LBy", "UBy", "LBz", "UBz", "LBu0", "UBu0", "LBN", "UBN"):
            a = np.clip(a, -INF_VALUE, INF_VALUE)
        arrays.append((k, a))
    off = _align(HEADER_BYTES + ENTRY_BYTES * len(arrays))
    entries, payload = [], []
    for k, a in arrays:
        dims = list(a.shape)[:4] + [0] * (4 - min(a.ndim, 4))
        entries.append(struct.pack(_ENT, ARRAY_ID[k], 1 if k in INT_ARRAYS else 0, off, a.size, *dims, 0, 0))
        payload.append((off, a.tobytes()))
        off = _align(off + a.nbytes)
    total = off
    flags = (1 if v.get("rho_is_scalar", True) else 0) | (2 if v.get("use_soc", False) else 0) | (
        4 if v.get("time_varying", False) else 0) | (8 if v.get("in_engineering", False) else 0) | (16 if v.get("var_bounds", False) else 0) | (
        32 if not v.get("is_diag", True) else 0) | (64 if v.get("coupled", False) and v.get("submethod") == "split" else 0)
    res = [float(v.get("sigma", 0.0)), float(v.get("sigma_i", 0.0)), float(v.get("tol_d", 0.0)), float(v.get("alpha", 0.0)), float(v.get("r", 0.0))]
    hdr = struct.pack(_HDR, MAGIC, VERSION, HEADER_BYTES, FORMULATION[v["formulation"]], METHOD[v["method"]],
                      SUBMETHOD[v.get("submethod", "")], flags, int(v["n"]), int(v["m"]), int(v["N"]), int(v["k_max"]),
                      len(arrays), 0, total, float(v["tol"]), float(v["rho"]), float(v["rho_i"]), *res)
    buf = bytearray(total)
    buf[:HEADER_BYTES] = hdr
    p = HEADER_BYTES
    for e in entries:
        buf[p:p + ENTRY_BYTES] = e
        p += ENTRY_BYTES
    for o, b in payload:
        buf[o:o + len(b)] = b
    return bytes(buf)


def unpack(blob):
    """Inverse of :func:`pack` (used by tests and by the multi-GPU broadcast receiver)."""
    (magic, version, hb, form, meth, sub, flags, n, m, N, k_max, n_arr, _r0, total, tol, rho, rho_i,
     *_res) = struct.unpack_from(_HDR, blob, 0)
    if magic != MAGIC or version != VERSION or hb != HEADER_BYTES or total != len(blob):
        raise ValueError("not a spcies problem blob")
    inv = lambda d, x: next(k for k, val in d.items() if val == x)
    v = dict(formulation=inv(FORMULATION, form), method=inv(METHOD, meth), n=n, m=m, N=N, k_max=k_max, tol=tol,
             rho=rho, rho_i=rho_i, rho_is_scalar=bool(flags & 1), time_varying=bool(flags & 4),
             in_engineering=bool(flags & 8), var_bounds=bool(flags & 16), is_diag=not bool(flags & 32))
    v["terminal"] = v["formulation"] != "equMPC"
    for i in range(n_arr):
        aid, dtype, off, count, d0, d1, d2, d3, _p0, _p1 = struct.unpack_from(_ENT, blob, HEADER_BYTES + i * ENTRY_BYTES)
        shape = tuple(d for d in (d0, d1, d2, d3) if d) or (count,)
        v[_ID_NAME[aid]] = np.frombuffer(blob, dtype="<i4" if dtype == 1 else "<f8", count=count,
                                         offset=off).reshape(shape).copy()
    return v
