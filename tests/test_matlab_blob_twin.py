"""The MATLAB side of the boundary cannot run here (no MATLAB / Octave).  What can be pinned on CPU:

* ``matlab/+HIP/write_blob_generic.m`` restated byte for byte in Python (``_write_blob_generic_twin``: the same transposes,
  permutes, alignment, header and directory order) produces blobs that ``spcies_amd.blob.unpack`` - the mirror of the engine's
  parser - reads back with the same header and arrays as ``blob.pack`` of the same ingredients;
* the ``cons_*_HIP.m`` constructors list their arrays as ``{id, vars.<name>, is_int}``: every id they use exists in
  ``include/spcies_hip.h`` and carries the reference variable the engine expects under that id (the rename table below is the
  toolbox's own naming, ``compute_*_ingredients.m``), every array the Python packer ships for that solver is shipped by the .m file
  too, and the header flags the .m files set are the ones ``blob.pack`` sets.
"""
import os
import re
import struct

import numpy as np
import pytest

from spcies_amd import benchmarks, blob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MDIR = os.path.join(ROOT, "matlab")


def _write_blob_generic_twin(hdr, arrays):
    """matlab/+HIP/write_blob_generic.m, line by line.  ``arrays``: list of (id, numpy array as MATLAB holds it, is_int)."""
    align = lambda x: -(-x // 64) * 64
    payload, dims = [], []
    for _id, a, is_int in arrays:
        a = np.asarray(a)
        d = [0, 0, 0, 0]
        if a.ndim == 3:  # MATLAB [i][j][k] -> the engine's [k][i][j]: permute [2 1 3], then column-major a(:)
            d[:3] = [a.shape[2], a.shape[0], a.shape[1]]
            flat = np.transpose(a, (1, 0, 2)).ravel(order="F")
        elif a.ndim <= 1 or 1 in a.shape[:2] and a.ndim == 2 and min(a.shape) == 1:  # isvector
            d[0] = a.size
            flat = a.ravel(order="F")
        else:  # 2-D: a.' then a(:)
            d[:2] = list(a.shape)
            flat = a.T.ravel(order="F")
        if not is_int:
            flat = np.clip(flat.astype("<f8"), -1e20, 1e20)
        payload.append(flat.astype("<i4") if is_int else flat.astype("<f8"))
        dims.append(d)
    na = len(arrays)
    off = align(128 + 48 * na)
    offs = []
    for i in range(na):
        offs.append(off)
        off = align(off + (4 if arrays[i][2] else 8) * payload[i].size)
    total = off
    out = bytearray()
    out += b"SPCSBLB1"
    out += struct.pack("<12I", 1, 128, hdr["formulation"], hdr["method"], hdr["submethod"], hdr["flags"], hdr["n"], hdr["m"], hdr["N"],
                       hdr["k_max"], na, 0)
    out += struct.pack("<Q", total)
    out += struct.pack("<8d", hdr["tol"], hdr["rho"], hdr["rho_i"], *hdr["reserved"])
    for i in range(na):
        out += struct.pack("<2I", arrays[i][0], 1 if arrays[i][2] else 0)
        out += struct.pack("<2Q", offs[i], payload[i].size)
        out += struct.pack("<6I", *dims[i], 0, 0)
    for i in range(na):
        out += bytes(offs[i] - len(out))
        out += payload[i].tobytes()
    out += bytes(total - len(out))
    return bytes(out)


# reference variable name (vars.<name> in the toolbox) -> key of the Python ingredients dict, where they differ
_RENAME = {"rho": None, "T": None, "LB_0": "LB0", "UB_0": "UB0", "L_CSC.val": "L_val", "L_CSC.col": "L_col", "L_CSC.row": "L_row",
           "Q_base_inv": "Q_bi", "Q_mult_inv": "Q_mi", "R_base_inv": "R_bi", "R_mult_inv": "R_mi", "AB_base_inv": "AB_bi",
           "AB_mult_inv": "AB_mi"}


def _m_arrays(path):
    """(id, vars-name) pairs of every `{id, vars.<name>...` cell entry of a cons_*_HIP.m file, and the header flags it can set."""
    src = open(path).read()
    src = re.sub(r"%.*", "", src)
    pairs = re.findall(r"(?<![\w.])(\d+),\s*vars\.([A-Za-z_][\w.]*)", src)
    zeros = re.findall(r"(?<![\w.])(\d+),\s*zeros\(n\)", src)  # equMPC: Hi_N and T travel as zeros(n)
    return [(int(i), nm) for i, nm in pairs] + [(int(i), "zeros(n)") for i in zeros], src


def _header_ids():
    text = open(os.path.join(ROOT, "include", "spcies_hip.h")).read()
    return {int(v) for v in re.findall(r"SPCIES_A_[A-Z0-9_]+\s*=\s*(\d+)", text)}


@pytest.mark.parametrize("mfile", sorted(
    os.path.relpath(os.path.join(d, f), MDIR) for d, _, fs in os.walk(os.path.join(MDIR, "formulations")) for f in fs if f.endswith(".m")))
def test_m_constructor_ids_exist_and_name_the_expected_variable(mfile):
    pairs, _ = _m_arrays(os.path.join(MDIR, mfile))
    if not pairs:  # pure delegates (cons_HMPC_SADMM_split_HIP.m)
        return
    ids = _header_ids()
    id_name = {v: k for k, v in blob.ARRAY_ID.items()}
    for i, nm in pairs:
        assert i in ids, f"{mfile}: id {i} is not declared in include/spcies_hip.h"
        if nm == "zeros(n)":
            continue
        want = id_name[i]
        base = nm.split("(")[0]
        ok = base == want or _RENAME.get(base) == want or (base, want) in {
            ("rho", "rho_mat"), ("rho", "rho_v"), ("rho_i", "rho_i_v"), ("T", "Tdiag"), ("T", "T"), ("rho", "rho_cs"), ("rho_i", "rho_i_cs"),
            ("LB", "LB0"), ("UB", "UB0"), ("LB", "LBN"), ("UB", "UBN"), ("Q", "Q"), ("LB0", "LB0"), ("LBs", "LBs"), ("b", "bh"),
            ("C_CSR.val", "C_val"), ("C_CSR.col", "C_col"), ("C_CSR.row", "C_row"), ("Ct_CSR.val", "Ct_val"), ("Ct_CSR.col", "Ct_col"),
            ("Ct_CSR.row", "Ct_row"), ("idx_x0", "idx_x0"), ("T", "Tz"), ("S", "Sz")} or want.lower().startswith(base.split(".")[0].lower()[:3])
        assert ok, f"{mfile}: id {i} is `{want}` in the engine but the constructor ships vars.{nm}"


_CASES = [  # (config, overrides, cons file, extra-flag expectation)
    ("C1", {}, "formulations/+laxMPC/cons_laxMPC_ADMM_HIP.m"),
    ("C1_lax_gen", {}, "formulations/+laxMPC/cons_laxMPC_ADMM_HIP.m"),
    ("C1_equ", {}, "formulations/+equMPC/cons_equMPC_ADMM_HIP.m"),
    ("C1_equ_gen", {}, "formulations/+equMPC/cons_equMPC_ADMM_HIP.m"),
    ("C1_lax_FISTA", {}, "formulations/+laxMPC/cons_laxMPC_FISTA_HIP.m"),
    ("C1_MPCT", {}, "formulations/+MPCT/cons_MPCT_EADMM_HIP.m"),
    ("C1_MPCT_nd", {}, "formulations/+MPCT/cons_MPCT_EADMM_HIP.m"),
    ("C1_ellip", {}, "formulations/+ellipMPC/cons_ellipMPC_ADMM_HIP.m"),
    ("C1_ellip_vec", {}, "formulations/+ellipMPC/cons_ellipMPC_ADMM_HIP.m"),
]


@pytest.mark.parametrize("cfg_name,overrides,mfile", _CASES)
def test_matlab_writer_twin_round_trips_like_the_python_packer(cfg_name, overrides, mfile):
    """Ingredients -> (ids as the .m constructor lists them, values as MATLAB would hold them) -> the writer's twin -> blob.unpack
    == blob.pack -> blob.unpack, array by array and header field by header field."""
    v = benchmarks.ingredients(benchmarks.config(cfg_name), **overrides)
    packed = blob.pack(v)
    ref = blob.unpack(packed)
    pairs, src = _m_arrays(os.path.join(MDIR, mfile))
    id_name = {val: k for k, val in blob.ARRAY_ID.items()}
    arrays, seen = [], set()
    for i, _nm in pairs:
        key = id_name[i]
        if key not in v or i in seen:
            continue  # a switch of the constructor that is off for this configuration (vector rho, VAR_BOUNDS, general Q R ...)
        seen.add(i)
        a = np.asarray(v[key])
        if a.ndim == 3:  # Python holds [k][i][j]; MATLAB holds [i][j][k]
            a = np.transpose(a, (1, 2, 0))
        arrays.append((i, a, key in blob.INT_ARRAYS))
    shipped_by_python = {k for k in blob.ARRAY_ID if k in v}
    missing = shipped_by_python - {id_name[i] for i, _, _ in arrays}
    if v["formulation"] == "equMPC":  # terminal-block placeholders the Python packer adds and the engine never asks an equMPC blob for
        missing -= {"LBN", "UBN", "rho_N", "rho_i_N"}
    assert not missing, f"{mfile} does not ship {sorted(missing)} (the Python packer does)"
    hb = struct.unpack_from(blob._HDR, packed, 0)
    hdr = dict(formulation=hb[3], method=hb[4], submethod=hb[5], flags=hb[6], n=hb[7], m=hb[8], N=hb[9], k_max=hb[10], tol=hb[14],
               rho=hb[15], rho_i=hb[16], reserved=list(hb[17:22]))
    for bit, needle in ((4, "time_varying"), (16, "bitor(hdr.flags, 16)"), (32, "bitor(hdr.flags, 32)")):
        if hdr["flags"] & bit:
            assert needle in src, f"{mfile} never sets header flag {bit}"
    twin = _write_blob_generic_twin(hdr, arrays)
    got = blob.unpack(twin)
    assert struct.unpack_from(blob._HDR, twin, 0)[:11] == hb[:11] and len(twin) % 64 == 0  # (n_arrays may differ: equMPC placeholders)
    for k, a in ref.items():
        if k not in got and v["formulation"] == "equMPC" and k in ("LBN", "UBN", "rho_N", "rho_i_N"):
            continue
        if isinstance(a, np.ndarray):
            assert k in got and got[k].shape == a.shape and np.array_equal(got[k], a), k
        else:
            assert got[k] == a, k


def test_writer_source_matches_the_twin_in_the_places_that_matter():
    """Guards the twin against drifting from the .m file: magic, header field order, directory entry order, alignment."""
    src = open(os.path.join(MDIR, "+HIP", "write_blob_generic.m")).read()
    assert "fwrite(f, 'SPCSBLB1', 'char');" in src
    assert "fwrite(f, [1 128 hdr.formulation hdr.method hdr.submethod hdr.flags hdr.n hdr.m hdr.N hdr.k_max na 0], 'uint32');" in src
    assert "fwrite(f, total, 'uint64');" in src and "fwrite(f, [hdr.tol hdr.rho hdr.rho_i hdr.reserved(:).'], 'double');" in src
    assert "fwrite(f, [arrays{i, 1} arrays{i, 3}], 'uint32');" in src and "fwrite(f, [offs(i) numel(payload{i})], 'uint64');" in src
    assert "fwrite(f, [dims(i, :) 0 0], 'uint32');" in src and "align = @(x) ceil(x/64)*64;" in src
    assert "a = permute(a, [2 1 3]);" in src and "a = a.';" in src and "off = align(128 + 48*na)" in src
