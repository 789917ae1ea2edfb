"""CPU: the C-ABI library loads and exports every symbol include/spcies_hip.h declares; error paths
that need no GPU behave (no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from spcies_amd import _lib, benchmarks, blob

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "spcies_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spcies_hip_[a-z_]+)\s*\(", text)))


def test_header_symbols_all_exported():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 10
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/spcies_hip.h but not exported"
    assert set(names) == set(_lib.EXPORTS)
    assert lib.spcies_hip_abi_version() == 1


def test_header_struct_sizes_match_blob_module():
    assert blob.HEADER_BYTES == 128 and blob.ENTRY_BYTES == 48


def test_create_fails_loudly_without_gpu_or_on_bad_blob():
    lib = _lib.load()
    h = C.c_void_p()
    n = C.c_int(-1)
    rc = lib.spcies_hip_device_count(C.byref(n))
    have_gpu = (rc == 0 and n.value > 0)
    rc = lib.spcies_hip_create(b"\0" * 256, 256, 0, C.byref(h))
    assert rc == -1 and b"magic" in lib.spcies_hip_last_error()
    good = blob.pack(benchmarks.ingredients(benchmarks.config("C1")))
    rc = lib.spcies_hip_create(good[:-64], len(good) - 64, 0, C.byref(h))
    assert rc == -1
    if not have_gpu:  # no CPU fallback: a well-formed blob still cannot produce a solver
        rc = lib.spcies_hip_create(good, len(good), 0, C.byref(h))
        assert rc == -2 and not h.value
        with pytest.raises(_lib.SpciesHipError):
            from spcies_amd.solver import HipSolver
            HipSolver(good)


def test_mex_gateway_source_is_valid_c(tmp_path):
    """Our MATLAB mex gateway (source-only deliverable) must at least be valid C against the real
    spcies_hip.h; mex.h comes from a test-only stub (tests/mex_stub/mex.h)."""
    import subprocess
    src = open(os.path.join(ROOT, "matlab", "formulations", "+laxMPC", "struct_laxMPC_ADMM_HIP_Matlab.c")).read()
    defs = "\n".join(["#define DEBUG 1", "#define nn_ 12", "#define mm_ 2", "#define nm_ 14", "#define NN_ 15",
                      "#define dim_ 210", '#define BLOB_PATH "laxMPC.spcb"'])
    src = src.replace("$INSERT_DEFINES$", defs).replace("$FORM$", "laxMPC").replace("$INSERT_NAME$", "laxMPC")
    f = tmp_path / "gateway.c"
    f.write_text(src)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "tests", "mex_stub"), str(f)])


def test_generic_mex_gateway_source_is_valid_c(tmp_path):
    """The generic gateway (all solvers; 0, 1 or 6 extra inputs) is valid C against the real spcies_hip.h."""
    import subprocess
    src0 = open(os.path.join(ROOT, "matlab", "+HIP", "struct_generic_HIP_Matlab.c")).read()
    for n_extra, debug in ((0, 1), (1, 0), (6, 1)):
        defs = "\n".join((["#define DEBUG 1"] if debug else []) + ["#define nn_ 12", "#define mm_ 2", f"#define N_EXTRA_ {n_extra}",
                                                                  '#define BLOB_PATH "solver.spcb"'])
        src = src0.replace("$INSERT_DEFINES$", defs).replace("$FORM$", "laxMPC").replace("$INSERT_NAME$", "laxMPC")
        f = tmp_path / f"gateway_{n_extra}.c"
        f.write_text(src)
        subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                               "-I", os.path.join(ROOT, "tests", "mex_stub"), str(f)])
