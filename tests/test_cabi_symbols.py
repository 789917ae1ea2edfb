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


_FUZZ_CONFIGS = ["C1", "C1_equ_FISTA", "C1_MPCT", "C1_soc", "C1_HMPC", "C1_HMPC_nosplit", "C1_MPCT_cs", "C1_ellip", "C1_lax_gen"]


@pytest.mark.parametrize("cfg_name", _FUZZ_CONFIGS)
def test_blob_parser_rejects_or_accepts_mutations_without_crashing(cfg_name):
    """The blob is the drop-in boundary's only untrusted input: `spcies_hip_create` parses it before it touches a device, so the
    parser can be exercised here.  Every well-formed blob is accepted by the parser (it then fails with ENODEV = -2 on a box
    without a GPU); truncations, a directory entry pointing outside the blob, counts that disagree with n / m / N and index arrays
    with out-of-range entries are refused with EINVAL = -1 - and nothing crashes on 200 seeded single-word mutations."""
    import struct

    import numpy as np
    lib = _lib.load()
    n = C.c_int(-1)
    have_gpu = (lib.spcies_hip_device_count(C.byref(n)) == 0 and n.value > 0)
    if have_gpu:
        pytest.skip("parser-only test: on a GPU box create() succeeds and the solver tests cover it")
    good = blob.pack(benchmarks.ingredients(benchmarks.config(cfg_name)))

    def create(b):
        h = C.c_void_p()
        rc = lib.spcies_hip_create(bytes(b), len(b), 0, C.byref(h))
        assert not h.value
        return rc
    assert create(good) == -2                                  # parsed; no device
    assert create(good[:len(good) // 2]) == -1                 # truncated
    n_arr = struct.unpack_from("<I", good, 8 + 4 * 10)[0]
    assert n_arr >= 5
    # split HMPC: M1 / M2 / bh_nat only enable the GEMM variant; non-split HMPC in diamond mode never reads d
    optional = {"C1_HMPC": {68, 69, 70}, "C1_HMPC_nosplit": {77}}.get(cfg_name, set())
    for i in range(n_arr):                                     # an entry that points past the end / holds too many elements
        aid = struct.unpack_from("<I", good, blob.HEADER_BYTES + i * blob.ENTRY_BYTES)[0]
        for field_off, val in ((8, len(good) + 64), (16, 1 << 40)):
            b = bytearray(good)
            struct.pack_into("<Q", b, blob.HEADER_BYTES + i * blob.ENTRY_BYTES + field_off, val)
            assert create(b) == (-2 if aid in optional else -1), (i, aid, field_off)
    b = bytearray(good)                                        # n disagrees with the arrays
    struct.pack_into("<I", b, 8 + 4 * 6, struct.unpack_from("<I", good, 8 + 4 * 6)[0] + 1)
    assert create(b) == -1
    for i in range(n_arr):                                     # an index array with a negative / huge entry
        aid, dtype, off, count = struct.unpack_from("<IIQQ", good, blob.HEADER_BYTES + i * blob.ENTRY_BYTES)
        if dtype == 1 and count > 2:
            for val in (-1, 1 << 30):
                b = bytearray(good)
                struct.pack_into("<i", b, off + 4 * (count // 2), val)
                assert create(b) == -1, (aid, val)
    rng = np.random.default_rng(5)
    words = len(good) // 4
    for _ in range(200):                                       # random single-word mutations: any verdict, no crash
        b = bytearray(good)
        w = int(rng.integers(0, min(words, (blob.HEADER_BYTES + n_arr * blob.ENTRY_BYTES) // 4 + 64)))
        struct.pack_into("<I", b, 4 * w, int(rng.integers(0, 1 << 32)))
        assert create(b) in (-1, -2, -3, -4)


def test_shard_range_matches_the_python_partition():
    """spcies_hip_shard_range (the split spcies_hip_multi_solve_batch applies to a host batch) == distributed.shard_range:
    contiguous, complete, sizes differ by at most one, larger shards first.  No GPU needed."""
    from spcies_amd import distributed
    lib = _lib.load()
    lo, cnt = C.c_long(), C.c_long()
    for B in (0, 1, 7, 64, 65536, 524288 + 5):
        for G in (1, 2, 3, 8):
            end = 0
            for g in range(G):
                assert lib.spcies_hip_shard_range(B, G, g, C.byref(lo), C.byref(cnt)) == 0
                assert (lo.value, lo.value + cnt.value) == distributed.shard_range(B, G, g) and lo.value == end
                end += cnt.value
            assert end == B
    assert lib.spcies_hip_shard_range(10, 0, 0, C.byref(lo), C.byref(cnt)) != 0
    assert lib.spcies_hip_shard_range(10, 2, 2, C.byref(lo), C.byref(cnt)) != 0
    assert b"shard" in lib.spcies_hip_last_error()


def test_create_multi_fails_loudly_without_gpu_or_on_bad_blob():
    lib = _lib.load()
    h = C.c_void_p()
    ids = (C.c_int * 2)(0, 0)
    assert lib.spcies_hip_create_multi(b"x" * 200, 200, ids, 2, C.byref(h)) != 0 and not h
    assert lib.spcies_hip_multi_destroy(None) == 0
