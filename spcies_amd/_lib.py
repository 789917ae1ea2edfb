"""ctypes binding of ``libspcies_hip.so`` (the C-ABI of ``include/spcies_hip.h``).

There is no CPU fallback: if the library is missing, or no HIP device is usable, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPCIES_HIP_LIB", os.path.join(_HERE, "libspcies_hip.so"))  # env override: diagnostic builds

EXPORTS = (
    "spcies_hip_abi_version", "spcies_hip_last_error", "spcies_hip_device_count", "spcies_hip_create",
    "spcies_hip_destroy", "spcies_hip_get_info", "spcies_hip_set_variant", "spcies_hip_set_exit",
    "spcies_hip_reserve", "spcies_hip_solve_batch", "spcies_hip_solve_batch_device", "spcies_hip_time_device",
    "spcies_hip_get_sol_layout", "spcies_hip_solve_batch_ex", "spcies_hip_solve_batch_device_ex",
    "spcies_hip_closed_loop",
    "spcies_hip_shard_range", "spcies_hip_create_multi", "spcies_hip_multi_destroy", "spcies_hip_multi_count", "spcies_hip_multi_get",
    "spcies_hip_get_extra_width", "spcies_hip_host_alloc", "spcies_hip_host_free", "spcies_hip_rtc_cache_stats", "spcies_hip_rtc_cache_stats_ex", "spcies_hip_rtc_cache_selftest", "spcies_hip_rtc_compile_selftest", "spcies_hip_residual_trace",
    "spcies_hip_k_histogram_device",
    "spcies_hip_get_notes", "spcies_hip_multi_set_variant", "spcies_hip_multi_set_exit", "spcies_hip_multi_solve_batch", "spcies_hip_multi_solve_batch_ex",
)

VARIANT_AUTO, VARIANT_STREAM, VARIANT_MFMA, VARIANT_MFMA4, VARIANT_MFMA4G, VARIANT_TILE, VARIANT_GEMM, VARIANT_BSP, VARIANT_FUSED, VARIANT_MFMA4R = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
VARIANTS = {"auto": VARIANT_AUTO, "stream": VARIANT_STREAM, "mfma": VARIANT_MFMA, "mfma4": VARIANT_MFMA4,
            "mfma4g": VARIANT_MFMA4G, "tile": VARIANT_TILE, "gemm": VARIANT_GEMM, "bsp": VARIANT_BSP, "fused": VARIANT_FUSED, "mfma4r": VARIANT_MFMA4R}


class SpciesHipError(RuntimeError):
    pass


class Info(C.Structure):
    _fields_ = [("formulation", C.c_int), ("method", C.c_int), ("submethod", C.c_int), ("n", C.c_int),
                ("m", C.c_int), ("N", C.c_int), ("dim", C.c_int), ("k_max", C.c_int), ("tol", C.c_double),
                ("rho", C.c_double), ("variant", C.c_int), ("device", C.c_int), ("dim_lambda", C.c_int)]


class Timing(C.Structure):
    _fields_ = [("update_time", C.c_double), ("solve_time", C.c_double), ("polish_time", C.c_double),
                ("run_time", C.c_double)]


_lib = None


def load():
    """Load the shared library (once) and declare the prototypes."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SpciesHipError(
            f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C spcies_amd/csrc`. The HIP platform has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    dp, ip, vp = C.POINTER(C.c_double), C.POINTER(C.c_int), C.c_void_p
    lib.spcies_hip_abi_version.restype = C.c_int
    lib.spcies_hip_last_error.restype = C.c_char_p
    lib.spcies_hip_device_count.argtypes = [ip]
    lib.spcies_hip_create.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.POINTER(vp)]
    lib.spcies_hip_destroy.argtypes = [vp]
    lib.spcies_hip_get_info.argtypes = [vp, C.POINTER(Info)]
    lib.spcies_hip_set_variant.argtypes = [vp, C.c_int]
    lib.spcies_hip_set_exit.argtypes = [vp, C.c_int, C.c_double]
    lib.spcies_hip_reserve.argtypes = [vp, C.c_long]
    lib.spcies_hip_solve_batch.argtypes = [vp, dp, dp, dp, C.c_int, C.c_long, dp, ip, ip, dp, dp, dp,
                                           C.POINTER(Timing)]
    lib.spcies_hip_solve_batch_device.argtypes = [vp, vp, vp, vp, C.c_int, C.c_long, vp, vp, vp, vp, vp, vp, vp]
    lib.spcies_hip_get_sol_layout.argtypes = [vp, ip, ip, C.POINTER(C.c_char_p)]
    lib.spcies_hip_solve_batch_ex.argtypes = [vp, dp, dp, dp, C.c_int, dp, C.c_int, C.c_long, dp, ip, ip,
                                              C.POINTER(dp), C.c_int, C.POINTER(Timing)]
    lib.spcies_hip_solve_batch_device_ex.argtypes = [vp, vp, vp, vp, C.c_int, vp, C.c_int, C.c_long, vp, vp, vp,
                                                     C.POINTER(vp), C.c_int, vp]
    lib.spcies_hip_time_device.argtypes = [vp, vp, vp, vp, C.c_int, C.c_long, vp, vp, vp, vp, C.c_int, dp]
    lib.spcies_hip_get_notes.argtypes = [vp, C.POINTER(C.c_char_p)]
    lp = C.POINTER(C.c_long)
    lib.spcies_hip_get_extra_width.argtypes = [vp, lp]
    lib.spcies_hip_host_alloc.argtypes = [C.c_size_t, C.POINTER(vp)]
    lib.spcies_hip_host_free.argtypes = [vp]
    lib.spcies_hip_rtc_cache_stats.argtypes = [lp, lp]
    lib.spcies_hip_residual_trace.argtypes = [vp, dp, dp, dp, C.c_int, C.c_long, C.c_int, dp, dp, ip]
    lib.spcies_hip_rtc_cache_stats_ex.argtypes = [lp, C.c_int]
    lib.spcies_hip_rtc_cache_selftest.argtypes = [C.c_char_p, C.c_int, C.c_int, ip, C.POINTER(C.c_ulonglong)]
    lib.spcies_hip_rtc_compile_selftest.argtypes = [C.c_char_p, ip, C.POINTER(C.c_ulong)]
    lib.spcies_hip_k_histogram_device.argtypes = [vp, vp, vp, C.c_long, C.c_int, lp, lp, vp]
    lib.spcies_hip_shard_range.argtypes = [C.c_long, C.c_int, C.c_int, lp, lp]
    lib.spcies_hip_create_multi.argtypes = [C.c_char_p, C.c_size_t, ip, C.c_int, C.POINTER(vp)]
    lib.spcies_hip_multi_destroy.argtypes = [vp]
    lib.spcies_hip_multi_count.argtypes = [vp, ip]
    lib.spcies_hip_multi_get.argtypes = [vp, C.c_int, C.POINTER(vp)]
    lib.spcies_hip_multi_set_variant.argtypes = [vp, C.c_int]
    lib.spcies_hip_multi_set_exit.argtypes = [vp, C.c_int, C.c_double]
    lib.spcies_hip_multi_solve_batch.argtypes = [vp, dp, dp, dp, C.c_int, C.c_long, dp, ip, ip, dp, dp, dp, C.POINTER(Timing)]
    lib.spcies_hip_multi_solve_batch_ex.argtypes = [vp, dp, dp, dp, C.c_int, dp, C.c_int, C.c_long, C.c_long, dp, ip, ip,
                                                    C.POINTER(dp), C.c_int, C.POINTER(Timing)]
    for name in EXPORTS:
        getattr(lib, name)  # AttributeError here = header and library out of sync
        if name not in ("spcies_hip_last_error",):
            getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        msg = load().spcies_hip_last_error()
        raise SpciesHipError(f"libspcies_hip error {rc}: {msg.decode() if msg else ''}")
