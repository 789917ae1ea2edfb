// Shared host-side declarations of libspcies_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spcies_hip.h"

namespace spcies {

extern thread_local std::string g_last_error;

inline int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define SPCIES_HIP_CHECK(expr)                                                                          \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return ::spcies::fail(SPCIES_HIP_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                                  __FILE__, __LINE__);                                                  \
    } while (0)

// Reference-style ingredients of a banded-Cholesky ADMM controller (laxMPC / equMPC), host copy.
struct AdmmHost {
    int n = 0, m = 0, N = 0, k_max = 0;
    bool terminal = true;
    double tol = 0, rho = 0, rho_i = 0;
    std::vector<double> AB, Alpha, Beta, Hi, Hi_0, Hi_N, Q, R, T, LB, UB;
    // ellipMPC ADMM (code_ellipMPC_ADMM_C.c): terminal ellipsoid (P, c, r) and stage-wise bounds
    bool ellip = false;
    double r_ell = 0;
    std::vector<double> P, P_half, Pinv_half, c_ell, LBz, UBz, LBu0, UBu0;
    // lax/equ switches: vector rho (no SCALAR_RHO) and VAR_BOUNDS (code_laxMPC_ADMM_C.c:323-348, 490-568).  When
    // either is set (`gen`), the stage-wise form of BOTH is kept: rho_0 [m], rho_v [N-1][n+m], rho_N [n] and their
    // reciprocals, bounds in LBu0/UBu0 [m], LBz/UBz [N-1][n+m], LBN/UBN [n]
    bool gen = false;
    std::vector<double> rho_0, rho_v, rho_N, rho_i_0, rho_i_v, rho_i_N, LBN, UBN;
    int dim() const { return N * (n + m) - (terminal ? 0 : n); }
};

}  // namespace spcies
#include "admm_dev.inc"  // struct AdmmDev: the STREAM kernels' argument (also part of their run-time specialised source)
namespace spcies {

}  // namespace spcies
