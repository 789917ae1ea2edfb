// Host side of the MFMA4R variant of the time-varying lax/equ ADMM solvers (admm_tvr.hpp): the explicit inverses after the update phase,
// then one wavefront per instance.  A translation unit of its own: the horizon is unrolled by #pragma unroll (register arrays), which
// needs clang's size limit lifted, and the kernels are large.
#include "admm_tvr.hpp"

namespace spcies {
namespace tvr {

template <int n, int m, int N>
static int launch_shape(bool terminal, bool want_sol, const Args &a, const double *TRI, const double *T, double *TVS, const double *x0, const double *xr,
                        const double *ur, double *u, int *k, int *e, double *z, double *v, double *lam, int num_cu, hipStream_t st) {
    hipLaunchKernelGGL((admm_tv_bi_kernel<n, m>), dim3((unsigned)(a.Bp / 64)), dim3(64), 0, st, N, a.B, a.Bp, TVS);
    const long groups = (a.B + 3) / 4;
    const unsigned grid = (unsigned)std::min<long>(groups, (long)num_cu);
#define SPCIES_TVR_GO(TT, SS) \
    hipLaunchKernelGGL((admm_tvr_kernel<n, m, N, TT, SS>), dim3(grid), dim3(256), 0, st, a, TRI, T, TVS, x0, xr, ur, u, k, e, z, v, lam)
    if (terminal) {
        if (want_sol) SPCIES_TVR_GO(true, true); else SPCIES_TVR_GO(true, false);
    } else {
        if (want_sol) SPCIES_TVR_GO(false, true); else SPCIES_TVR_GO(false, false);
    }
#undef SPCIES_TVR_GO
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch(int n, int m, int N, bool terminal, bool want_sol, const Args &a, const double *TRI, const double *T, double *TVS, const double *x0,
           const double *xr, const double *ur, double *u, int *k, int *e, double *z, double *v, double *lam, int num_cu, hipStream_t st) {
    if (want_sol && !(z && v && lam)) return fail(SPCIES_HIP_EINVAL, "MFMA4R (time-varying): pass all of z, v, lambda or none");
#define X(nn, mm, NN) \
    if (n == nn && m == mm && N == NN) return launch_shape<nn, mm, NN>(terminal, want_sol, a, TRI, T, TVS, x0, xr, ur, u, k, e, z, v, lam, num_cu, st);
    SPCIES_TVR_SHAPES(X)
#undef X
    return fail(SPCIES_HIP_ENOSUP, "MFMA4R (time-varying): no kernel for n = %d, m = %d, N = %d", n, m, N);
}

}  // namespace tvr
}  // namespace spcies
