// The build-time instantiations of the unit-box MFMA4 kernel (admm_mfma4u.hpp) in a translation unit of their own: it is compiled with
// -mllvm -amdgpu-mfma-vgpr-form.  The kernel's z' / w^+ accumulator chains are SEEDED by vector instructions and CONSUMED by vector
// instructions; with the accumulators in the AGPR half (LLVM's default for a 512-register kernel) every seed is a v_accvgpr_write and
// every result a v_accvgpr_read - 356 moves per iteration at BASELINE configs[1], none of which overlaps with an MFMA
// (profiles/r03_microbench_issue.txt).  In VGPR form 187 remain (the rotation of w through the AGPR half) and the vector instructions
// fall into 33 runs instead of 85: 6.24 -> 5.82 ms at configs[1] (profiles/r04_C2_mfma4u_*).  The other MFMA kernels of spcies_hip.hip
// keep the default form (their accumulators feed MFMAs).
#define SPCIES_NO_BUILTIN_LAUNCHERS 1
#include "admm_mfma4u.hpp"

namespace spcies {

template <int N, int KX, int KS>
static int launch_u(bool terminal, bool want_sol, dim3 grid, dim3 block, size_t shmem, hipStream_t st, const MfmaArgs &args, const double *table,
                    const double *x0, const double *xr, const double *ur, double *u, int *k, int *e, double *z, double *v, double *lam, double *dump) {
#define SPCIES_LAUNCH(TERM, SOL)                                                                                                           \
    do {                                                                                                                                   \
        auto kern = admm_mfma4u_kernel<N, KX, KS, TERM, SOL>;                                                                              \
        /* per device, so set before every launch (a handle may live on any GPU of the process) */                                         \
        SPCIES_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                 \
        hipLaunchKernelGGL(kern, grid, block, shmem, st, args, table, x0, xr, ur, u, k, e, z, v, lam, dump);                               \
    } while (0)
    if (terminal) {
        if (want_sol) SPCIES_LAUNCH(true, true); else SPCIES_LAUNCH(true, false);
    } else {
        if (want_sol) SPCIES_LAUNCH(false, true); else SPCIES_LAUNCH(false, false);
    }
#undef SPCIES_LAUNCH
    return 0;
}

int mfma4u_launch_builtin(int N, int KX, int KS, bool terminal, bool want_sol, dim3 grid, dim3 block, size_t shmem, hipStream_t st,
                          const MfmaArgs &args, const double *table, const double *x0, const double *xr, const double *ur, double *u, int *k,
                          int *e, double *z, double *v, double *lam, double *dump) {
#define X(NN, KKX, KKS) \
    if (N == NN && KX == KKX && KS == KKS) return launch_u<NN, KKX, KKS>(terminal, want_sol, grid, block, shmem, st, args, table, x0, xr, ur, u, k, e, z, v, lam, dump);
    SPCIES_MFMA4_SHAPES(X)
#undef X
    return fail(SPCIES_HIP_ENOSUP, "MFMA4 (unit-box) kernel not instantiated for N=%d KX=%d KS=%d", N, KX, KS);
}

}  // namespace spcies
