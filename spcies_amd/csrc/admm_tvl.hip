// The LDS form of the time-varying MFMA4R path (admm_tvl_kernel.inc): its source text for hiprtc and its LDS sizing, in a translation unit of
// their own - the kernels are always run-time specialised, nothing of them is instantiated here (admm_tvr.hip, which instantiates the
// register-resident kernels and takes minutes to compile, does not see this text).
#include "admm_tvr.hpp"
#include "tv_update_kernel.inc"
#include "admm_tvr_kernel.inc"
#include "admm_tvl_kernel.inc"

namespace spcies {
namespace tvr {

// the host's grid arithmetic (admm_tvr.hpp) and the kernel's lane layout must agree for every plant size
template <int n>
constexpr bool coop_layouts_agree() {
    if constexpr (n == 0) return true;
    else return 64 / tvl_coop_lpi(n) == coop_instances_per_wavefront(n) && coop_layouts_agree<n - 1>();
}
static_assert(coop_layouts_agree<32>(), "coop_instances_per_wavefront (admm_tvr.hpp) != 64 / tvl_coop_lpi (admm_tvl_kernel.inc)");

const char *tvl_source() {
    static const char *const text =
#include "admm_tvl_src.inc"
        ;
    return text;
}

// the cooperative update phase in its instance-major form (FORM 1) for the (n, m) of the build-time register-resident kernels
template <int n, int m>
static void coop_go(bool terminal, bool fista, unsigned grid, hipStream_t st, int N, double c0, const double *Tc, const double *model, long model_stride, long B,
                    long Bp, double *TVS) {
#define SPCIES_GO(TT, FF) hipLaunchKernelGGL((tv_update_coop_kernel<n, m, TT, FF, 1>), dim3(grid), dim3(64), 0, st, N, c0, Tc, model, model_stride, B, Bp, TVS)
    if (terminal) { if (fista) SPCIES_GO(true, true); else SPCIES_GO(true, false); }
    else { if (fista) SPCIES_GO(false, true); else SPCIES_GO(false, false); }
#undef SPCIES_GO
}
int launch_coop_builtin(int n, int m, int N, bool terminal, bool fista, double c0, const double *Tc, const double *model, long model_stride, long B, long Bp,
                        double *TVS, hipStream_t st) {
    const int g = coop_instances_per_wavefront(n);
    const unsigned grid = (unsigned)((B + g - 1) / g);
    if (n == 6 && m == 2) coop_go<6, 2>(terminal, fista, grid, st, N, c0, Tc, model, model_stride, B, Bp, TVS);
    else if (n == 12 && m == 2) coop_go<12, 2>(terminal, fista, grid, st, N, c0, Tc, model, model_stride, B, Bp, TVS);
    else return fail(SPCIES_HIP_ENOSUP, "cooperative update phase: no build-time kernel for n=%d m=%d", n, m);
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

long tvl_lds_bytes(int n, int m, int N, bool terminal, bool fista) {
    return 8L * (fista ? ftvl_image_doubles(n, m, N) : tvl_image_doubles(n, m, N, terminal));
}

}  // namespace tvr
}  // namespace spcies
