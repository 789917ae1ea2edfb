// The LDS form of the time-varying MFMA4R path (admm_tvl_kernel.inc): its source text for hiprtc and its LDS sizing, in a translation unit of
// their own - the kernels are always run-time specialised, nothing of them is instantiated here (admm_tvr.hip, which instantiates the
// register-resident kernels and takes minutes to compile, does not see this text).
#include "admm_tvr.hpp"
#include "tv_update_kernel.inc"
#include "admm_tvr_kernel.inc"
#include "admm_tvl_kernel.inc"

namespace spcies {
namespace tvr {

const char *tvl_source() {
    static const char *const text =
#include "admm_tvl_src.inc"
        ;
    return text;
}

long tvl_lds_bytes(int n, int m, int N, bool terminal, bool fista) {
    return 8L * (fista ? ftvl_image_doubles(n, m, N) : tvl_image_doubles(n, m, N, terminal));
}

}  // namespace tvr
}  // namespace spcies
