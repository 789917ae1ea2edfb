// Variant MFMA of the banded-Cholesky ADMM solver - placeholder until the kernel lands.
#pragma once
#include "common.hpp"

namespace spcies {

struct MfmaPlan {
    bool ok = false;
    std::string why = "MFMA variant not built yet";
};

inline int mfma_plan_build(MfmaPlan &p, const AdmmHost &) { p.ok = false; return 0; }
inline void mfma_plan_free(MfmaPlan &) {}
inline int launch_mfma(MfmaPlan &, const AdmmHost &, const double *, const double *, const double *, int, long,
                       double *, int *, int *, double *, double *, double *, hipStream_t) {
    return fail(SPCIES_HIP_ENOSUP, "MFMA variant not built yet");
}

}  // namespace spcies
