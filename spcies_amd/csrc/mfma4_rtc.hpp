// Run-time specialisation of the MFMA4 kernel (admm_mfma4.hpp) for a controller whose (N, ceil(n/4), ceil((n+m)/4))
// is not among the shapes instantiated at build time.  Spcies is a code generator - its C platform prints one
// solver per controller - and the register-resident MFMA4 kernel is specialised on the horizon in the same way:
// the marked regions of the MFMA4 headers (gen_rtc_src.py -> mfma4_rtc_src.inc) are compiled with hiprtc for the
// shape at hand, loaded as a module and launched like the built-in instantiations.  Measured: about one second per
// controller at create time, 2.0-2.3x faster than MFMA4G on the shapes of tools/bench_rtc.py.  SPCIES_HIP_RTC=0 in
// the environment turns it off; libhiprtc.so is bound with dlopen on first use.
#pragma once
#include "admm_mfma4u.hpp"
#include "rtc_common.hpp"

namespace spcies {
namespace rtc {

static const char *const kMfma4Source =
#include "mfma4_rtc_src.inc"
    ;

struct Mfma4Module {
    hipModule_t module = nullptr;
    hipFunction_t fn[2] = {nullptr, nullptr};  // WANT_SOL = false, true
    bool ok = false;
};
inline void module_free(Mfma4Module &m) {
    if (m.module) rtc::unload_module(m.module);
    m.module = nullptr;
    m.ok = false;
}

// compile admm_mfma4_kernel<N, KX, KS, TERMINAL, false / true> for gfx950
inline int compile_mfma4(Mfma4Module &out, int N, int KX, int KS, bool terminal, bool unit = false) {
    std::vector<std::string> names;
    for (int s = 0; s < 2; s++) {
        char nm[128];
        snprintf(nm, sizeof(nm), "spcies::admm_mfma4%s_kernel<%d, %d, %d, %s, %s>", unit ? "u" : "", N, KX, KS, terminal ? "true" : "false",
                 s ? "true" : "false");
        names.push_back(nm);
    }
    std::vector<std::string> extra = {"-DSPCIES_RTC_STATIC_LDS=1"};
    if (unit) {  // as the build-time instantiations (admm_mfma4u.hip): accumulators seeded and consumed by vector instructions belong in VGPRs
        extra.push_back("-mllvm");
        extra.push_back("-amdgpu-mfma-vgpr-form");
    }
    // experiments: SPCIES_MFMA4_RTC_FLAGS holds extra options, blank-separated
    for (const std::string &e : split_flags(getenv("SPCIES_MFMA4_RTC_FLAGS"))) extra.push_back(e);
    int rc = compile_module(kMfma4Source, "spcies_mfma4_rtc.hip", names, extra, &out.module, out.fn);  // cached per process
    if (rc) return rc;
    out.ok = true;
    return 0;
}

// launch like launch_mfma4_shape (admm_mfma4.hpp); the table lives in static LDS
inline int launch_mfma4(const Mfma4Module &m, const Mfma4Plan &pl, const AdmmHost &a, const double *x0, const double *xr,
                        const double *ur, int ref_stride, long B, double *u, int *k, int *e, double *z, double *v, double *lam,
                        hipStream_t st) {
    const bool want_sol = (z || v || lam);
    if (want_sol && !(z && v && lam)) return fail(SPCIES_HIP_EINVAL, "MFMA4 variant: pass all of z, v, lambda or none");
    MfmaArgs args{a.n, a.m, a.k_max, a.tol, a.rho, a.rho_i, B, ref_stride};
    const long n_tiles = (B + 15) / 16;
    long wgs = (n_tiles + 3) / 4;
    if (wgs > pl.num_cu) wgs = pl.num_cu;
    const double *table = pl.d_table;
    double *dump = pl.d_table + pl.table_bytes / sizeof(double);
    void *params[] = {&args, &table, &x0, &xr, &ur, &u, &k, &e, &z, &v, &lam, &dump};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel(m.fn[want_sol ? 1 : 0], (unsigned)wgs, 1, 1, 256, 1, 1, 0, st, params, nullptr));
    return 0;
}

}  // namespace rtc
}  // namespace spcies
