// Run-time specialisation of the MFMA4 kernel (admm_mfma4.hpp) for a controller whose (N, ceil(n/4), ceil((n+m)/4))
// is not among the shapes instantiated at build time.  Spcies is a code generator - its C platform prints one
// solver per controller - and the register-resident MFMA4 kernel is specialised on the horizon in the same way:
// the marked regions of the MFMA4 headers (gen_rtc_src.py -> mfma4_rtc_src.inc) are compiled with hiprtc for the
// shape at hand, loaded as a module and launched like the built-in instantiations.  Measured: about one second per
// controller at create time, 2.0-2.3x faster than MFMA4G on the shapes of tools/bench_rtc.py.  SPCIES_HIP_RTC=0 in
// the environment turns it off; libhiprtc.so is bound with dlopen on first use.
#pragma once
#include "admm_mfma4.hpp"
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
    if (m.module) hipModuleUnload(m.module);
    m.module = nullptr;
    m.ok = false;
}

// compile admm_mfma4_kernel<N, KX, KS, TERMINAL, false / true> for gfx950
inline int compile_mfma4(Mfma4Module &out, int N, int KX, int KS, bool terminal) {
    Hiprtc &rt = hiprtc();
    int rc = rt.open();
    if (rc) return rc;
    rt.sync_env();
    void *prog = nullptr;
    if (rt.create(&prog, kMfma4Source, "spcies_mfma4_rtc.hip", 0, nullptr, nullptr) != 0)
        return fail(SPCIES_HIP_EHIP, "hiprtcCreateProgram failed");
    char names[2][128];
    for (int s = 0; s < 2; s++) {
        snprintf(names[s], sizeof(names[s]), "spcies::admm_mfma4_kernel<%d, %d, %d, %s, %s>", N, KX, KS, terminal ? "true" : "false",
                 s ? "true" : "false");
        if (rt.add_name(prog, names[s]) != 0) {
            rt.destroy(&prog);
            return fail(SPCIES_HIP_EHIP, "hiprtcAddNameExpression failed");
        }
    }
    std::vector<std::string> extra;  // experiments: SPCIES_MFMA4_RTC_FLAGS holds extra options, blank-separated
    if (const char *ev = getenv("SPCIES_MFMA4_RTC_FLAGS")) {
        std::string tok;
        for (const char *c = ev;; c++) {
            if (*c == ' ' || *c == '\0') {
                if (!tok.empty()) extra.push_back(tok);
                tok.clear();
                if (!*c) break;
            } else {
                tok.push_back(*c);
            }
        }
    }
    std::vector<const char *> opts = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-honor-nans", "-DSPCIES_RTC_STATIC_LDS=1"};
    for (const std::string &e : extra) opts.push_back(e.c_str());
    const int crc = rt.compile(prog, (int)opts.size(), opts.data());
    if (crc != 0) {
        size_t ls = 0;
        rt.log_size(prog, &ls);
        std::string lg(ls + 1, '\0');
        if (ls) rt.log(prog, &lg[0]);
        rt.destroy(&prog);
        return fail(SPCIES_HIP_EHIP, "hiprtcCompileProgram failed: %.400s", lg.c_str());
    }
    size_t cs = 0;
    rt.code_size(prog, &cs);
    std::vector<char> code(cs);
    rt.code(prog, code.data());
    std::string lowered[2];
    for (int s = 0; s < 2; s++) {
        const char *ln = nullptr;
        if (rt.lowered(prog, names[s], &ln) != 0 || !ln) {
            rt.destroy(&prog);
            return fail(SPCIES_HIP_EHIP, "hiprtcGetLoweredName failed");
        }
        lowered[s] = ln;
    }
    rt.destroy(&prog);
    SPCIES_HIP_CHECK(hipModuleLoadData(&out.module, code.data()));
    for (int s = 0; s < 2; s++) SPCIES_HIP_CHECK(hipModuleGetFunction(&out.fn[s], out.module, lowered[s].c_str()));
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
