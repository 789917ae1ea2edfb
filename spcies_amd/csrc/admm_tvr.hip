// Host side of the MFMA4R variant of the time-varying lax/equ ADMM solvers (admm_tvr.hpp): the explicit inverses after the update phase,
// then one wavefront per instance.  A translation unit of its own: the horizon is unrolled by #pragma unroll (register arrays), which
// needs clang's size limit lifted; accumulators in VGPR form (they are seeded and consumed by vector instructions).
#include "admm_tvr.hpp"
#include "admm_stream.hpp"   // tv_layout: the rows the update phase writes
#include "fista_stream.hpp"  // fista_tv_layout
#include "admm_tvr_kernel.inc"
#include "rtc_common.hpp"

namespace spcies {
namespace tvr {

static const char *const kSourceSolve =
#include "admm_tvr_src.inc"
    ;
static const char *const kSourceUpdate =
#include "tv_update_src.inc"
    ;

void plan_free(Plan &p) {
    if (p.module) rtc::unload_module((hipModule_t)p.module);
    p.module = nullptr;
    p.ok = false;
}

int plan_build(Plan &p, int n, int m, int N, bool terminal, bool fista) {
    p.ok = false;
    p.n = n; p.m = m; p.N = N; p.terminal = terminal; p.fista = fista;
    p.update_builtin = (n == 6 && m == 2) || (n == 12 && m == 2);  // admm_stream.hpp / fista_stream.hpp instantiate the update phase for these
    p.lds = false;
    if (n < 1 || m < 1) { p.why = "MFMA4R (time-varying): n, m >= 1"; return 0; }
    if (N < 2) { p.why = "N < 2"; return 0; }
    {   // do the instance's factors fit the wavefront's registers?  Otherwise: the LDS form (admm_tvl_kernel.inc)
        const int KXr = (n + 3) / 4, NLr = nl_of(N, KXr);
        const int need = N * KXr + 2 * (N - 1 - NLr) * KXr + 16 + (fista ? 3 * N : 2 * N + 1) + 40;
        const char *pv = getenv("SPCIES_HIP_TVL");
        const bool prefer_lds = pv && pv[0] == '2';  // (experiments: the LDS form also where the registers would hold the factors)
        if (n + m > 16 || 2 * need > 500 || prefer_lds) {
            const char *ev = getenv("SPCIES_HIP_TVL");
            if (ev && ev[0] == '0') { p.why = "MFMA4R (time-varying): the instance's factors do not fit the wavefront's registers and SPCIES_HIP_TVL=0 (use STREAM)"; return 0; }
            if (n + m > 32) { p.why = "MFMA4R (time-varying): n + m <= 32 (a stage vector is at most two registers in the D layout; use STREAM)"; return 0; }
            const long bytes = tvl_lds_bytes(n, m, N, terminal, fista);  // (admm_tvl.hip)
            if (bytes > 160 * 1024) { p.why = "MFMA4R (time-varying): the instance's factors do not fit the CU's LDS (use STREAM)"; return 0; }
            if (2 * ((n + m + 15) / 16) * (fista ? 3 * N : 2 * N + 1) > 400) /* w and mu (FISTA: y, lambda, d): 2 N + 1 (3 N) vectors of one or two registers of doubles */ { p.why = "MFMA4R (time-varying, LDS form): the iteration state does not fit the registers (use STREAM)"; return 0; }
            p.lds = true;
            p.lds_per_cu = (int)std::min<long>(8, (160 * 1024) / bytes);
            if (const char *cv = getenv("SPCIES_TVL_PER_CU")) p.lds_per_cu = std::max(1, std::min(atoi(cv), (int)((160 * 1024) / bytes)));
            p.update_builtin = false;  // (everything from one module: update phase - rolled past n = 16 -, inverses, solve)
        }
    }
    p.rows_all = (fista ? (long)frows_of(n, m, N).Bi : (long)rows_of(n, m, N).Bi) + (long)N * n * n;
    {   // the register-resident form takes its images from the cooperative update phase, instance-major (SPCIES_TVR_COOP=0: one lane per instance + tv_ms_kernel)
        const char *cv = getenv("SPCIES_TVR_COOP");
        p.coop_im = !p.lds && !(cv && cv[0] == '0');
    }
    if (fista) {   // the kernel's restatement of the update phase's row layout must be the layout
        const FistaTvLayout a = fista_tv_layout(n, m, N);
        const FRows b = frows_of(n, m, N);
        if (a.AB != b.AB || a.Alpha != b.Alpha || a.Beta != b.Beta || a.Q != b.Q || a.R != b.R || a.QRi != b.QRi || a.LB != b.LB || a.UB != b.UB || a.rows != b.Bi)
            return fail(SPCIES_HIP_EINVAL, "MFMA4R (time-varying FISTA): row layout mismatch between fista_stream.hpp and admm_tvr_kernel.inc");
    } else {
        const TvLayout a = tv_layout(n, m, N);
        const Rows b = rows_of(n, m, N);
        if (a.AB != b.AB || a.Alpha != b.Alpha || a.Beta != b.Beta || a.Hi != b.Hi || a.Q != b.Q || a.R != b.R || a.LB != b.LB || a.UB != b.UB || a.Bi != b.Bi)
            return fail(SPCIES_HIP_EINVAL, "MFMA4R (time-varying): row layout mismatch between admm_stream.hpp and admm_tvr_kernel.inc");
    }
    if (p.lds) {
        const char *ev = getenv("SPCIES_HIP_RTC");
        if (ev && ev[0] == '0') { p.why = "the LDS form is run-time specialised and SPCIES_HIP_RTC=0"; return 0; }
        std::vector<std::string> nm;
        char name[160];
        for (int s = 0; s < 2; s++) {
            snprintf(name, sizeof(name), "spcies::tvr::%s_tvl_kernel<%d, %d, %d, %s, %s>", fista ? "fista" : "admm", n, m, N, terminal ? "true" : "false", s ? "true" : "false");
            nm.push_back(name);
        }
        snprintf(name, sizeof(name), "spcies::tvr::tv_bi_rolled_kernel<%d>", n);
        nm.push_back(name);
        snprintf(name, sizeof(name), "spcies::%s_tv_update_kernel<%d, %d, %s, false>", fista ? "fista" : "admm", n, m, terminal ? "true" : "false");
        nm.push_back(name);
        snprintf(name, sizeof(name), "spcies::tvr::tv_update_coop_kernel<%d, %d, %s, %s>", n, m, terminal ? "true" : "false", fista ? "true" : "false");
        nm.push_back(name);
        const std::string source = std::string(kSourceUpdate) + "\n" + kSourceSolve + "\n" + tvl_source();
        std::vector<std::string> extra = {"-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-amdgpu-mfma-vgpr-form"};
        for (const std::string &e : rtc::split_flags(getenv("SPCIES_TVR_RTC_FLAGS"))) extra.push_back(e);
        hipModule_t mod = nullptr;
        hipFunction_t fns[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        int rc = rtc::compile_module(source.c_str(), "spcies_admm_tvl_rtc.hip", nm, extra, &mod, fns);
        if (rc) { p.why = std::string("MFMA4R (time-varying, LDS form): run-time specialisation failed: ") + spcies_hip_last_error(); p.build_failed = true; p.lds = false; return 0; }
        p.module = mod;
        p.fn[0] = nullptr; p.fn[1] = fns[0]; p.fn[2] = fns[1];
        p.fn_bi = (void *)fns[2];
        p.fn_update = (void *)fns[3];
        p.fn_coop = (void *)fns[4];
        {   // SPCIES_TVL_COOP=0: the one-lane-per-instance update phase and inverses (the cross-check of the cooperative kernel: the same bits)
            const char *cv = getenv("SPCIES_TVL_COOP");
            p.coop = !(cv && cv[0] == '0');
        }
        p.fn_ms = nullptr;
        p.builtin = false;
        p.ok = true;
        p.why.clear();
        return 0;
    }
    // (registers: S_l - N KX doubles per lane; rounds 3-4: Bi and Bi', 2 N KX -, the Alpha blocks the LDS does not hold, the state - 2 N + 1; FISTA: 3 N -, constants
    // and temporaries: checked above)
    p.builtin = shape_built(n, m, N);
    if (const char *ev = getenv("SPCIES_TVR_RTC"))  // kernel experiments: re-specialise a built-in shape (with SPCIES_TVR_RTC_FLAGS)
        if (ev[0] == '1') p.builtin = false;
    if (!p.builtin || !p.update_builtin) {
        const char *ev = getenv("SPCIES_HIP_RTC");
        if (ev && ev[0] == '0') { p.why = "horizon not instantiated at build time and SPCIES_HIP_RTC=0"; return 0; }
        std::vector<std::string> nm;
        char name[160];
        for (int s = 0; s < 2; s++) {
            snprintf(name, sizeof(name), "spcies::tvr::%s_tvr_kernel<%d, %d, %d, %s, %s>", fista ? "fista" : "admm", n, m, N, terminal ? "true" : "false", s ? "true" : "false");
            nm.push_back(name);
        }
        snprintf(name, sizeof(name), "spcies::tvr::tv_ms_kernel<%d>", n);
        nm.push_back(name);
        if (!p.update_builtin) {  // an (n, m) without a build-time update phase: the same text as admm_stream.hpp compiles, specialised here
            snprintf(name, sizeof(name), "spcies::%s_tv_update_kernel<%d, %d, %s, true>", fista ? "fista" : "admm", n, m, terminal ? "true" : "false");
            nm.push_back(name);
            snprintf(name, sizeof(name), "spcies::tvr::tv_update_coop_kernel<%d, %d, %s, %s, 1>", n, m, terminal ? "true" : "false", fista ? "true" : "false");
            nm.push_back(name);
        }
        const std::string source = std::string(kSourceUpdate) + "\n" + kSourceSolve + (p.update_builtin ? "" : std::string("\n") + tvl_source());
        std::vector<std::string> extra = {"-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-amdgpu-mfma-vgpr-form"};
        for (const std::string &e : rtc::split_flags(getenv("SPCIES_TVR_RTC_FLAGS"))) extra.push_back(e);
        hipModule_t mod = nullptr;
        hipFunction_t fns[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        int rc = rtc::compile_module(source.c_str(), "spcies_admm_tvr_rtc.hip", nm, extra, &mod, fns);
        if (rc) { p.why = std::string("MFMA4R (time-varying): run-time specialisation failed: ") + spcies_hip_last_error(); p.build_failed = true; return 0; }
        p.module = mod;
        p.fn[0] = nullptr; p.fn[1] = fns[0]; p.fn[2] = fns[1];  // (fn[0]: the inverses' kernel of rounds 3-4; the update phase writes them now)
        p.fn_ms = (void *)fns[2];
        p.fn_update = p.update_builtin ? nullptr : (void *)fns[3];
        p.fn_coop = p.update_builtin ? nullptr : (void *)fns[4];
        p.builtin = false;  // (a build-time horizon of an (n, m) whose update phase is not: everything from the module)
    }
    p.ok = true;
    p.why.clear();
    return 0;
}

template <int n, int m, int N>
static int launch_shape(bool terminal, bool want_sol, const Args &a, const double *TRI, const double *T, double *TVS, const double *x0, const double *xr,
                        const double *ur, double *u, int *k, int *e, double *z, double *v, double *lam, unsigned grid, hipStream_t st) {
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

int launch(const Plan &p, bool want_sol, const Args &a, const double *TRI, const double *T, double *TVS, const double *x0, const double *xr,
           const double *ur, double *u, int *k, int *e, double *z, double *v, double *lam, int num_cu, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4R (time-varying) unavailable: %s", p.why.c_str());
    if (want_sol && !(z && v && lam)) return fail(SPCIES_HIP_EINVAL, "MFMA4R (time-varying): pass all of z, v, lambda or none");
    if (p.lds) {  // one 64-lane workgroup per instance, as many per CU as its LDS holds
        const unsigned g = (unsigned)std::min<long>(a.B, (long)num_cu * p.lds_per_cu);
        Args aa = a;
        void *params[] = {&aa, &TRI, &T, &TVS, &x0, &xr, &ur, &u, &k, &e, &z, &v, &lam};
        SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn[want_sol ? 2 : 1], g, 1, 1, 64, 1, 1, 0, st, params, nullptr));
        return 0;
    }
    const long groups = (a.B + 3) / 4;
    const unsigned grid = (unsigned)std::min<long>(groups, (long)num_cu);
    Args aa = a;
    aa.RA = p.coop_im ? p.rows_all : 0;
    if (p.builtin) {
#define X(nn, mm, NN) \
    if (p.n == nn && p.m == mm && p.N == NN) return launch_shape<nn, mm, NN>(p.terminal, want_sol, aa, TRI, T, TVS, x0, xr, ur, u, k, e, z, v, lam, grid, st);
        SPCIES_TVR_SHAPES(X)
#undef X
        return fail(SPCIES_HIP_ENOSUP, "MFMA4R (time-varying): bad build-time shape");
    }
    void *params[] = {&aa, &TRI, &T, &TVS, &x0, &xr, &ur, &u, &k, &e, &z, &v, &lam};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn[want_sol ? 2 : 1], grid, 1, 1, 256, 1, 1, 0, st, params, nullptr));
    return 0;
}

template <int n, int m, int N>
static int launch_fista_shape(bool terminal, bool want_sol, const Args &a, const double *T, const double *Ti, double *TVS, const double *x0, const double *xr,
                              const double *ur, double *u, int *k, int *e, double *z, double *lam, unsigned grid, hipStream_t st) {
#define SPCIES_TVR_GO(TT, SS) \
    hipLaunchKernelGGL((fista_tvr_kernel<n, m, N, TT, SS>), dim3(grid), dim3(256), 0, st, a, T, Ti, TVS, x0, xr, ur, u, k, e, z, lam)
    if (terminal) {
        if (want_sol) SPCIES_TVR_GO(true, true); else SPCIES_TVR_GO(true, false);
    } else {
        if (want_sol) SPCIES_TVR_GO(false, true); else SPCIES_TVR_GO(false, false);
    }
#undef SPCIES_TVR_GO
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_fista(const Plan &p, bool want_sol, const Args &a, const double *T, const double *Ti, double *TVS, const double *x0, const double *xr,
                 const double *ur, double *u, int *k, int *e, double *z, double *lam, int num_cu, hipStream_t st) {
    if (!p.ok || !p.fista) return fail(SPCIES_HIP_ENOSUP, "MFMA4R (time-varying FISTA) unavailable: %s", p.why.c_str());
    if (want_sol && !(z && lam)) return fail(SPCIES_HIP_EINVAL, "MFMA4R (time-varying FISTA): pass both z and lambda or neither");
    if (p.lds) {  // one 64-lane workgroup per instance, as many per CU as its LDS holds
        const unsigned g = (unsigned)std::min<long>(a.B, (long)num_cu * p.lds_per_cu);
        Args aa = a;
        void *params[] = {&aa, &T, &Ti, &TVS, &x0, &xr, &ur, &u, &k, &e, &z, &lam};
        SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn[want_sol ? 2 : 1], g, 1, 1, 64, 1, 1, 0, st, params, nullptr));
        return 0;
    }
    const long groups = (a.B + 3) / 4;
    const unsigned grid = (unsigned)std::min<long>(groups, (long)num_cu);
    Args aa = a;
    aa.RA = p.coop_im ? p.rows_all : 0;
    if (p.builtin) {
#define X(nn, mm, NN) \
    if (p.n == nn && p.m == mm && p.N == NN) return launch_fista_shape<nn, mm, NN>(p.terminal, want_sol, aa, T, Ti, TVS, x0, xr, ur, u, k, e, z, lam, grid, st);
        SPCIES_TVR_SHAPES(X)
#undef X
        return fail(SPCIES_HIP_ENOSUP, "MFMA4R (time-varying FISTA): bad build-time shape");
    }
    void *params[] = {&aa, &T, &Ti, &TVS, &x0, &xr, &ur, &u, &k, &e, &z, &lam};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn[want_sol ? 2 : 1], grid, 1, 1, 256, 1, 1, 0, st, params, nullptr));
    return 0;
}

// the build-time shapes of the update phase WITH the inverses (BI): instantiated here, in this translation unit, next to the kernels that consume them
template <int n, int m>
static void launch_update_builtin(const Plan &p, double c0, const double *Tc, const double *model, long model_stride, long B, long Bp, double *TVS,
                                  hipStream_t st) {
    const dim3 grid((unsigned)(Bp / 64)), block(64);
    const int N = p.N;
    if (p.fista) {
        if (p.terminal) hipLaunchKernelGGL((fista_tv_update_kernel<n, m, true, true>), grid, block, 0, st, N, Tc, model, model_stride, B, Bp, TVS);
        else hipLaunchKernelGGL((fista_tv_update_kernel<n, m, false, true>), grid, block, 0, st, N, Tc, model, model_stride, B, Bp, TVS);
    } else {
        if (p.terminal) hipLaunchKernelGGL((admm_tv_update_kernel<n, m, true, true>), grid, block, 0, st, N, c0, Tc, model, model_stride, B, Bp, TVS);
        else hipLaunchKernelGGL((admm_tv_update_kernel<n, m, false, true>), grid, block, 0, st, N, c0, Tc, model, model_stride, B, Bp, TVS);
    }
}

// the update phase's rows -> the block L D L' form the solve kernels consume (tv_ms_kernel: Alpha_l <- Bi_l Alpha_l, Bi_l <- Bi_l Bi_l', in place)
static int launch_ms(const Plan &p, long B, long Bp, double *TVS, hipStream_t st) {
    int N = p.N, row_bi = p.fista ? frows_of(p.n, p.m, p.N).Bi : rows_of(p.n, p.m, p.N).Bi;
    int row_alpha = p.fista ? frows_of(p.n, p.m, p.N).Alpha : rows_of(p.n, p.m, p.N).Alpha;
    const unsigned grid = (unsigned)((B + 63) / 64);
    if (p.fn_ms) {
        void *params[] = {&N, &row_bi, &row_alpha, &B, &Bp, &TVS};
        SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn_ms, grid, 1, 1, 64, 1, 1, 0, st, params, nullptr));
        return 0;
    }
    if (p.n == 6) hipLaunchKernelGGL((tv_ms_kernel<6>), dim3(grid), dim3(64), 0, st, N, row_bi, row_alpha, B, Bp, TVS);
    else if (p.n == 12) hipLaunchKernelGGL((tv_ms_kernel<12>), dim3(grid), dim3(64), 0, st, N, row_bi, row_alpha, B, Bp, TVS);
    else return fail(SPCIES_HIP_ENOSUP, "time-varying update phase: no L D L' transform for n=%d", p.n);
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_update(const Plan &p, double c0, const double *Tc, const double *model, long model_stride, long B, long Bp, double *TVS, hipStream_t st) {
    if (p.ok && p.lds && p.coop) {  // the cooperative update phase (tv_update_coop_kernel): 8 / 16 / 32 lanes per instance, the small rows and the packed triangles of the Bi_l out
        int N = p.N;
        const int g = coop_instances_per_wavefront(p.n);
        void *up[] = {&N, &c0, &Tc, &model, &model_stride, &B, &Bp, &TVS};
        SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn_coop, (unsigned)((B + g - 1) / g), 1, 1, 64, 1, 1, 0, st, up, nullptr));
        return 0;
    }
    if (p.ok && p.lds) {  // cross-check path: update phase without the inverses (rolled past n = 16), then the inverses by the rolled kernel (packed triangles)
        int N = p.N, row_beta = p.fista ? frows_of(p.n, p.m, p.N).Beta : rows_of(p.n, p.m, p.N).Beta;
        int row_bi = p.fista ? frows_of(p.n, p.m, p.N).Bi : rows_of(p.n, p.m, p.N).Bi;
        if (p.fista) {
            void *up[] = {&N, &Tc, &model, &model_stride, &B, &Bp, &TVS};
            SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn_update, (unsigned)(Bp / 64), 1, 1, 64, 1, 1, 0, st, up, nullptr));
        } else {
            void *up[] = {&N, &c0, &Tc, &model, &model_stride, &B, &Bp, &TVS};
            SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn_update, (unsigned)(Bp / 64), 1, 1, 64, 1, 1, 0, st, up, nullptr));
        }
        void *bp[] = {&N, &row_beta, &row_bi, &B, &Bp, &TVS};
        SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn_bi, (unsigned)((B + 63) / 64), 1, 1, 64, 1, 1, 0, st, bp, nullptr));
        return 0;
    }
    if (p.ok && p.coop_im) {  // the register form's images from the cooperative update phase, instance-major (FORM 1): S_l and M_l rows, no transform kernel
        if (p.update_builtin) return launch_coop_builtin(p.n, p.m, p.N, p.terminal, p.fista, c0, Tc, model, model_stride, B, Bp, TVS, st);
        if (!p.fn_coop) return fail(SPCIES_HIP_ENOSUP, "time-varying update phase: no cooperative kernel for n=%d m=%d", p.n, p.m);
        int N = p.N;
        const int g = coop_instances_per_wavefront(p.n);
        void *up[] = {&N, &c0, &Tc, &model, &model_stride, &B, &Bp, &TVS};
        SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn_coop, (unsigned)((B + g - 1) / g), 1, 1, 64, 1, 1, 0, st, up, nullptr));
        return 0;
    }
    if (p.ok && p.update_builtin) {
        if (p.n == 6 && p.m == 2) launch_update_builtin<6, 2>(p, c0, Tc, model, model_stride, B, Bp, TVS, st);
        else launch_update_builtin<12, 2>(p, c0, Tc, model, model_stride, B, Bp, TVS, st);
        SPCIES_HIP_CHECK(hipGetLastError());
        return launch_ms(p, B, Bp, TVS, st);
    }
    if (!p.ok || !p.fn_update) return fail(SPCIES_HIP_ENOSUP, "time-varying update phase: no run-time specialised kernel for n=%d m=%d", p.n, p.m);
    int N = p.N;
    if (p.fista) {
        void *params[] = {&N, &Tc, &model, &model_stride, &B, &Bp, &TVS};
        SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn_update, (unsigned)(Bp / 64), 1, 1, 64, 1, 1, 0, st, params, nullptr));
    } else {
        void *params[] = {&N, &c0, &Tc, &model, &model_stride, &B, &Bp, &TVS};
        SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn_update, (unsigned)(Bp / 64), 1, 1, 64, 1, 1, 0, st, params, nullptr));
    }
    return launch_ms(p, B, Bp, TVS, st);
}

}  // namespace tvr
}  // namespace spcies
