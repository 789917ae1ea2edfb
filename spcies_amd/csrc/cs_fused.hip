// Host side of the FUSED MPCT-cs variant (cs_fused.hpp): the dense operator, table layout, launch, run-time specialisation.
#include "cs_fused.hpp"

#include <cmath>

#include "cs_fused_kernel.inc"
#include "rtc_common.hpp"

namespace spcies {
namespace csfused {

static const char *const kCsFusedSource =
#include "cs_fused_src.inc"
    ;

// ---- shapes instantiated at build time: (n, m, N).  Everything else is compiled by hiprtc at create time.
#define SPCIES_CSFUSED_SHAPES(X) X(12, 2, 15)

namespace {

struct Dims {  // mirrors Shape<> of the kernel file
    int dnm, dim, NR, NRP, NXS, NUS, NE, NK, JC, NCH, CHB;
    Dims(int n, int m, int N) {
        dnm = 2 * (n + m);
        dim = N * dnm;
        NR = (dim + 15) / 16;
        NRP = (NR + 1) / 2 * 2;
        NXS = (n + 3) / 4;
        NUS = (m + 3) / 4;
        NE = 2 * NXS + NUS;
        NK = 4 * NR;
        JC = (SPCIES_CSFUSED_CHUNK / (NRP * 512)) > 0 ? (SPCIES_CSFUSED_CHUNK / (NRP * 512)) : 1;
        while (JC > 1 && 3 * ((JC * NRP * 512 + 4095) / 4096 * 4096) + 2 * 16 * NR * 8 > 160 * 1024) JC--;  // (as Shape::pick_jc)
        NCH = ((NK + JC - 1) / JC + 2) / 3 * 3;
        CHB = (JC * NRP * 512 + 4095) / 4096 * 4096;
    }
};

int builtin_index(int n, int m, int N) {
    int idx = 0;
#define X(nn, mm, NN)                               \
    if (n == nn && m == mm && N == NN) return idx; \
    idx++;
    SPCIES_CSFUSED_SHAPES(X)
#undef X
    return -1;
}

template <int n, int m, int N>
int launch_builtin(const Args &a, const double *ME, const double *PRO, const double *C, const double *x0, const double *xr, const double *ur,
                   double *u, int *k, int *e, double *z, double *v, double *lam, unsigned grid, hipStream_t st) {
    if (z || v || lam)
        hipLaunchKernelGGL((cs_fused_kernel<n, m, N, true>), dim3(grid), dim3(kNWV * 64), 0, st, a, ME, PRO, C, x0, xr, ur, u, k, e, z, v, lam);
    else
        hipLaunchKernelGGL((cs_fused_kernel<n, m, N, false>), dim3(grid), dim3(kNWV * 64), 0, st, a, ME, PRO, C, x0, xr, ur, u, k, e, nullptr,
                           nullptr, nullptr);
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

// The reference's z-update as a linear map (code_MPCT_ADMM_cs_C.c:113-164):  z = Hi q_hat + HiA (L D L')^-1 (AHi q_hat - bb)
// (AHi = -Aeq Hhat^-1, HiA = -Hhat^-1 Aeq', Hi = -Hhat^-1 as the constructor prints them; bb = [x0; 0])
struct Operator {
    const Host &h;
    std::vector<double> mu;
    explicit Operator(const Host &hh) : h(hh), mu(hh.nrow) {}
    void apply(const double *q_hat, const double *bb, double *z) {
        const int nrow = h.nrow, dim = h.dim;
        for (int i = 0; i < nrow; i++) {
            double r = 0.0;
            if (q_hat)
                for (int j = h.AHi_row[i]; j < h.AHi_row[i + 1]; j++) r += h.AHi_val[j] * q_hat[h.AHi_col[j]];
            mu[i] = r - (bb ? bb[i] : 0.0);
        }
        for (int i = 0; i < nrow; i++)
            for (int j = h.L_col[i]; j < h.L_col[i + 1]; j++) mu[h.L_row[j]] -= h.L_val[j] * mu[i];
        for (int j = 0; j < nrow; j++) mu[j] *= h.Dinv[j];
        for (int i = nrow - 1; i >= 0; i--)
            for (int j = h.L_col[i]; j < h.L_col[i + 1]; j++) mu[i] -= h.L_val[j] * mu[h.L_row[j]];
        for (int i = 0; i < dim; i++) {
            double acc = 0.0;
            if (q_hat)
                for (int j = h.Hi_row[i]; j < h.Hi_row[i + 1]; j++) acc += h.Hi_val[j] * q_hat[h.Hi_col[j]];
            for (int j = h.HiA_row[i]; j < h.HiA_row[i + 1]; j++) acc += h.HiA_val[j] * mu[h.HiA_col[j]];
            z[i] = acc;
        }
    }
};

int put(std::vector<double> &flat, const std::vector<double> &v) {
    const int off = (int)flat.size();
    flat.insert(flat.end(), v.begin(), v.end());
    while (flat.size() % 8) flat.push_back(0.0);
    return off;
}

}  // namespace

void plan_free(Plan &p) {
    if (p.d_ME) hipFree(p.d_ME);
    if (p.d_PRO) hipFree(p.d_PRO);
    if (p.d_C) hipFree(p.d_C);
    if (p.module) rtc::unload_module((hipModule_t)p.module);
    p.d_ME = p.d_PRO = p.d_C = nullptr;
    p.module = nullptr;
    p.ok = false;
}

int plan_build(Plan &p, const Host &h) {
    const int n = h.n, m = h.m, N = h.N;
    const Dims D(n, m, N);
    p.ok = false;
    if (D.dim != h.dim) { p.why = "unexpected MPCT-cs dimensions"; return 0; }
    if (D.NR > 30) { p.why = "FUSED: more than 30 row registers (2 N (n + m) > 480)"; return 0; }
    if (3 * D.CHB + 2 * 16 * D.NR * 8 > 160 * 1024) { p.why = "FUSED: chunk buffers exceed the LDS"; return 0; }
    const int dim = h.dim, dnm = D.dnm, NP = 16 * D.NR;
    // ---- dense operator, column by column: Mz = d z / d q_hat  [dim][dim],  Kb = d z / d x0  [dim][n]
    Operator op(h);
    std::vector<double> Mz((size_t)dim * dim), Kb((size_t)dim * n), e(dim, 0.0), bb(h.nrow, 0.0), col(dim);
    for (int j = 0; j < dim; j++) {
        e[j] = 1.0;
        op.apply(e.data(), nullptr, col.data());
        e[j] = 0.0;
        for (int i = 0; i < dim; i++) Mz[(size_t)i * dim + j] = col[i];
    }
    for (int j = 0; j < n; j++) {
        bb[j] = 1.0;
        op.apply(nullptr, bb.data(), col.data());
        bb[j] = 0.0;
        for (int i = 0; i < dim; i++) Kb[(size_t)i * n + j] = col[i];
    }
    for (double x : Mz)
        if (!std::isfinite(x)) { p.why = "non-finite operator"; return 0; }
    // ---- is the dense form as good as the sparse one?  A random q_hat / x0 through both; an ill-conditioned W (cond 1e9 at the
    // C4 shape) shows here and leaves the controller on TILE
    {
        std::vector<double> qh(dim), zz(dim);
        unsigned long long sd = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() { sd = sd * 6364136223846793005ull + 1442695040888963407ull; return (double)((sd >> 11) & 0xFFFFF) / 524288.0 - 1.0; };
        for (double &x : qh) x = rnd();
        for (int j = 0; j < n; j++) bb[j] = rnd();
        op.apply(qh.data(), bb.data(), zz.data());
        double worst = 0.0, scale = 1.0;
        for (int i = 0; i < dim; i++) {
            double acc = 0.0;
            for (int j = 0; j < dim; j++) acc += Mz[(size_t)i * dim + j] * qh[j];
            for (int j = 0; j < n; j++) acc += Kb[(size_t)i * n + j] * bb[j];
            worst = std::max(worst, std::fabs(acc - zz[i]));
            scale = std::max(scale, std::fabs(zz[i]));
        }
        for (int j = 0; j < n; j++) bb[j] = 0.0;
        if (!(worst <= 1e-11 * scale)) {
            char msg[160];
            snprintf(msg, sizeof(msg), "FUSED: dense operator differs from the sparse one by %.1e (ill-conditioned W): TILE / STREAM", worst / scale);
            p.why = msg;
            return 0;
        }
    }
    std::vector<double> rho(dim, h.rho);
    if (!h.scalar_rho)
        for (int j = 0; j < dim; j++) rho[j] = h.rho_v[j];
    // ---- iteration table: ME = Mz diag(rho), internal rows / columns = natural order, pads zero
    std::vector<double> tab((size_t)D.NCH * (D.CHB / 8), 0.0);
    auto me = [&](int r, int c) { return (r < dim && c < dim) ? Mz[(size_t)r * dim + c] * rho[c] : 0.0; };
    for (int c = 0; c < D.NCH; c++)
        for (int jj = 0; jj < D.JC; jj++) {
            const int J = c * D.JC + jj;
            if (J >= D.NK) continue;  // (zero blocks pad the last chunks)
            for (int R = 0; R < D.NR; R++)
                for (int l = 0; l < 64; l++) {
                    const int k = l >> 4, b = (l >> 2) & 3, i = l & 3;
                    tab[(size_t)c * (D.CHB / 8) + ((size_t)(jj * (D.NRP / 2) + R / 2) * 64 + l) * 2 + (R & 1)] = me(16 * R + 4 * b + i, 4 * J + k);
                }
        }
    // ---- prologue table: c = Kb x0 + Mr xr + Mu ur,  Mr = (sum over stages of Mz[:, x_s columns]) Tz,  Mu likewise with Sz
    // (q = [0; Tz xr; 0; Sz ur] repeated over the stages, :76-85 and :103-109); columns: x0 | xr | ur in k-slabs of four
    std::vector<double> Pm((size_t)NP * 4 * D.NE, 0.0);
    for (int i = 0; i < dim; i++) {
        double *row = &Pm[(size_t)i * 4 * D.NE];
        for (int c = 0; c < n; c++) row[c] = Kb[(size_t)i * n + c];
        for (int j = 0; j < n; j++) {
            double sj = 0.0;
            for (int s = 0; s < N; s++) sj += Mz[(size_t)i * dim + s * dnm + n + j];
            for (int c = 0; c < n; c++) row[4 * D.NXS + c] += sj * h.Tz[j * n + c];
        }
        for (int j = 0; j < m; j++) {
            double sj = 0.0;
            for (int s = 0; s < N; s++) sj += Mz[(size_t)i * dim + s * dnm + 2 * n + m + j];
            for (int c = 0; c < m; c++) row[8 * D.NXS + c] += sj * h.Sz[j * m + c];
        }
    }
    std::vector<double> ptab((size_t)D.NE * (D.NRP / 2) * 128, 0.0);
    for (int J = 0; J < D.NE; J++)
        for (int R = 0; R < D.NR; R++)
            for (int l = 0; l < 64; l++) {
                const int k = l >> 4, b = (l >> 2) & 3, i = l & 3;
                ptab[((size_t)(J * (D.NRP / 2) + R / 2) * 64 + l) * 2 + (R & 1)] = Pm[(size_t)(16 * R + 4 * b + i) * 4 * D.NE + 4 * J + k];
            }
    std::vector<double> flat, lbv(NP, 0.0), ubv(NP, 0.0), rhov(NP, 1.0);
    for (int r = 0; r < dim; r++) { lbv[r] = h.LB[r]; ubv[r] = h.UB[r]; rhov[r] = rho[r]; }
    p.oLB = put(flat, lbv);
    p.oUB = put(flat, ubv);
    p.oRho = put(flat, rhov);
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_ME, tab.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_ME, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_PRO, ptab.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_PRO, ptab.data(), ptab.size() * sizeof(double), hipMemcpyHostToDevice));
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_C, flat.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_C, flat.data(), flat.size() * sizeof(double), hipMemcpyHostToDevice));
    p.n = n; p.m = m; p.N = N; p.NR = D.NR; p.NCH = D.NCH; p.CHB = D.CHB;
    hipDeviceProp_t prop;
    int dev = 0;
    SPCIES_HIP_CHECK(hipGetDevice(&dev));
    SPCIES_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    p.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    p.builtin = builtin_index(n, m, N);
    const char *force = getenv("SPCIES_CSFUSED_RTC");  // kernel experiments: re-specialise a built-in shape (with SPCIES_CSFUSED_FLAGS)
    if (force && force[0] == '1') p.builtin = -1;
    if (p.builtin < 0) {
        const char *ev = getenv("SPCIES_HIP_RTC");
        if (ev && ev[0] == '0') { p.why = "shape not instantiated at build time and SPCIES_HIP_RTC=0"; return 0; }
        std::vector<std::string> names, extra = {"-mllvm", "-amdgpu-mfma-vgpr-form", "-mllvm", "-pragma-unroll-threshold=1000000"};
        for (int s = 0; s < 2; s++) {
            char nm[160];
            snprintf(nm, sizeof(nm), "spcies::csfused::cs_fused_kernel<%d, %d, %d, %s>", n, m, N, s ? "true" : "false");
            names.push_back(nm);
        }
        if (const char *fl = getenv("SPCIES_CSFUSED_FLAGS")) {
            std::string tok;
            for (const char *c = fl;; c++) {
                if (*c == ' ' || *c == '\0') {
                    if (!tok.empty()) extra.push_back(tok);
                    tok.clear();
                    if (!*c) break;
                } else {
                    tok.push_back(*c);
                }
            }
        }
        hipModule_t mod = nullptr;
        hipFunction_t fns[2] = {nullptr, nullptr};
        if (rtc::compile_module(kCsFusedSource, "spcies_cs_fused.hip", names, extra, &mod, fns) != 0) {
            p.why = g_last_error;
            p.build_failed = true;
            return 0;  // not an error: AUTO falls back, the reason is reported when FUSED is asked for
        }
        p.module = mod;
        p.fn[0] = fns[0];
        p.fn[1] = fns[1];
    }
    p.ok = true;
    p.why.clear();
    return 0;
}

int launch(Plan &p, int k_max, double tol, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
           int *k, int *e, double *z, double *v, double *lam, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "FUSED variant unavailable: %s", p.why.c_str());
    Args a{};
    a.B = B; a.ref_stride = ref_stride; a.k_max = k_max; a.tol = tol; a.oLB = p.oLB; a.oUB = p.oUB; a.oRho = p.oRho;
    const long groups = (B + 31) / 32;
    const unsigned grid = (unsigned)std::min<long>(groups, p.num_cu);
    const double *ME = p.d_ME, *PRO = p.d_PRO, *C = p.d_C;
    if (p.builtin >= 0) {
        int idx = 0;
#define X(nn, mm, NN)                                                                                                  \
    if (p.builtin == idx) return launch_builtin<nn, mm, NN>(a, ME, PRO, C, x0, xr, ur, u, k, e, z, v, lam, grid, st); \
    idx++;
        SPCIES_CSFUSED_SHAPES(X)
#undef X
        return fail(SPCIES_HIP_ENOSUP, "FUSED: bad build-time shape index");
    }
    const bool want_sol = z || v || lam;
    void *params[] = {&a, &ME, &PRO, &C, &x0, &xr, &ur, &u, &k, &e, &z, &v, &lam};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn[want_sol ? 1 : 0], grid, 1, 1, kNWV * 64, 1, 1, 0, st, params, nullptr));
    return 0;
}

}  // namespace csfused
}  // namespace spcies
