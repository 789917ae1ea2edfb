// Host side of the FUSED HMPC variant (hmpc_fused.hpp): table layout, constants, launch, run-time specialisation.
#include "hmpc_fused.hpp"

#include <cmath>

#include "hmpc_fused_kernel.inc"
#include "rtc_common.hpp"

namespace spcies {
namespace hfused {

static const char *const kFusedSource =
#include "hmpc_fused_src.inc"
    ;

// ---- shapes instantiated at build time: (n, m, N, SYM, USE_SOC).  Everything else is compiled by hiprtc at create time.
#define SPCIES_HFUSED_SPLIT_SHAPES(X) X(12, 2, 15, true, false)

namespace {

struct Dims {  // mirrors Shape<> of the kernel file for run-time values
    int nm, dim, n_soc, n_s, n_box, NZ, NC, NR, NX, NK, JC, NCH, CHB, o0;
    Dims(int n, int m, int N, bool use_soc) {
        nm = n + m;
        dim = (N - 1) * nm + m + 3 * nm;
        n_soc = use_soc ? 2 * nm : nm;
        n_s = 3 * n_soc;
        n_box = dim - 3 * nm;
        NZ = (dim + 15) / 16;
        NC = (n_soc + 3) / 4;
        NR = NZ + NC;
        NX = (n + 3) / 4;
        NK = 4 * NR + NX + 1;
        JC = (40960 / (NR * 512)) > 0 ? (40960 / (NR * 512)) : 1;
        NCH = (NK + JC - 1) / JC;
        CHB = ((JC * NR * 512 + 1023) / 1024) * 1024;
        o0 = (N - 1) * nm + m;
    }
};

int builtin_index(int n, int m, int N, bool sym, bool use_soc) {
    int idx = 0;
#define X(nn, mm, NN, SS, UU)                                                     \
    if (n == nn && m == mm && N == NN && sym == SS && use_soc == UU) return idx; \
    idx++;
    SPCIES_HFUSED_SPLIT_SHAPES(X)
#undef X
    return -1;
}

template <int n, int m, int N, bool SYM, bool USE_SOC>
int launch_builtin(const Args &a, const double *ME, const double *C, const double *x0, const double *xr, const double *ur, double *u,
                   int *k, int *e, double *const *f, bool want_sol, unsigned grid, hipStream_t st) {
    if (want_sol)
        hipLaunchKernelGGL((hmpc_split_fused_kernel<n, m, N, SYM, USE_SOC, true>), dim3(grid), dim3(512), 0, st, a, ME, C, x0, xr, ur, u, k, e,
                           f[0], f[1], f[2], f[3], f[4], f[5]);
    else
        hipLaunchKernelGGL((hmpc_split_fused_kernel<n, m, N, SYM, USE_SOC, false>), dim3(grid), dim3(512), 0, st, a, ME, C, x0, xr, ur, u, k, e,
                           nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace

void plan_free(Plan &p) {
    if (p.d_ME) hipFree(p.d_ME);
    if (p.d_C) hipFree(p.d_C);
    if (p.module) hipModuleUnload((hipModule_t)p.module);
    p.d_ME = p.d_C = nullptr;
    p.module = nullptr;
    p.ok = false;
}

int plan_build_split(Plan &p, const SplitHost &h) {
    const int n = h.n, m = h.m, N = h.N;
    const Dims D(n, m, N, h.use_soc != 0);
    if (D.dim != h.dim || D.n_s != h.n_s || D.n_soc != h.n_soc) { p.why = "unexpected HMPC dimensions"; return 0; }
    if (D.NR > 24) { p.why = "FUSED: more than 24 row registers (dim + padded cones > 384 rows)"; return 0; }
    const int np = h.dim + h.n_s, nc = h.n_eq + h.n_s;
    // internal row -> row of (z, s), or -1 for a pad
    const int NP = 16 * D.NR;
    std::vector<int> orig(NP, -1);
    for (int r = 0; r < h.dim; r++) orig[r] = r;
    for (int t = 0; t < D.n_soc; t++)
        for (int i = 0; i < 3; i++) orig[16 * D.NZ + 4 * t + i] = h.dim + 3 * t + i;
    // extended matrix: hat = -M1 q_hat + M2 bh,  M2 bh = c_const + (-M2[:, :n] A) x0  (:97-104, :174-190)
    const int NKP = D.NCH * D.JC, ncol = 4 * NKP;
    std::vector<double> Mx((size_t)NP * ncol, 0.0);
    for (int ri = 0; ri < NP; ri++) {
        const int ro = orig[ri];
        if (ro < 0) continue;
        double *row = &Mx[(size_t)ri * ncol];
        for (int ci = 0; ci < NP; ci++)
            if (orig[ci] >= 0) row[ci] = -h.M1[(size_t)ro * np + orig[ci]];
        for (int c = 0; c < n; c++) {
            double acc = 0.0;
            for (int j = 0; j < n; j++) acc -= h.M2[(size_t)ro * nc + j] * h.A[j * n + c];
            row[NP + c] = acc;
        }
        double cc = 0.0;
        for (int j = n; j < nc; j++) cc += h.M2[(size_t)ro * nc + j] * h.bh_nat[j];
        row[NP + 4 * D.NX] = cc;
    }
    for (double x : Mx)
        if (!std::isfinite(x)) { p.why = "non-finite M1 / M2"; return 0; }
    // the table in issue order: chunk | k-slab in chunk | row register | lane (k = l >> 4, b = (l >> 2) & 3, i = l & 3)
    std::vector<double> tab((size_t)D.NCH * (D.CHB / 8), 0.0);
    for (int c = 0; c < D.NCH; c++)
        for (int jj = 0; jj < D.JC; jj++) {
            const int J = c * D.JC + jj;
            for (int R = 0; R < D.NR; R++)
                for (int l = 0; l < 64; l++) {
                    const int k = l >> 4, b = (l >> 2) & 3, i = l & 3;
                    tab[(size_t)c * (D.CHB / 8) + (size_t)(jj * D.NR + R) * 64 + l] = Mx[(size_t)(16 * R + 4 * b + i) * ncol + 4 * J + k];
                }
        }
    // constants: QQ, Te, Se, bounds per internal z row, cone shifts per internal cone row
    std::vector<double> flat;
    auto put = [&](const std::vector<double> &v) {
        const int off = (int)flat.size();
        flat.insert(flat.end(), v.begin(), v.end());
        while (flat.size() % 8) flat.push_back(0.0);
        return off;
    };
    std::vector<double> lbv(16 * D.NZ, 0.0), ubv(16 * D.NZ, 0.0), d1(16 * D.NC, 0.0), d2(16 * D.NC, 0.0);
    for (int r = 0; r < h.dim; r++) {
        lbv[r] = r < D.n_box ? h.LB[r] : -1e300;
        ubv[r] = r < D.n_box ? h.UB[r] : 1e300;
    }
    if (!h.use_soc)
        for (int t = 0; t < D.n_soc; t++)
            for (int i = 0; i < 4; i++) { d1[4 * t + i] = h.LBy[t]; d2[4 * t + i] = h.UBy[t]; }
    p.oQQ = put(std::vector<double>(h.QQ, h.QQ + n * n));
    p.oTe = put(std::vector<double>(h.Te, h.Te + n * n));
    p.oSe = put(std::vector<double>(h.Se, h.Se + m * m));
    p.oLB = put(lbv);
    p.oUB = put(ubv);
    p.oD1 = put(d1);
    p.oD2 = put(d2);
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_ME, tab.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_ME, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_C, flat.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_C, flat.data(), flat.size() * sizeof(double), hipMemcpyHostToDevice));
    p.n = n; p.m = m; p.N = N; p.use_soc = h.use_soc; p.symmetric = h.symmetric;
    p.NR = D.NR; p.NK = D.NK; p.NCH = D.NCH; p.CHB = D.CHB;
    hipDeviceProp_t prop;
    int dev = 0;
    SPCIES_HIP_CHECK(hipGetDevice(&dev));
    SPCIES_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    p.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    p.builtin = builtin_index(n, m, N, h.symmetric != 0, h.use_soc != 0);
    if (p.builtin < 0) {  // not among the build-time shapes: specialise now (SPCIES_HIP_RTC=0 turns it off)
        const char *ev = getenv("SPCIES_HIP_RTC");
        if (ev && ev[0] == '0') { p.why = "shape not instantiated at build time and SPCIES_HIP_RTC=0"; return 0; }
        std::vector<std::string> names;
        for (int s = 0; s < 2; s++) {
            char nm[160];
            snprintf(nm, sizeof(nm), "spcies::hfused::hmpc_split_fused_kernel<%d, %d, %d, %s, %s, %s>", n, m, N, h.symmetric ? "true" : "false",
                     h.use_soc ? "true" : "false", s ? "true" : "false");
            names.push_back(nm);
        }
        hipModule_t mod = nullptr;
        hipFunction_t fns[2] = {nullptr, nullptr};
        if (rtc::compile_module(kFusedSource, "spcies_hmpc_fused.hip", names, {}, &mod, fns) != 0) {
            p.why = g_last_error;
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

int launch_split(Plan &p, int k_max, double tol_p, double tol_d, double rho, double rho_i, double sigma, double sigma_i, double alpha,
                 const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u, int *k, int *e,
                 double *const *f, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "FUSED variant unavailable: %s", p.why.c_str());
    Args a{};
    a.B = B; a.ref_stride = ref_stride; a.k_max = k_max; a.tol_p = tol_p; a.tol_d = tol_d; a.rho = rho; a.rho_i = rho_i;
    a.sigma = sigma; a.sigma_i = sigma_i; a.alpha = alpha;
    a.oQQ = p.oQQ; a.oTe = p.oTe; a.oSe = p.oSe; a.oLB = p.oLB; a.oUB = p.oUB; a.oD1 = p.oD1; a.oD2 = p.oD2;
    bool want_sol = false;
    for (int i = 0; i < 6; i++) want_sol |= f[i] != nullptr;
    const long groups = (B + 31) / 32;
    const unsigned grid = (unsigned)std::min<long>(groups, p.num_cu);
    const double *ME = p.d_ME, *C = p.d_C;
    if (p.builtin >= 0) {
        int idx = 0;
#define X(nn, mm, NN, SS, UU)                                                                                            \
    if (p.builtin == idx) return launch_builtin<nn, mm, NN, SS, UU>(a, ME, C, x0, xr, ur, u, k, e, f, want_sol, grid, st); \
    idx++;
        SPCIES_HFUSED_SPLIT_SHAPES(X)
#undef X
        return fail(SPCIES_HIP_ENOSUP, "FUSED: bad build-time shape index");
    }
    double *f0 = f[0], *f1 = f[1], *f2 = f[2], *f3 = f[3], *f4 = f[4], *f5 = f[5];
    void *params[] = {&a, &ME, &C, &x0, &xr, &ur, &u, &k, &e, &f0, &f1, &f2, &f3, &f4, &f5};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn[want_sol ? 1 : 0], grid, 1, 1, 512, 1, 1, 0, st, params, nullptr));
    return 0;
}

}  // namespace hfused
}  // namespace spcies
