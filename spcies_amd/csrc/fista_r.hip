// Host side of the MFMA4R variant (fista_r.hpp): table packer, kernel specialisation (hiprtc / build-time), launch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/spcies_hip.h"
#include "fista_r.hpp"
#define SPCIES_FR_NFULL 1  // the build-time instantiations serve controllers whose n is a multiple of 4 (plan_build checks)
#define SPCIES_FR_PD 7     // ... and keep the d of the last SEVEN blocks in registers: at configs[2] 33.1 ms against 34.1 with three (round 4 sweep:
                           // 2: 34.5, 3: 34.1, 4: 33.9, 5: 33.6, 6: 35.0, 7: 33.1, 8: 35.3, 10: 34.9 - the allocator's luck as much as the traffic);
                           // run-time specialised shapes keep three (SPCIES_FR_PD in the environment changes it)
#include "fista_r_kernel.inc"
#include "rtc_common.hpp"

namespace spcies {
namespace fr {

static const char *const kSource =
#include "fista_r_src.inc"
    ;

namespace {

struct DM {
    int r = 0, c = 0;
    std::vector<double> a;
    DM() {}
    DM(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};
DM mul(const DM &A, const DM &B) {
    DM C(A.r, B.c);
    for (int i = 0; i < A.r; i++)
        for (int k = 0; k < A.c; k++)
            for (int j = 0; j < B.c; j++) C(i, j) += A(i, k) * B(k, j);
    return C;
}
DM tr(const DM &A) {
    DM T(A.c, A.r);
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < A.c; j++) T(j, i) = A(i, j);
    return T;
}
DM neg(DM A) {
    for (auto &x : A.a) x = -x;
    return A;
}
// inverse of the upper-triangular Beta block as the reference stores it (reciprocal diagonal, compute_laxMPC_FISTA_ingredients.m)
DM beta_inverse(const double *beta, int n) {
    DM U(n, n), X(n, n);
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) U(i, j) = (i == j) ? 1.0 / beta[i * n + j] : beta[i * n + j];
    for (int j = 0; j < n; j++) {
        X(j, j) = 1.0 / U(j, j);
        for (int i = j - 1; i >= 0; i--) {
            double s = 0.0;
            for (int k = i + 1; k <= j; k++) s += U(i, k) * X(k, j);
            X(i, j) = -s / U(i, i);
        }
    }
    return X;
}
// appends the non-zero 4x4 blocks of M in issue order (J outer, I inner), element-interleaved pairs
struct BlockWriter {
    double *base;
    int cursor = 0;
    bool structure_ok = true;
    explicit BlockWriter(double *b) : base(b) {}
    void emit(const DM &M, int KI, int KJ, int pat) {
        auto at = [&](int i, int j) { return (i < M.r && j < M.c) ? M(i, j) : 0.0; };
        for (int J = 0; J < KJ; J++)
            for (int I = 0; I < KI; I++) {
                if (!blk_nz(I, J, pat)) {
                    for (int i = 0; i < 4; i++)
                        for (int k = 0; k < 4; k++)
                            if (at(4 * I + i, 4 * J + k) != 0.0) structure_ok = false;
                    continue;
                }
                double *t = base + (size_t)(cursor / 2) * 32 + (cursor % 2);
                for (int k = 0; k < 4; k++)
                    for (int i = 0; i < 4; i++) t[2 * (k * 4 + i)] = at(4 * I + i, 4 * J + k);
                cursor++;
            }
    }
};

template <int KX, int KS>
int pack(Plan &p, const Host &h, std::vector<double> &tab) {
    using LY = Layout<KX, KS>;
    const int n = h.n, m = h.m, N = h.N, nm = n + m;
    tab.assign((size_t)LY::table_doubles(N), 0.0);
    DM AB(n, nm);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < nm; j++) AB(i, j) = h.AB[(size_t)i * nm + j];
    auto rc = [&](int which) { return tab.data() + (size_t)which * LY::RC; };
    for (int j = 0; j < m; j++) {
        rc(LY::C_HD0)[n + j] = h.QRi[n + j];
        rc(LY::C_LB0)[n + j] = h.LB[n + j];
        rc(LY::C_UB0)[n + j] = h.UB[n + j];
        rc(LY::C_QR)[n + j] = h.R[j];
    }
    for (int j = 0; j < n; j++) {
        rc(LY::C_QR)[j] = h.Q[j];
        rc(LY::C_TD)[j] = h.terminal ? h.Td[j] : 0.0;
    }
    for (int j = 0; j < nm; j++) {
        rc(LY::C_HDM)[j] = h.QRi[j];
        rc(LY::C_LBM)[j] = h.LB[j];
        rc(LY::C_UBM)[j] = h.UB[j];
    }
    if (h.terminal)
        for (int j = 0; j < n; j++) {
            rc(LY::C_HDT)[j] = h.Ti[j];
            rc(LY::C_LBT)[j] = h.LB[j];
            rc(LY::C_UBT)[j] = h.UB[j];
        }
    std::vector<DM> Bi(N), Al(N - 1);
    for (int l = 0; l < N; l++) Bi[l] = beta_inverse(h.Beta + (size_t)l * n * n, n);
    for (int l = 0; l < N - 1; l++) {
        Al[l] = DM(n, n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Al[l](i, j) = h.Alpha[((size_t)l * n + i) * n + j];
    }
    const DM Zero(n, n), nABt = neg(tr(AB)), nAB = neg(AB);
    bool ok = true;
    for (int s = 0; s < 2 * N; s++) {
        BlockWriter w(tab.data() + LY::chunk_off(s, N));
        if (s < N) {  // forward chunk of block l = s: one linear stream per stage
            const int l = s;
            const DM BiT = tr(Bi[l]);
            w.emit(nABt, KS, KX, DENSE);
            w.emit(nAB, KX, KS, DENSE);
            w.emit(BiT, KX, KX, LOWER);
            w.emit(l >= 1 ? neg(mul(BiT, tr(Al[l - 1]))) : Zero, KX, KX, DENSE);
            ok = ok && w.structure_ok && w.cursor == LY::NTF;
        } else {  // backward chunk of block l = 2N-1-s
            const int l = 2 * N - 1 - s;
            w.emit(Bi[l], KX, KX, UPPER);
            w.emit(l < N - 1 ? neg(mul(Bi[l], Al[l])) : Zero, KX, KX, DENSE);
            ok = ok && w.structure_ok && w.cursor == LY::NTB;
        }
    }
    if (!ok) { p.why = "MFMA4R packer: block structure mismatch"; return 0; }
    for (double x : tab)
        if (!std::isfinite(x)) { p.why = "non-finite folded constant (singular Beta block?)"; return 0; }
    p.KX = KX;
    p.KS = KS;
    // LDS: header + four chunk slots + NLDS state vectors per wavefront; the rest of the 2 N KX state vectors in registers
    const int lds_d = 163840 / 8 - LY::HDR_D - 4 * LY::CF - 128, NV = 2 * N * KX;  // y and lambda on the chip
    int max_reg_vecs = 150;  // 300 of the 512 registers for state (what the register allocator places without spilling)
    if (const char *ev = getenv("SPCIES_FR_MAX_REG_VECS")) max_reg_vecs = atoi(ev);
    int want_nw = 0;
    if (const char *ev = getenv("SPCIES_FR_NW")) want_nw = atoi(ev);
    p.NW = 0;
    for (int nw = 4; nw >= 1; nw--) {
        if (want_nw && nw != want_nw) continue;
        const int nl = std::min(NV, lds_d / (nw * 64));
        if (NV - nl <= max_reg_vecs) {
            p.NW = nw;
            p.NLDS = nl;
            break;
        }
    }
    if (!p.NW || (p.NW < 3 && !want_nw)) { p.why = "state does not fit registers + LDS at three wavefronts per CU (use MFMA4G)"; p.NW = 0; return 0; }
    return 1;
}

// build-time instantiations (N, KX, KS, TERMINAL, NW, NLDS; SPCIES_FR_PD = 3): BASELINE configs[2], equMPC-FISTA n = 12, m = 2, N = 30
#define SPCIES_FR_BUILTIN(X) X(30, 3, 4, false, 4, 68)

// (KX, KS) = (ceil(n / 4), ceil((n + m) / 4)): up to 32 rows, up to three more slabs of inputs than of states (the packers are host code;
// the kernels are specialised by name)
#define SPCIES_FR_SHAPES(X) X(1, 1) X(1, 2) X(2, 2) X(1, 3) X(2, 3) X(3, 3) X(1, 4) X(2, 4) X(3, 4) X(4, 4) X(2, 5) X(3, 5) X(4, 5) X(5, 5) X(3, 6) X(4, 6) X(5, 6) X(6, 6) X(4, 7) X(5, 7) X(6, 7) X(7, 7) X(5, 8) X(6, 8) X(7, 8) X(8, 8)

}  // namespace

void plan_free(Plan &p) {
    if (p.d_table) hipFree(p.d_table);
    p.d_table = nullptr;
    if (p.d_scr) hipFree(p.d_scr);
    p.d_scr = nullptr;
    if (p.module) rtc::unload_module((hipModule_t)p.module);
    p.module = nullptr;
    p.ok = false;
}

int plan_build(Plan &p, const Host &h) {
    p.ok = false;
    p.n = h.n; p.m = h.m; p.N = h.N; p.terminal = h.terminal;
    if (h.N < 2) { p.why = "N < 2"; return 0; }
    const int KX = (h.n + 3) / 4, KS = (h.n + h.m + 3) / 4;
    std::vector<double> tab;
    int got = -1;
#define X(KKX, KKS) \
    if (KX == KKX && KS == KKS) got = pack<KKX, KKS>(p, h, tab);
    SPCIES_FR_SHAPES(X)
#undef X
    if (got < 0) { p.why = "MFMA4R: (ceil(n/4), ceil((n+m)/4)) outside the packer's shapes"; return 0; }
    if (got == 0) return 0;
    p.builtin = -1;
    {
        int idx = 0;
#define X(NN, KKX, KKS, TT, WW, LL)                                                                                         \
    if (h.N == NN && KX == KKX && KS == KKS && h.terminal == TT && p.NW == WW && p.NLDS == LL && !getenv("SPCIES_FR_RTC_FLAGS") && \
        !getenv("SPCIES_FR_PD") && h.n % 4 == 0)                                                                             \
        p.builtin = idx;                                                                                                     \
    idx++;
        SPCIES_FR_BUILTIN(X)
#undef X
    }
    p.PD = SPCIES_FR_PD;  // (the build-time kernel's)
    if (p.builtin < 0) {
    // ---- specialise the kernel for this controller (about ten seconds at N = 30)
    char names[2][160];
    std::vector<std::string> nm;
    for (int s = 0; s < 2; s++) {
        snprintf(names[s], sizeof(names[s]), "spcies::fr::fista_r_kernel<%d, %d, %d, %s, %s, %d, %d>", h.N, KX, KS,
                 h.terminal ? "true" : "false", s ? "true" : "false", p.NW, p.NLDS);
        nm.push_back(names[s]);
    }
    int pd = 3;
    if (const char *ev = getenv("SPCIES_FR_PD")) pd = atoi(ev);
    p.PD = std::min(std::max(pd, 1), h.N);
    // the horizon is unrolled by #pragma unroll: lift the size limit under which clang honours the pragma
    std::vector<std::string> extra = {"-DSPCIES_FR_PD=" + std::to_string(p.PD), std::string("-DSPCIES_FR_NFULL=") + (h.n % 4 == 0 ? "1" : "0"),
                                      "-mllvm", "-pragma-unroll-threshold=1000000",
                                      // MFMA results in either register file: without it the y / lambda values that do not fit
                                      // the 256 architectural registers are spilled to scratch memory instead of AGPRs
                                      "-mllvm", "-amdgpu-mfma-vgpr-form"};
    // experiments: SPCIES_FR_RTC_FLAGS holds extra options, blank-separated
    if (const char *ev = getenv("SPCIES_FR_RTC_FLAGS")) {
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
    hipModule_t mod = nullptr;
    hipFunction_t fns[2] = {nullptr, nullptr};
    int rc = rtc::compile_module(kSource, "spcies_fista_r_rtc.hip", nm, extra, &mod, fns);
    if (rc) { p.why = std::string("MFMA4R: run-time specialisation failed: ") + spcies_hip_last_error(); p.build_failed = true; return 0; }
    p.module = mod;
    p.fn[0] = fns[0];
    p.fn[1] = fns[1];
    }
    p.table_bytes = tab.size() * sizeof(double);
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_table, p.table_bytes + 64));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_table, tab.data(), p.table_bytes, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    int dev = 0;
    SPCIES_HIP_CHECK(hipGetDevice(&dev));
    SPCIES_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    p.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // d scratch: one slot per resident wavefront (fista_r_kernel.inc)
    const size_t slot = (size_t)std::max(h.N - p.PD, 1) * KX * 512;
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_scr, slot * p.num_cu * p.NW));
    if (const char *ev = getenv("SPCIES_HIP_POISON"))  // (test runs: see ensure_scratch in spcies_hip.hip)
        if (ev[0] == '1') SPCIES_HIP_CHECK(hipMemset(p.d_scr, 0xFF, slot * p.num_cu * p.NW));
    p.ok = true;
    p.why.clear();
    return 0;
}

int launch(Plan &p, int k_max, double tol, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
           int *k, int *e, double *z, double *lam, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4R variant unavailable: %s", p.why.c_str());
    const bool want_sol = (z || lam);
    if (want_sol && !(z && lam)) return fail(SPCIES_HIP_EINVAL, "MFMA4R variant: pass both of z, lambda or none");
    Args args{p.n, p.m, k_max, ref_stride, tol, B};
    const long n_tiles = (B + 15) / 16, n_groups = (n_tiles + p.NW - 1) / p.NW;
    const long wgs = std::min<long>(n_groups, p.num_cu);
    if (wgs <= 0) return 0;
    const double *table = p.d_table;
    double *dump = p.d_table + p.table_bytes / sizeof(double);
    double *dscr = p.d_scr;
    if (p.builtin >= 0) {
        int idx = 0;
#define X(NN, KKX, KKS, TT, WW, LL)                                                                                              \
    if (p.builtin == idx) {                                                                                                       \
        if (want_sol)                                                                                                             \
            hipLaunchKernelGGL((fista_r_kernel<NN, KKX, KKS, TT, true, WW, LL>), dim3((unsigned)wgs), dim3(WW * 64), 0, st, args, table, x0, xr, \
                               ur, u, k, e, z, lam, dump, dscr);                                                                  \
        else                                                                                                                      \
            hipLaunchKernelGGL((fista_r_kernel<NN, KKX, KKS, TT, false, WW, LL>), dim3((unsigned)wgs), dim3(WW * 64), 0, st, args, table, x0, xr, \
                               ur, u, k, e, z, lam, dump, dscr);                                                                  \
    }                                                                                                                             \
    idx++;
        SPCIES_FR_BUILTIN(X)
#undef X
        SPCIES_HIP_CHECK(hipGetLastError());
        return 0;
    }
    void *params[] = {&args, &table, &x0, &xr, &ur, &u, &k, &e, &z, &lam, &dump, &dscr};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn[want_sol ? 1 : 0], (unsigned)wgs, 1, 1, p.NW * 64, 1, 1, 0, st, params, nullptr));
    return 0;
}

}  // namespace fr
}  // namespace spcies
