// Host side of the MFMA4R variant of MPCT EADMM (eadmm_r.hpp): table packer for the "H" lane layout, kernel specialisation
// (hiprtc / build-time), launch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/spcies_hip.h"
#include "eadmm_r.hpp"
#include "eadmm_r_kernel.inc"
#include "rtc_common.hpp"

namespace spcies {
namespace er {

static const char *const kSource =
#include "eadmm_r_src.inc"
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
        for (int k = 0; k < A.c; k++) {
            const double v = A(i, k);
            if (v == 0.0) continue;
            for (int j = 0; j < B.c; j++) C(i, j) += v * B(k, j);
        }
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
DM scale_cols(DM A, const double *d) {
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < A.c; j++) A(i, j) *= d[j];
    return A;
}
// inverse of the upper-triangular Beta block as the reference stores it (reciprocal diagonal)
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
// Appends the MFMA records of M (KI row slabs x KJ k-slabs) in issue order - J outer, register R inner, the rule of mf_nz - to a
// stream of pair records: MFMA c, element (k, r, i) = M[4 (2 R + r) + i][4 J + k] at 64 (c / 2) + 2 (4 (2 k + r) + i) + c % 2.
// pair = false: single records of 32 doubles (header products).
struct RecordWriter {
    double *base;
    bool pair;
    int cursor = 0;
    bool structure_ok = true;
    RecordWriter(double *b, bool pr) : base(b), pair(pr) {}
    void emit(const DM &M, int KI, int KJ, int pat) {
        auto at = [&](int i, int j) { return (i < M.r && j < M.c) ? M(i, j) : 0.0; };
        // one record: block r = 0 carries M[2 R][J0], block r = 1 carries M[2 R + 1][J1] (J < 0: no block)
        auto put = [&](int R, int J0, int J1, bool issued) {
            const int Jr[2] = {J0, J1};
            if (!issued) {
                for (int r = 0; r < 2; r++)
                    for (int i = 0; i < 4 && Jr[r] >= 0; i++)
                        for (int k = 0; k < 4; k++)
                            if (at(4 * (2 * R + r) + i, 4 * Jr[r] + k) != 0.0) structure_ok = false;
                return;
            }
            double *t = pair ? base + (size_t)(cursor / 2) * 64 + (cursor % 2) : base + (size_t)cursor * 32;
            for (int k = 0; k < 4; k++)
                for (int r = 0; r < 2; r++)
                    for (int i = 0; i < 4; i++) {
                        // a block outside the pattern that shares its MFMA with one inside must be structurally zero too
                        const bool inside = Jr[r] >= 0 && in_blk(2 * R + r, Jr[r], KI, KJ, pat);
                        const double v = Jr[r] >= 0 ? at(4 * (2 * R + r) + i, 4 * Jr[r] + k) : 0.0;
                        if (!inside && v != 0.0) structure_ok = false;
                        t[(pair ? 2 : 1) * (4 * (2 * k + r) + i)] = inside ? v : 0.0;
                    }
            cursor++;
        };
        // pair order (eadmm_r_kernel.inc, mf_count): for P, for R: the diagonal record, the cross record; then an odd last k-slab
        for (int P = 0; P < KJ / 2; P++)
            for (int R = 0; R < (KI + 1) / 2; R++) {
                put(R, 2 * P, 2 * P + 1, nz_diag(R, P, KI, KJ, pat));
                put(R, 2 * P + 1, 2 * P, nz_cross(R, P, KI, KJ, pat));
            }
        if (KJ % 2)
            for (int R = 0; R < (KI + 1) / 2; R++) put(R, KJ - 1, KJ - 1, mf_nz(R, KJ - 1, KI, pat));
    }
};

template <int KX, int KS, bool MIDSAME, bool GEN>
int pack(Plan &p, const Host &h, std::vector<double> &tab) {
    using LY = Layout<KX, KS, MIDSAME, GEN>;
    const int n = h.n, m = h.m, N = h.N, nm = n + m;
    tab.assign((size_t)LY::table_doubles(N), 0.0);
    DM AB(n, nm), W2(nm, nm), TS(nm, nm);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < nm; j++) AB(i, j) = h.AB[(size_t)i * nm + j];
    for (int i = 0; i < nm; i++)
        for (int j = 0; j < nm; j++) W2(i, j) = h.W2[(size_t)i * nm + j];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) TS(i, j) = h.T[(size_t)i * n + j];
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) TS(n + i, n + j) = h.S[(size_t)i * m + j];
    for (int l = 0; l <= N; l++) {
        const double *lb = (l == 0) ? h.LB0 : (l == N ? h.LBs : h.LB), *ub = (l == 0) ? h.UB0 : (l == N ? h.UBs : h.UB);
        for (int j = 0; j < nm; j++) {
            tab[LY::k_off(l, N, LY::K_RHO) + j] = h.rho[(size_t)l * nm + j];  // (MIDSAME: the middle stages write the same table)
            tab[LY::k_off(l, N, LY::K_H1I) + j] = h.H1i[(size_t)l * nm + j];
            tab[LY::k_off(l, N, LY::K_NH3I) + j] = GEN ? 0.0 : -h.H3i[(size_t)l * nm + j];
            tab[LY::k_off(l, N, LY::K_LB) + j] = lb[j];
            tab[LY::k_off(l, N, LY::K_UB) + j] = ub[j];
        }
    }
    for (int j = 0; j < nm; j++) {
        tab[LY::c_off(N, LY::C_RHO0) + j] = h.rho0[j];
        tab[LY::c_off(N, LY::C_RHOS) + j] = h.rhos[j];
    }
    bool ok = true;
    const DM ABt = tr(AB);
    // GEN: G_l = -blkdiag(Q_i, R_i) of stage 0 (Q_mi, R_bi), the middle stages (Q_bi, R_bi), stage N (Q_mi, R_mi) (:321-366)
    auto Gm = [&](int l) {
        DM G(nm, nm);
        if (!GEN) return G;
        const double *Qi = (l == 0 || l == N) ? h.Q_mi : h.Q_bi, *Ri = (l == N) ? h.R_mi : h.R_bi;
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) G(i, j) = -Qi[(size_t)i * n + j];
        for (int i = 0; i < m; i++)
            for (int j = 0; j < m; j++) G(n + i, n + j) = -Ri[(size_t)i * m + j];
        return G;
    };
    {
        RecordWriter w(tab.data() + LY::inv_off(N), true);  // HW: W2 (pair records: the header's products are part of the record stream)
        w.emit(W2, KS, KS, DENSE);
        ok = ok && w.structure_ok && w.cursor == LY::HW_M;
        RecordWriter wb(tab.data() + LY::hb_off(N), true);  // HB: AB' for stage 0, then (general Q, R) G_0
        wb.emit(ABt, KS, KX, DENSE);
        if (GEN) wb.emit(Gm(0), KS, KS, DENSE);
        ok = ok && wb.structure_ok && wb.cursor == LY::HB_M;
        RecordWriter w2(tab.data() + LY::ts_off(N), false);  // setup only: read from L2, not kept in LDS
        w2.emit(TS, KS, KS, DENSE);
        ok = ok && w2.structure_ok && w2.cursor == LY::M_W2;
    }
    std::vector<DM> Bi(N), Al(N - 1);
    for (int l = 0; l < N; l++) Bi[l] = beta_inverse(h.Beta + (size_t)l * n * n, n);
    for (int l = 0; l < N - 1; l++) {
        Al[l] = DM(n, n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Al[l](i, j) = h.Alpha[((size_t)l * n + i) * n + j];
    }
    const DM Zero(n, n);
    for (int s = 0; s < 2 * N; s++) {
        RecordWriter w(tab.data() + LY::chunk_off(s, N), true);
        if (s < N) {  // forward chunk of block l: y_l = F1 q3_{l+1}[x] + F2 q3_l + F3 y_{l-1}
            const int l = s;
            const DM BiT = tr(Bi[l]);
            if (GEN) {  // (:184-217) the reference's folded AB_mi (stage 0) / AB_bi and Q_mi (stage N) / Q_bi
                DM ABh(n, nm), Qh(n, n);
                const double *abh = (l == 0) ? h.AB_mi : h.AB_bi, *qh = (l + 1 == N) ? h.Q_mi : h.Q_bi;
                for (int i = 0; i < n; i++) {
                    for (int j = 0; j < nm; j++) ABh(i, j) = abh[(size_t)i * nm + j];
                    for (int j = 0; j < n; j++) Qh(i, j) = qh[(size_t)i * n + j];
                }
                w.emit(neg(mul(BiT, ABh)), KX, KS, DENSE);
                w.emit(l >= 1 ? neg(mul(BiT, tr(Al[l - 1]))) : Zero, KX, KX, DENSE);
                w.emit(mul(BiT, Qh), KX, KX, DENSE);
            } else {
                w.emit(neg(mul(BiT, scale_cols(AB, h.H3i + (size_t)l * nm))), KX, KS, DENSE);  // F2, F3 first: they do not wait for q3_{l+1}
                w.emit(l >= 1 ? neg(mul(BiT, tr(Al[l - 1]))) : Zero, KX, KX, DENSE);
                w.emit(scale_cols(BiT, h.H3i + (size_t)(l + 1) * nm), KX, KX, LOWER);
            }
            ok = ok && w.structure_ok && w.cursor == LY::MF;
        } else {  // backward chunk of block l = 2N-1-s: mu_l = B1 y_l + B2 mu_{l+1}; AB' for stage l + 1
            const int l = 2 * N - 1 - s;
            w.emit(Bi[l], KX, KX, UPPER);
            w.emit(l < N - 1 ? neg(mul(Bi[l], Al[l])) : Zero, KX, KX, DENSE);
            w.emit(ABt, KS, KX, DENSE);
            if (GEN) w.emit(Gm(l + 1), KS, KS, DENSE);
            ok = ok && w.structure_ok && w.cursor == LY::MB;
        }
    }
    if (!ok) { p.why = "MFMA4R packer: block structure mismatch"; return 0; }
    for (double x : tab)
        if (!std::isfinite(x)) { p.why = "non-finite folded constant (singular Beta block?)"; return 0; }
    p.KX = KX;
    p.KS = KS;
    // LDS: header, four chunk slots, per-wavefront x0 and c2, three y blocks per wavefront on their way back from the scratch slot, and z3 /
    // lambda of the first NLS stages; the other stages' z3 / lambda in registers (2 RS doubles per stage and lane) next to lambda_0,
    // lambda_{N+2}, z2 (old and new), q2, the middle stages' row constants and the last two y blocks
    const int RX = LY::RX, RS = LY::RS;
    int max_reg = 90;  // doubles per lane of z3 / lambda the register allocator places without spilling (measured at configs[3])
    if (const char *ev = getenv("SPCIES_ER_MAX_REG")) max_reg = atoi(ev);
    const long lds_free = 163840 - 8L * (LY::hdr_lds(N) + 4 * LY::CMAX + 2 * RS * 256 + 3 * 4 * RX * 64) - 512;
    if (lds_free < 0) { p.why = "MFMA4R: header and chunk ring exceed the LDS"; return 0; }
    const int nls_max = (int)std::min<long>(N + 1, lds_free / (8L * 2 * RS * 256));
    int nls = std::max(0, ((N + 1) * 2 * RS - max_reg + 2 * RS - 1) / (2 * RS));
    if (const char *ev = getenv("SPCIES_ER_NLS")) nls = atoi(ev);
    // (the LDS is short of the preferred split - general Q, R at configs[3]: a larger header and ring - : up to 104 doubles of state go to
    // the registers; the allocator then spills a few row constants of the set-up, measured harmless: 118.0 ms with 364 B, 118.4 ms with none)
    if (nls > nls_max && ((N + 1) - nls_max) * 2 * RS <= 104 && !getenv("SPCIES_ER_NLS")) nls = nls_max;
    if (nls > nls_max || nls < 0) { p.why = "MFMA4R: the iteration state does not fit registers + LDS (use MFMA4G)"; return 0; }
    p.NLS = nls;
    p.midsame = MIDSAME;
    p.general = GEN;
    p.RX = RX;
    return 1;
}

// build-time instantiations (N, KX, KS, NLS, MIDSAME): BASELINE configs[3], MPCT-EADMM n = 20, m = 2, N = 20
#ifndef SPCIES_ER_BUILTIN
#define SPCIES_ER_BUILTIN(X) X(20, 5, 6, 6, true, false)
#endif

// (KX, KS) = (ceil(n / 4), ceil((n + m) / 4)): up to 32 rows, up to three more slabs of inputs than of states (the packers are host code;
// the kernels are specialised by name)
#define SPCIES_ER_SHAPES(X) X(1, 1) X(1, 2) X(2, 2) X(1, 3) X(2, 3) X(3, 3) X(1, 4) X(2, 4) X(3, 4) X(4, 4) X(2, 5) X(3, 5) X(4, 5) X(5, 5) X(3, 6) X(4, 6) X(5, 6) X(6, 6) X(4, 7) X(5, 7) X(6, 7) X(7, 7) X(5, 8) X(6, 8) X(7, 8) X(8, 8)

}  // namespace

void plan_free(Plan &p) {
    if (p.d_table) hipFree(p.d_table);
    p.d_table = nullptr;
    if (p.d_yscr) hipFree(p.d_yscr);
    p.d_yscr = nullptr;
    if (p.module) rtc::unload_module((hipModule_t)p.module);
    p.module = nullptr;
    p.ok = false;
}

int plan_build(Plan &p, const Host &h) {
    p.ok = false;
    p.n = h.n; p.m = h.m; p.N = h.N;
    if (h.N < 2) { p.why = "N < 2"; return 0; }
    if (const char *ev = getenv("SPCIES_ER_DISABLE"))
        if (ev[0] == '1') { p.why = "disabled (SPCIES_ER_DISABLE=1)"; return 0; }
    const int KX = (h.n + 3) / 4, KS = (h.n + h.m + 3) / 4;
    // one table of row constants for the stages 1 .. N - 1 when they are all the same (one rho, one pair of bounds: the usual case)
    bool midsame = true;
    {
        const int nm = h.n + h.m;
        for (int l = 2; l < h.N && midsame; l++)
            for (int j = 0; j < nm; j++)
                if (h.rho[(size_t)l * nm + j] != h.rho[(size_t)nm + j] || h.H1i[(size_t)l * nm + j] != h.H1i[(size_t)nm + j] ||
                    (!h.general && h.H3i[(size_t)l * nm + j] != h.H3i[(size_t)nm + j]))
                    midsame = false;
        if (getenv("SPCIES_ER_NO_MIDSAME")) midsame = false;
    }
    std::vector<double> tab;
    int got = -1;
#define X(KKX, KKS)                                                                                                              \
    if (KX == KKX && KS == KKS)                                                                                                  \
        got = h.general ? (midsame ? pack<KKX, KKS, true, true>(p, h, tab) : pack<KKX, KKS, false, true>(p, h, tab))             \
                        : (midsame ? pack<KKX, KKS, true, false>(p, h, tab) : pack<KKX, KKS, false, false>(p, h, tab));
    SPCIES_ER_SHAPES(X)
#undef X
    if (got < 0) { p.why = "MFMA4R: (ceil(n/4), ceil((n+m)/4)) outside the packer's shapes"; return 0; }
    if (got == 0) return 0;
    p.builtin = -1;
    {
        int idx = 0;
#define X(NN, KKX, KKS, LL, MM, GG)                                                                                                                  \
    if (h.N == NN && KX == KKX && KS == KKS && p.NLS == LL && p.midsame == MM && p.general == GG && !getenv("SPCIES_ER_RTC_FLAGS")) p.builtin = idx; \
    idx++;
        SPCIES_ER_BUILTIN(X)
#undef X
    }
    if (p.builtin < 0) {
        const char *ev = getenv("SPCIES_HIP_RTC");
        if (ev && ev[0] == '0') { p.why = "shape not instantiated at build time and SPCIES_HIP_RTC=0"; return 0; }
        std::vector<std::string> nm;
        for (int s = 0; s < 2; s++) {
            char name[160];
            snprintf(name, sizeof(name), "spcies::er::eadmm_r_kernel<%d, %d, %d, %s, %d, %s, %s>", h.N, KX, KS, s ? "true" : "false", p.NLS,
                     p.midsame ? "true" : "false", p.general ? "true" : "false");
            nm.push_back(name);
        }
        // the horizon is unrolled by #pragma unroll: lift the size limit under which clang honours the pragma; MFMA results in
        // either register file (the state beyond 256 architectural registers would otherwise be spilled to scratch memory)
        std::vector<std::string> extra = {"-mllvm", "-pragma-unroll-threshold=1000000", "-mllvm", "-amdgpu-mfma-vgpr-form"};
        for (const std::string &e : rtc::split_flags(getenv("SPCIES_ER_RTC_FLAGS"))) extra.push_back(e);
        hipModule_t mod = nullptr;
        hipFunction_t fns[2] = {nullptr, nullptr};
        int rc = rtc::compile_module(kSource, "spcies_eadmm_r_rtc.hip", nm, extra, &mod, fns);
        if (rc) { p.why = std::string("MFMA4R: run-time specialisation failed: ") + spcies_hip_last_error(); p.build_failed = true; return 0; }
        int scratch = 0;
        if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fns[0]) != hipSuccess) scratch = 0;
        if (getenv("SPCIES_ER_VERBOSE"))
            fprintf(stderr, "[spcies eadmm_r] N=%d KX=%d KS=%d NLS=%d midsame=%d general=%d scratch=%d B per lane\n", h.N, KX, KS, p.NLS, (int)p.midsame, (int)p.general, scratch);
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
    // y scratch: one slot per resident wavefront (eadmm_r_kernel.inc; the last kYKeep = 2 blocks stay in registers)
    const size_t slot = (size_t)std::max(h.N - kYKeep, 1) * p.RX * 512;
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_yscr, slot * p.num_cu * 4));
    if (const char *ev = getenv("SPCIES_HIP_POISON"))  // (test runs: see ensure_scratch in spcies_hip.hip)
        if (ev[0] == '1') SPCIES_HIP_CHECK(hipMemset(p.d_yscr, 0xFF, slot * p.num_cu * 4));
    p.ok = true;
    p.why.clear();
    return 0;
}

int launch(Plan &p, int k_max, double tol, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
           int *k, int *e, double *z1, double *z2, double *z3, double *lam, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4R variant unavailable: %s", p.why.c_str());
    const bool want_sol = (z1 || z2 || z3 || lam);
    if (want_sol && !(z1 && z2 && z3 && lam)) return fail(SPCIES_HIP_EINVAL, "MFMA4R variant: pass all of z1, z2, z3, lambda or none");
    Args args{p.n, p.m, k_max, ref_stride, tol, B};
    const long n_groups = (B + 31) / 32;
    const long wgs = std::min<long>(n_groups, p.num_cu);
    if (wgs <= 0) return 0;
    // the reference copies only the first n entries of every multiplier block out, contiguously (code_MPCT_EADMM_C.c:495-513): the
    // rest of the (N + 3)(n + m) record is zero
    if (want_sol) SPCIES_HIP_CHECK(hipMemsetAsync(lam, 0, (size_t)B * (size_t)(p.N + 3) * (p.n + p.m) * sizeof(double), st));
    const double *table = p.d_table;
    double *yscr = p.d_yscr;
    if (p.builtin >= 0) {
        int idx = 0;
#define X(NN, KKX, KKS, LL, MM, GG)                                                                                                      \
    if (p.builtin == idx) {                                                                                                              \
        if (want_sol)                                                                                                                    \
            hipLaunchKernelGGL((eadmm_r_kernel<NN, KKX, KKS, true, LL, MM, GG>), dim3((unsigned)wgs), dim3(256), 0, st, args, table, x0, xr, ur, u, k, e, \
                               z1, z2, z3, lam, yscr);                                                                                   \
        else                                                                                                                             \
            hipLaunchKernelGGL((eadmm_r_kernel<NN, KKX, KKS, false, LL, MM, GG>), dim3((unsigned)wgs), dim3(256), 0, st, args, table, x0, xr, ur, u, k, e, \
                               z1, z2, z3, lam, yscr);                                                                                   \
    }                                                                                                                                    \
    idx++;
        SPCIES_ER_BUILTIN(X)
#undef X
        SPCIES_HIP_CHECK(hipGetLastError());
        return 0;
    }
    void *params[] = {&args, &table, &x0, &xr, &ur, &u, &k, &e, &z1, &z2, &z3, &lam, &yscr};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn[want_sol ? 1 : 0], (unsigned)wgs, 1, 1, 256, 1, 1, 0, st, params, nullptr));
    return 0;
}

}  // namespace er
}  // namespace spcies
