// Host side of the MFMA4R variant of the lax / equ ADMM solvers (admm_r.hpp): table packer, kernel specialisation (hiprtc), launch.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/spcies_hip.h"
#include "admm_r.hpp"
#include "admm_r_kernel.inc"
#include "rtc_common.hpp"

namespace spcies {
namespace ar {

static const char *const kSource =
#include "admm_r_src.inc"
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
DM scale_cols(DM A, const std::vector<double> &d) {
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < A.c; j++) A(i, j) *= d[j];
    return A;
}
DM scale_rows(DM A, const std::vector<double> &d) {
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < A.c; j++) A(i, j) *= d[i];
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
// appends the non-zero 4x4 blocks of M in issue order (J outer, I inner): element-interleaved pairs (the chunk stream) or single blocks of
// 16 doubles (the header)
struct BlockWriter {
    double *base;
    bool pair;
    int cursor = 0;
    bool structure_ok = true;
    BlockWriter(double *b, bool pr) : base(b), pair(pr) {}
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
                double *t = pair ? base + (size_t)(cursor / 2) * 32 + (cursor % 2) : base + (size_t)cursor * 16;
                for (int k = 0; k < 4; k++)
                    for (int i = 0; i < 4; i++) t[(pair ? 2 : 1) * (k * 4 + i)] = at(4 * I + i, 4 * J + k);
                cursor++;
            }
    }
};

template <int KX, int KS, bool TERMINAL, bool GEN>
int pack(Plan &p, const AdmmHost &a, std::vector<double> &tab) {
    using LY = Layout<KX, KS, TERMINAL, GEN>;
    const int n = a.n, m = a.m, N = a.N, nm = n + m;
    tab.assign((size_t)LY::table_doubles(N), 0.0);
    DM AB(n, nm), A(n, n), HiN(n, n), T(n, n);
    for (int i = 0; i < n; i++) {
        for (int j = 0; j < nm; j++) AB(i, j) = a.AB[(size_t)i * nm + j];
        for (int j = 0; j < n; j++) {
            A(i, j) = AB(i, j);
            HiN(i, j) = TERMINAL ? a.Hi_N[(size_t)i * n + j] : 0.0;
            T(i, j) = TERMINAL ? a.T[(size_t)i * n + j] : 0.0;
        }
    }
    // Hd of stage t = 0 .. N - 1 over the (n + m) rows (stage 0: u rows only)
    auto Hd = [&](int t) {
        std::vector<double> d(nm, 0.0);
        if (t == 0)
            for (int j = 0; j < m; j++) d[n + j] = a.Hi_0[j];
        else
            for (int j = 0; j < nm; j++) d[j] = a.Hi[(size_t)(t - 1) * nm + j];
        return d;
    };
    // rho and the bounds of row j of stage t = 0 .. N (stage 0: u rows, stage N: x rows); scalar rho / constant bounds unless `gen`
    // (vector rho, VAR_BOUNDS: code_laxMPC_ADMM_C.c:323-348, 490-568 - rho_0 / rho_v / rho_N and LBu0 / LBz / LBN of AdmmHost)
    auto has = [&](int t, int j) { return t == 0 ? (j >= n && j < nm) : (t == N ? (TERMINAL && j < n) : j < nm); };
    auto rho_of = [&](int t, int j) -> double {
        if (!a.gen) return a.rho;
        return t == 0 ? a.rho_0[j - n] : (t == N ? a.rho_N[j] : a.rho_v[(size_t)(t - 1) * nm + j]);
    };
    auto lb_of = [&](int t, int j) -> double {
        if (!a.gen) return a.LB[j];
        return t == 0 ? a.LBu0[j - n] : (t == N ? a.LBN[j] : a.LBz[(size_t)(t - 1) * nm + j]);
    };
    auto ub_of = [&](int t, int j) -> double {
        if (!a.gen) return a.UB[j];
        return t == 0 ? a.UBu0[j - n] : (t == N ? a.UBN[j] : a.UBz[(size_t)(t - 1) * nm + j]);
    };
    if (!GEN) {  // the middle stages share their row constants (scalar rho, constant bounds: Hi{l} is the same vector for l = 1 .. N - 1)
        for (int t = 2; t < N; t++)
            for (int j = 0; j < nm; j++)
                if (a.Hi[(size_t)(t - 1) * nm + j] != a.Hi[j]) { p.why = "MFMA4R (ADMM): stage-wise Hi"; return 0; }
    }
    // ---- unit-box coordinates (admm_r_kernel.inc, UNIT): D = ub - lb per row and stage when every real row has a finite box that contains
    // 0 (w = 0, the cold start, is then a state of the scaled iteration) and is not wider than 1e5 (absolute rounding of w = D w' stays
    // below 1e-11); D = 1, lb = 0 on rows that do not exist.  SPCIES_AR_UNIT=0 keeps the plain coordinates (not with GEN).
    bool unit = a.rho > 0 || a.gen;
    for (int t = 0; t <= N && unit; t++)
        for (int j = 0; j < nm && unit; j++) {
            if (!has(t, j)) continue;
            const double lo = lb_of(t, j), hi = ub_of(t, j), r = rho_of(t, j);
            if (!(std::isfinite(lo) && std::isfinite(hi)) || !(lo <= 0.0 && hi >= 0.0) || !(hi - lo > 1e-9) || hi - lo > 1e5 || !(r > 0)) unit = false;
        }
    if (const char *ev = getenv("SPCIES_AR_UNIT"))
        if (ev[0] == '0') unit = false;
    if (GEN && !unit) { p.why = "MFMA4R (ADMM): vector rho / stage-wise bounds need a finite box around 0 on every row (unit-box coordinates)"; return 0; }
    p.unit = unit;
    auto Dof = [&](int t) { std::vector<double> v(4 * KS, 1.0); for (int j = 0; j < nm; j++) if (unit && has(t, j)) v[j] = ub_of(t, j) - lb_of(t, j); return v; };
    auto Lof = [&](int t) { std::vector<double> v(4 * KS, 0.0); for (int j = 0; j < nm; j++) if (unit && has(t, j)) v[j] = lb_of(t, j); return v; };
    auto rhoD = [&](int t, int cnt) {  // column scaling of the blocks that multiply s_t (q_hat = rho D s)
        std::vector<double> v(cnt, 1.0), D = Dof(t);
        for (int j = 0; j < cnt; j++) v[j] = (unit && has(t, j)) ? rho_of(t, j) * D[j] : 1.0;
        return v;
    };
    auto invD = [&](int t, int cnt) {  // row scaling of the blocks that produce z_t
        std::vector<double> v(cnt, 1.0), D = Dof(t);
        for (int j = 0; j < cnt; j++) v[j] = unit ? 1.0 / D[j] : 1.0;
        return v;
    };
    auto rc = [&](int which) { return tab.data() + (size_t)which * LY::RC; };
    // the eight row constants of stage t in unit-box coordinates (Layout::G_*), rows that do not exist: IQ = A3 = A1 = A2 = LB = RD = 0, B1 = D = 1
    auto unit_consts = [&](int t, double *iq, double *a3, double *a1, double *b1, double *a2, double *dd, double *lbv, double *rd) {
        const std::vector<double> D = Dof(t), L = Lof(t), hd = t < N ? Hd(t) : std::vector<double>(nm, 0.0);
        for (int j = 0; j < 4 * KS; j++) {
            const bool h = j < nm && has(t, j);
            const double r = h ? rho_of(t, j) : 0.0;
            if (iq) iq[j] = h ? 1.0 / (r * D[j]) : 0.0;
            if (a3) a3[j] = h ? L[j] / D[j] : 0.0;
            if (a1) a1[j] = (h && t < N) ? r * hd[j] : 0.0;
            if (b1) b1[j] = 1.0 - ((h && t < N) ? r * hd[j] : 0.0);
            if (a2) a2[j] = (h && t > 0 && t < N && j < n) ? hd[j] / D[j] : 0.0;
            if (dd) dd[j] = D[j];
            if (lbv) lbv[j] = L[j];
            if (rd) rd[j] = h ? r * D[j] : 0.0;
        }
    };
    if (unit) {
        unit_consts(0, rc(LY::C_IQ0), rc(LY::C_A30), rc(LY::C_A10), rc(LY::C_B10), nullptr, rc(LY::C_D0), nullptr, rc(LY::C_RD0));
        if (!GEN) unit_consts(1, rc(LY::C_IQM), rc(LY::C_A3M), rc(LY::C_A1M), rc(LY::C_B1M), rc(LY::C_A2M), rc(LY::C_DM), nullptr, rc(LY::C_RDM));
        if (TERMINAL) unit_consts(N, rc(LY::C_IQT), rc(LY::C_A3T), nullptr, nullptr, nullptr, rc(LY::C_DT), nullptr, rc(LY::C_RDT));
        else for (int j = 0; j < 4 * KS; j++) rc(LY::C_DT)[j] = 1.0;
    }
    for (int j = 0; j < m; j++) {
        rc(LY::C_HD0)[n + j] = a.Hi_0[j];
        rc(LY::C_LB0)[n + j] = lb_of(0, n + j);
        rc(LY::C_UB0)[n + j] = ub_of(0, n + j);
        rc(LY::C_QR)[n + j] = a.R[j];
    }
    for (int j = 0; j < n; j++) rc(LY::C_QR)[j] = a.Q[j];
    for (int j = 0; j < nm; j++) {
        rc(LY::C_HDM)[j] = a.Hi[j];
        rc(LY::C_LBM)[j] = lb_of(1, j);
        rc(LY::C_UBM)[j] = ub_of(1, j);
    }
    if (TERMINAL)
        for (int j = 0; j < n; j++) {
            rc(LY::C_LBT)[j] = lb_of(N, j);
            rc(LY::C_UBT)[j] = ub_of(N, j);
        }
    std::vector<DM> Bi(N), Al(N - 1);
    for (int l = 0; l < N; l++) Bi[l] = beta_inverse(a.Beta.data() + (size_t)l * n * n, n);
    for (int l = 0; l < N - 1; l++) {
        Al[l] = DM(n, n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Al[l](i, j) = a.Alpha[((size_t)l * n + i) * n + j];
    }
    const DM ABt = tr(AB), Zero(n, n), ZeroZ(nm, n);
    bool ok = true;
    {
        BlockWriter w(tab.data() + LY::Z0_OFF, false);
        w.emit(scale_rows(neg(scale_rows(ABt, Hd(0))), invD(0, nm)), KS, KX, DENSE);  // Z_0 = -Hd_0 AB' (unit-box: rows / D_0)
        ok = ok && w.structure_ok && w.cursor == KS * KX;
        BlockWriter w0(tab.data() + LY::S_C0, false), wt(tab.data() + LY::S_T, false), wn(tab.data() + LY::S_CN, false);
        w0.emit(mul(tr(Bi[0]), A), KX, KX, DENSE);  // c_0 = Bi_0' A x0
        wt.emit(T, KX, KX, DENSE);
        wn.emit(neg(tr(Bi[N - 1])), KX, KX, DENSE);  // c_{N-1} = -Bi_{N-1}' xr (equMPC)
        ok = ok && w0.structure_ok && wt.structure_ok && wn.structure_ok;
    }
    for (int s = 0; s < 2 * N; s++) {
        double *chunk = tab.data() + LY::chunk_off(s, N);
        BlockWriter w(chunk, true);
        if (s < N) {  // forward chunk of block l: F2, F3, F1
            const int l = s;
            const DM BiT = tr(Bi[l]);
            w.emit(scale_cols(neg(mul(BiT, scale_cols(AB, Hd(l)))), rhoD(l, nm)), KX, KS, DENSE);  // F2 (unit-box: columns x rho D_l)
            w.emit(l >= 1 ? neg(mul(BiT, tr(Al[l - 1]))) : Zero, KX, KX, DENSE);
            if (l + 1 < N) {
                std::vector<double> dx = Hd(l + 1);
                dx.resize(n);
                w.emit(scale_cols(scale_cols(BiT, dx), rhoD(l + 1, n)), KX, KX, LOWER);  // F1 (unit-box: columns x rho D_{l+1})
            } else if (TERMINAL) {
                w.emit(scale_cols(mul(BiT, HiN), rhoD(N, n)), KX, KX, DENSE);
            } else {
                w.emit(Zero, KX, KX, LOWER);
            }
            ok = ok && w.structure_ok && w.cursor == LY::ntf(l, N);
            if (GEN && l + 1 < N) unit_consts(l + 1, chunk + LY::GF_OFF, chunk + LY::GF_OFF + LY::RC, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        } else {  // backward chunk of block l = 2N-1-s: B1, B2 (block N - 1: Hi_N), Z_{l+1}
            const int l = 2 * N - 1 - s;
            w.emit(Bi[l], KX, KX, UPPER);
            w.emit(l < N - 1 ? neg(mul(Bi[l], Al[l])) : (TERMINAL ? scale_rows(HiN, invD(N, n)) : Zero), KX, KX, DENSE);  // (unit-box: Hi_N rows / D_N)
            w.emit(l < N - 1 ? scale_rows(neg(scale_rows(ABt, Hd(l + 1))), invD(l + 1, nm)) : ZeroZ, KS, KX, DENSE);       // Z_{l+1} (unit-box: rows / D_{l+1})
            ok = ok && w.structure_ok && w.cursor == LY::NTB;
            if (GEN && l + 1 < N) {
                double *g = chunk + LY::GB_OFF;
                unit_consts(l + 1, g + LY::G_IQ * LY::RC, g + LY::G_A3 * LY::RC, g + LY::G_A1 * LY::RC, g + LY::G_B1 * LY::RC, g + LY::G_A2 * LY::RC,
                            g + LY::G_D * LY::RC, g + LY::G_LB * LY::RC, g + LY::G_RD * LY::RC);
            }
        }
    }
    if (!ok) { p.why = "MFMA4R (ADMM) packer: block structure mismatch"; return 0; }
    for (double x : tab)
        if (!std::isfinite(x)) { p.why = "non-finite folded constant (singular Beta block?)"; return 0; }
    p.KX = KX;
    p.KS = KS;
    // LDS: header + four chunk slots + NLDS state vectors per wavefront; the rest of the (N + 1) KS state vectors in registers
    const int lds_d = 163840 / 8 - LY::HDR_LDS - 4 * LY::CMAX - 128, NV = (N + 1) * KS;
    int max_reg_vecs = 110;  // doubles per lane of state the register allocator places next to the kernel's working set
    if (const char *ev = getenv("SPCIES_AR_MAX_REG_VECS")) max_reg_vecs = atoi(ev);
    int want_nw = 0;
    if (const char *ev = getenv("SPCIES_AR_NW")) want_nw = atoi(ev);
    p.NW = 0;
    for (int nw = 4; nw >= 1; nw--) {
        if (want_nw && nw != want_nw) continue;
        const int nl = std::max(0, std::min(NV, lds_d / (nw * 64)));
        if (NV - nl <= max_reg_vecs) {
            p.NW = nw;
            p.NLDS = nl;
            break;
        }
    }
    if (const char *ev = getenv("SPCIES_AR_NLDS")) p.NLDS = std::min(NV, atoi(ev));
    if (!p.NW || (p.NW < 3 && !want_nw)) { p.why = "state does not fit registers + LDS at three wavefronts per CU (use MFMA4G)"; p.NW = 0; return 0; }
    return 1;
}

// (KX, KS) = (ceil(n / 4), ceil((n + m) / 4)): up to 32 rows, up to three more slabs of inputs than of states (the packers are host code;
// the kernels are specialised by name)
#define SPCIES_AR_SHAPES(X) X(1, 1) X(1, 2) X(2, 2) X(1, 3) X(2, 3) X(3, 3) X(1, 4) X(2, 4) X(3, 4) X(4, 4) X(2, 5) X(3, 5) X(4, 5) X(5, 5) X(3, 6) X(4, 6) X(5, 6) X(6, 6) X(4, 7) X(5, 7) X(6, 7) X(7, 7) X(5, 8) X(6, 8) X(7, 8) X(8, 8)

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

int plan_build(Plan &p, const AdmmHost &a) {
    p.ok = false;
    p.n = a.n; p.m = a.m; p.N = a.N; p.terminal = a.terminal; p.rho = a.rho;
    if (a.N < 2) { p.why = "N < 2"; return 0; }
    if (a.ellip) { p.why = "ellipMPC: the block program (BSP) carries it"; return 0; }
    p.gen = a.gen;  // vector rho / stage-wise bounds: the middle stages' row constants ride in the chunk stream (unit-box coordinates only)
    if (const char *ev = getenv("SPCIES_AR_DISABLE"))
        if (ev[0] == '1') { p.why = "disabled (SPCIES_AR_DISABLE=1)"; return 0; }
    {
        const char *ev = getenv("SPCIES_HIP_RTC");
        if (ev && ev[0] == '0') { p.why = "run-time specialisation switched off (SPCIES_HIP_RTC=0)"; return 0; }
    }
    const int KX = (a.n + 3) / 4, KS = (a.n + a.m + 3) / 4;
    std::vector<double> tab;
    int got = -1;
#define X(KKX, KKS)                                                                                                          \
    if (KX == KKX && KS == KKS)                                                                                              \
        got = a.gen ? (a.terminal ? pack<KKX, KKS, true, true>(p, a, tab) : pack<KKX, KKS, false, true>(p, a, tab))          \
                    : (a.terminal ? pack<KKX, KKS, true, false>(p, a, tab) : pack<KKX, KKS, false, false>(p, a, tab));
    SPCIES_AR_SHAPES(X)
#undef X
    if (got < 0) { p.why = "MFMA4R (ADMM): (ceil(n/4), ceil((n+m)/4)) outside the packer's shapes"; return 0; }
    if (got == 0) return 0;
    int pd = 3;
    if (const char *ev = getenv("SPCIES_AR_PD")) pd = atoi(ev);
    p.PD = std::min(std::max(pd, 1), a.N);
    // The horizon is unrolled by #pragma unroll: lift the size limit under which clang honours the pragma.  NOT -amdgpu-mfma-vgpr-form
    // (fista_r / eadmm_r use it): ROCm 7.2's "Rewrite AGPR-Copy-MFMA" pass crashes on these kernels when the allocator spills (the record
    // kernel at n = 12, N = 30; the solve kernel of equMPC at n = 20) - inside hiprtc, i.e. inside the caller's process - and the kernels
    // run as fast without it (20.9 / 31.3 ms against 20.1 / 31.8 at the two benchmark shapes).
    // The middle stages' row constants: in registers up to KS = 4 (n + m <= 16), read from LDS beyond - at KS = 6 the 18 registers are
    // worth more than the exposed reads (C4 shape: 43.2 -> 31.8 ms, scratch 1 212 -> 348 B; n = 12, N = 30 the other way: 20.1 / 24.5 ms)
    std::vector<std::string> nm;
    for (int s = 0; s < 2; s++) {
        char name[160];
        snprintf(name, sizeof(name), "spcies::ar::admm_r_kernel<%d, %d, %d, %s, %s, %d, %d, %s, %s>", a.N, KX, KS, a.terminal ? "true" : "false",
                 s ? "true" : "false", p.NW, p.NLDS, p.unit ? "true" : "false", p.gen ? "true" : "false");
        nm.push_back(name);
    }
    std::vector<std::string> extra = {"-DSPCIES_AR_PD=" + std::to_string(p.PD), std::string("-DSPCIES_AR_KREG=") + (KS <= 4 ? "1" : "0"),
                                      "-mllvm", "-pragma-unroll-threshold=1000000"};
    for (const std::string &e : rtc::split_flags(getenv("SPCIES_AR_RTC_FLAGS"))) extra.push_back(e);
    hipModule_t mod = nullptr;
    hipFunction_t fns[2] = {nullptr, nullptr};
    int rc = -1;
    // Accumulators in VGPR form (762 -> 198 accumulator moves per iteration at n = 12, N = 30: 18.0 -> 17.3 ms) where the compiler takes it: ROCm
    // 7.2's "Rewrite AGPR-Copy-MFMA" pass crashes on some of these kernels when the allocator spills (n = 20 laxMPC) - survivable since the compiler runs in
    // a process of its own (rtc_helper.cpp), and remembered per machine (<digest>.crashed) - so: unit-box kernels up to 16 rows, only with the helper,
    // the plain options on any failure.  SPCIES_AR_VGPR_FORM=0: never; =1: for every shape.
    {
        const char *ev = getenv("SPCIES_AR_VGPR_FORM");
        const bool want = ev ? ev[0] == '1' : (p.unit && KS <= 4);
        bool have_helper = false;
        if (want) {
            std::lock_guard<std::mutex> lk(rtc::rtc_mutex());
            have_helper = rtc::helper_proc().find();
        }
        if (want && have_helper) {
            std::vector<std::string> extra_vf = extra;
            extra_vf.push_back("-mllvm");
            extra_vf.push_back("-amdgpu-mfma-vgpr-form");
            rc = rtc::compile_module(kSource, "spcies_admm_r_rtc.hip", nm, extra_vf, &mod, fns);
            if (rc) mod = nullptr;
        }
    }
    if (rc) rc = rtc::compile_module(kSource, "spcies_admm_r_rtc.hip", nm, extra, &mod, fns);
    if (rc) { p.why = std::string("MFMA4R (ADMM): run-time specialisation failed: ") + spcies_hip_last_error(); p.build_failed = true; return 0; }
    if (getenv("SPCIES_AR_VERBOSE")) {
        int scratch = 0;
        if (hipFuncGetAttribute(&scratch, HIP_FUNC_ATTRIBUTE_LOCAL_SIZE_BYTES, fns[0]) != hipSuccess) scratch = 0;
        fprintf(stderr, "[spcies admm_r] N=%d KX=%d KS=%d NW=%d NLDS=%d of %d unit=%d scratch=%d B per lane\n", a.N, KX, KS, p.NW, p.NLDS, (a.N + 1) * KS, (int)p.unit, scratch);
    }
    p.module = mod;
    p.fn[0] = fns[0];
    p.fn[1] = fns[1];
    p.table_bytes = tab.size() * sizeof(double);
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_table, p.table_bytes + 64));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_table, tab.data(), p.table_bytes, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    int dev = 0;
    SPCIES_HIP_CHECK(hipGetDevice(&dev));
    SPCIES_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    p.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const size_t slot = (size_t)std::max(a.N - p.PD, 1) * KX * 512;
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_scr, slot * p.num_cu * p.NW));
    if (const char *ev = getenv("SPCIES_HIP_POISON"))  // (test runs: see ensure_scratch in spcies_hip.hip)
        if (ev[0] == '1') SPCIES_HIP_CHECK(hipMemset(p.d_scr, 0xFF, slot * p.num_cu * p.NW));
    p.ok = true;
    p.why.clear();
    return 0;
}

int launch(Plan &p, int k_max, double tol, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
           int *k, int *e, double *z, double *v, double *lam, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4R variant unavailable: %s", p.why.c_str());
    const bool want_sol = (z || v || lam);
    if (want_sol && !(z && v && lam)) return fail(SPCIES_HIP_EINVAL, "MFMA4R variant: pass all of z, v, lambda or none");
    Args args{p.n, p.m, k_max, ref_stride, tol, p.rho, B};
    const long n_tiles = (B + 15) / 16, n_groups = (n_tiles + p.NW - 1) / p.NW;
    const long wgs = std::min<long>(n_groups, p.num_cu);
    if (wgs <= 0) return 0;
    const double *table = p.d_table;
    double *dump = p.d_table + p.table_bytes / sizeof(double);
    double *yscr = p.d_scr;
    void *params[] = {&args, &table, &x0, &xr, &ur, &u, &k, &e, &z, &v, &lam, &dump, &yscr};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel((hipFunction_t)p.fn[want_sol ? 1 : 0], (unsigned)wgs, 1, 1, p.NW * 64, 1, 1, 0, st, params, nullptr));
    return 0;
}

}  // namespace ar
}  // namespace spcies
