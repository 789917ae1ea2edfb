// Variant MFMA4G of the banded-Cholesky ADMM solver (laxMPC / equMPC), see mfma4g.hpp for the design: the
// solver of admm_mfma4.hpp for the shapes that kernel is not instantiated for or cannot hold in registers -
// any horizon N (a run-time value), n + m up to 24.  Algorithm: code_laxMPC_ADMM_C.c:308-633.
//
// Stages t = 0..N, z_t = (x_t; u_t) (x_0 rows are structurally zero, stage N has no u rows), blocks l = 0..N-1:
//   q_hat_t = q_t + rho (w_t - 2 clamp(w_t)),  w_t = z_t + lambda_t / rho  (the one state vector per stage, see
//             admm_mfma.hpp: v = clamp(w), lambda = rho (w - v));  iteration 1 uses q_hat = q (v = lambda = 0)
//   r_l  = Dx_{l+1} q_hat_{l+1}[x] - AB (Hd_l o q_hat_l)  (- b for l = 0;  - xr for l = N-1 without terminal block)
//   y_l  = Bi_l' r_l - Bi_l' Alpha_{l-1}' y_{l-1};      mu_l = Bi_l y_l - Bi_l Alpha_l mu_{l+1}
//   z_t  = -Hd_t o (q_hat_t - [mu_{t-1}; 0] + AB' mu_t)  (t < N),   z_N = -Hi_N (q_hat_N - mu_{N-1})
//   w_t <- z_t + (w_t - clamp(w_t))
// State in HBM per 16 instances: w ((N+1) KS slab vectors), the forward-substituted y (N KX); traffic per
// iteration and stage (3 KS + 2 KX) x 512 B.  The per-stage chunks are FISTA's (Bi', Bi' Alpha' | Bi, Bi Alpha
// and hd, lb, ub of stage l + 1).
#pragma once
#include "mfma4g.hpp"

namespace spcies {
namespace g4 {

#pragma clang fp contract(fast)

template <int KX, int KS>
struct AdmmGLayout {
    static constexpr int RC = 4 * KS;
    // stage-invariant blocks: -AB (KX x KS), AB' (KS x KX), Hi_N (KX x KX), T (KX x KX, negated weight)
    static constexpr int T_NAB = 0, T_ABT = KX * KS, T_HIN = 2 * KX * KS, T_T = 2 * KX * KS + KX * KX;
    static constexpr int INV_TILES = (2 * KX * KS + 2 * KX * KX + 1) / 2 * 2;
    enum { C_HD0, C_LB0, C_UB0, C_QR, C_RHO0, C_COUNT };  // Hd of stage 0 (u rows only), its bounds, [Q; R] (negated), rho
    static constexpr int INV_D = INV_TILES * 16 + C_COUNT * RC;
    static constexpr int NT = blk_count(KX, KX, LOWER) + KX * KX, NT_PAD = (NT + 1) / 2 * 2;
    enum { K_HD, K_LB, K_UB, K_RHO, K_COUNT };  // of stage l + 1 (both sweeps)
    static constexpr int CHD = NT_PAD * 16 + K_COUNT * RC;
    static constexpr int LDS_D = INV_D + 2 * CHD;
    static size_t chunks_end(int N) { return (size_t)INV_D + (size_t)2 * N * CHD; }
    // after the chunks: stage-wise lb, ub, rho ([N+1][n+m] each) for the record kernel
    static size_t table_doubles(int N, int nm) { return chunks_end(N) + (size_t)3 * (N + 1) * nm; }
};

// bounds and penalty of every stage t = 0..N and row, from the scalar / constant form or the stage-wise one
// (vector rho, VAR_BOUNDS); rows that do not exist (x of stage 0, u of stage N) get lb = ub = 0, rho = 1
inline void admm_stage_arrays(const AdmmHost &a, std::vector<double> &lb, std::vector<double> &ub, std::vector<double> &rho) {
    const int n = a.n, m = a.m, N = a.N, nm = n + m;
    lb.assign((size_t)(N + 1) * nm, 0.0);
    ub.assign((size_t)(N + 1) * nm, 0.0);
    rho.assign((size_t)(N + 1) * nm, 1.0);
    for (int t = 0; t <= N; t++)
        for (int j = 0; j < nm; j++) {
            const bool exists = (t == 0) ? (j >= n) : (t == N ? (a.terminal && j < n) : true);
            if (!exists) continue;
            double l, u, r;
            if (a.gen) {
                if (t == 0) { l = a.LBu0[j - n]; u = a.UBu0[j - n]; r = a.rho_0[j - n]; }
                else if (t == N) { l = a.LBN[j]; u = a.UBN[j]; r = a.rho_N[j]; }
                else { l = a.LBz[(size_t)(t - 1) * nm + j]; u = a.UBz[(size_t)(t - 1) * nm + j]; r = a.rho_v[(size_t)(t - 1) * nm + j]; }
            } else {
                l = a.LB[j]; u = a.UB[j]; r = a.rho;
            }
            lb[(size_t)t * nm + j] = l; ub[(size_t)t * nm + j] = u; rho[(size_t)t * nm + j] = r;
        }
}

template <int KX, int KS>
inline int admm_plan_build_shape(Plan &p, const AdmmHost &a) {
    using LY = AdmmGLayout<KX, KS>;
    const int n = a.n, m = a.m, N = a.N, nm = n + m;
    std::vector<double> tab(LY::table_doubles(N, nm), 0.0);
    std::vector<double> lbS, ubS, rhoS;
    admm_stage_arrays(a, lbS, ubS, rhoS);
    std::copy(lbS.begin(), lbS.end(), tab.begin() + LY::chunks_end(N));
    std::copy(ubS.begin(), ubS.end(), tab.begin() + LY::chunks_end(N) + lbS.size());
    std::copy(rhoS.begin(), rhoS.end(), tab.begin() + LY::chunks_end(N) + 2 * lbS.size());
    DM AB(n, nm), HiN(n, n), T(n, n);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < nm; j++) AB(i, j) = a.AB[(size_t)i * nm + j];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            HiN(i, j) = a.terminal ? a.Hi_N[(size_t)i * n + j] : 0.0;
            T(i, j) = a.terminal ? a.T[(size_t)i * n + j] : 0.0;
        }
    bool ok = true;
    {
        BlockWriter w(tab, 0);
        w.emit(neg(AB), KX, KS, DENSE);
        w.emit(tr(AB), KS, KX, DENSE);
        w.emit(HiN, KX, KX, DENSE);
        w.emit(T, KX, KX, DENSE);
        ok = ok && w.structure_ok && w.cursor == 2 * KX * KS + 2 * KX * KX;
        double *rc = tab.data() + LY::INV_TILES * 16;
        for (int j = 0; j < m; j++) {
            rc[LY::C_HD0 * LY::RC + n + j] = a.Hi_0[j];
            rc[LY::C_QR * LY::RC + n + j] = a.R[j];
        }
        for (int j = 0; j < nm; j++) {
            rc[LY::C_LB0 * LY::RC + j] = lbS[j];
            rc[LY::C_UB0 * LY::RC + j] = ubS[j];
            rc[LY::C_RHO0 * LY::RC + j] = rhoS[j];
        }
        for (int j = 0; j < n; j++) rc[LY::C_QR * LY::RC + j] = a.Q[j];
    }
    std::vector<DM> Bi(N), Al(N - 1);
    for (int l = 0; l < N; l++) Bi[l] = beta_inverse(a.Beta.data() + (size_t)l * n * n, n);
    for (int l = 0; l < N - 1; l++) {
        Al[l] = DM(n, n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Al[l](i, j) = a.Alpha[((size_t)l * n + i) * n + j];
    }
    const DM Zero(n, n);
    for (int s = 0; s < 2 * N; s++) {
        const size_t base = (size_t)LY::INV_D + (size_t)s * LY::CHD;
        BlockWriter w(tab, base);
        const int l = (s < N) ? s : 2 * N - 1 - s;
        if (s < N) {
            const DM BiT = tr(Bi[l]);
            w.emit(BiT, KX, KX, LOWER);
            w.emit(l >= 1 ? neg(mul(BiT, tr(Al[l - 1]))) : Zero, KX, KX, DENSE);
        } else {
            w.emit(Bi[l], KX, KX, UPPER);
            w.emit(l < N - 1 ? neg(mul(Bi[l], Al[l])) : Zero, KX, KX, DENSE);
        }
        ok = ok && w.structure_ok && w.cursor == LY::NT;
        double *rc = tab.data() + base + LY::NT_PAD * 16;
        const int t = l + 1;  // constants of stage l + 1: Hd (diagonal; stage N: x rows only, the dense Hi_N is a block)
        for (int j = 0; j < nm; j++) {
            if (t < N) rc[LY::K_HD * LY::RC + j] = a.Hi[(size_t)(t - 1) * nm + j];
            rc[LY::K_LB * LY::RC + j] = lbS[(size_t)t * nm + j];
            rc[LY::K_UB * LY::RC + j] = ubS[(size_t)t * nm + j];
            rc[LY::K_RHO * LY::RC + j] = rhoS[(size_t)t * nm + j];
        }
    }
    if (!ok) { p.why = "MFMA4G packer: block structure mismatch"; return 0; }
    p.KX = KX;
    p.KS = KS;
    return plan_upload(p, tab);
}

struct AdmmGArgs {
    Args a;
    double rho;
};

// -------------------------------------------------------------------------------------------------
template <int KX, int KS, bool TERMINAL, bool WANT_SOL, int WG_PER_CU>
__global__ __launch_bounds__(256, WG_PER_CU) void admm_g_kernel(AdmmGArgs pa, const double *__restrict__ tab,
                                                                const double *__restrict__ x0g,
                                                                const double *__restrict__ xrg,
                                                                const double *__restrict__ urg, double *__restrict__ Wg,
                                                                double *__restrict__ Yg, double *__restrict__ u_out,
                                                                int *__restrict__ k_out, int *__restrict__ e_out,
                                                                double *__restrict__ z_out) {
    using LY = AdmmGLayout<KX, KS>;
    const Args &p = pa.a;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = p.n, m = p.m, nm = n + m, N = p.N;
    for (int i = threadIdx.x; i < LY::INV_D / 2; i += 256)
        reinterpret_cast<double2 *>(lds)[i] = reinterpret_cast<const double2 *>(tab)[i];
    double *ring = lds + LY::INV_D;
    const double *seq = tab + LY::INV_D;
    const double *inv_rc = lds + LY::INV_TILES * 16;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int g = lane >> 4, c = lane & 15;
    const int ao = g * 4 + (lane & 3);
    const long n_tiles = (p.B + 15) / 16, n_groups = (n_tiles + 3) / 4;
    const double tol = p.tol;
    const int dim = TERMINAL ? N * nm : N * nm - n;
    Stager<LY::CHD> stg;
#define SPCIES_RC(K, which, s) (K)[(which) * LY::RC + 4 * (s) + g]

    for (long group = blockIdx.x; group < n_groups; group += gridDim.x) {
        const long tile = group * 4 + wave;
        const long inst = tile * 16 + c;
        const bool valid = inst < p.B;
        const SlabBuf Wt(Wg + tile * (long)(N + 1) * KS * 64, (long)(N + 1) * KS), Yt(Yg + tile * (long)N * KX * 64, (long)N * KX);
        const int voff = lane * 8;
        // ---- per-instance setup (code_laxMPC_ADMM_C.c:282-299): q, qT = T xr, b = -A x0
        double qm[KS], qT[KX], xrv[KX], bvec[KX];
        {
            double x0v[KS], xrs[KX];
            const double *xrp = p.ref_stride ? xrg + inst * n : xrg;
            const double *urp = p.ref_stride ? urg + inst * m : urg;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int row = 4 * s + g;
                double xu = 0.0, xr_ = 0.0;
                x0v[s] = 0.0;
                if (valid && row < n) {
                    x0v[s] = x0g[inst * n + row];
                    xr_ = xrp[row];
                    xu = xr_;
                } else if (valid && row < nm) {
                    xu = urp[row - n];
                }
                if (s < KX) {
                    xrv[s < KX ? s : 0] = xr_;
                    xrs[s < KX ? s : 0] = xr_;
                }
                qm[s] = SPCIES_RC(inv_rc, LY::C_QR, s) * xu;
            }
            __syncthreads();  // invariant region visible (first group) / previous group done with the ring
#pragma unroll
            for (int s = 0; s < KX; s++) {
                bvec[s] = 0.0;
                qT[s] = 0.0;
            }
            int tix = LY::T_NAB;
            double2 cur;
            prod<KX, KS, DENSE>(bvec, x0v, lds, ao, tix, cur);  // -A x0 (= the reference's b)
            if constexpr (TERMINAL) {
                tix = LY::T_T;
                prod<KX, KX, DENSE>(qT, xrs, lds, ao, tix, cur);
            }
        }
        stg.issue(seq);
        stg.commit(ring);
        __syncthreads();
        int slot = 0;
        int ao_l = ao;
        bool active = valid;
        int kk = 0;
        // q_hat of a stage from its w; cw = clamp(w)
        auto qhat = [&](const double (&w)[KS], const double *K, int which_lb, int which_ub, int which_rho, bool lastt, double fz,
                        double (&cw)[KS], double (&qh)[KS]) {
#pragma unroll
            for (int s = 0; s < KS; s++) {
                cw[s] = fmin(fmax(w[s], SPCIES_RC(K, which_lb, s)), SPCIES_RC(K, which_ub, s));
                const double q = lastt ? ((s < KX) ? qT[s < KX ? s : 0] : 0.0) : qm[s];
                qh[s] = q + fz * SPCIES_RC(K, which_rho, s) * (w[s] - 2.0 * cw[s]);
            }
        };
        while (true) {
            kk += 1;
            const double fz = (kk == 1) ? 0.0 : 1.0;  // cold start: v = lambda = 0 in iteration 1
            // ======================= forward sweep =======================
            double wc[KS], cw[KS], qc[KS], yprev[KX];
#pragma unroll
            for (int s = 0; s < KS; s++) wc[s] = Wt.ld(s, voff);
#pragma unroll
            for (int s = 0; s < KX; s++) yprev[s] = 0.0;
            qhat(wc, inv_rc, LY::C_LB0, LY::C_UB0, LY::C_RHO0, false, fz, cw, qc);  // stage 0
            double hdc[KS];  // Hd of the current stage
#pragma unroll
            for (int s = 0; s < KS; s++) hdc[s] = SPCIES_RC(inv_rc, LY::C_HD0, s);
            double wpre[KS];  // w_{l+1}, in flight
#pragma unroll
            for (int s = 0; s < KS; s++) wpre[s] = Wt.ld(KS + s, voff);
            for (int l = 0; l < N; l++) {
                asm volatile("" : "+v"(ao_l));
                stg.issue(seq + (long)(l + 1) * LY::CHD);
                const double *ch = ring + slot * LY::CHD;
                const double *K = ch + LY::NT_PAD * 16;
                const bool lastt = (l + 1 == N);
                double wn[KS], qn[KS], cwn[KS];
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    wn[s] = wpre[s];
                    wpre[s] = (l + 2 <= N) ? Wt.ld((l + 2) * KS + s, voff) : 0.0;
                }
                qhat(wn, K, LY::K_LB, LY::K_UB, LY::K_RHO, lastt, fz, cwn, qn);
                // right-hand side (:355-381)
                double r[KX], t1[KS];
#pragma unroll
                for (int s = 0; s < KS; s++) t1[s] = hdc[s] * qc[s];
#pragma unroll
                for (int s = 0; s < KX; s++) r[s] = 0.0;
                if (lastt) {
                    if constexpr (TERMINAL) {
                        double qx[KX];
#pragma unroll
                        for (int s = 0; s < KX; s++) qx[s] = qn[s];
                        int tix = LY::T_HIN;
                        double2 cur;
                        prod<KX, KX, DENSE>(r, qx, lds, ao_l, tix, cur);
                    } else {
#pragma unroll
                        for (int s = 0; s < KX; s++) r[s] = -xrv[s];
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < KX; s++) r[s] = (4 * s + g < n) ? SPCIES_RC(K, LY::K_HD, s) * qn[s] : 0.0;
                }
                if (l == 0) {
#pragma unroll
                    for (int s = 0; s < KX; s++) r[s] -= bvec[s];  // bvec = -A x0 is the reference's b; the rhs is ... - b
                }
                {
                    int tix = LY::T_NAB;
                    double2 cur;
                    prod<KX, KS, DENSE>(r, t1, lds, ao_l, tix, cur);
                }
                // forward substitution (:388-417)
                double y[KX];
#pragma unroll
                for (int s = 0; s < KX; s++) y[s] = 0.0;
                {
                    int tix = 0;
                    double2 cur;
                    prod<KX, KX, LOWER>(y, r, ch, ao_l, tix, cur);
                    prod<KX, KX, DENSE>(y, yprev, ch, ao_l, tix, cur);
                }
#pragma unroll
                for (int s = 0; s < KX; s++) {
                    Yt.st(l * KX + s, voff, y[s]);
                    yprev[s] = y[s];
                }
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    qc[s] = qn[s];
                    hdc[s] = SPCIES_RC(K, LY::K_HD, s);
                }
                stg.commit(ring + (slot ^ 1) * LY::CHD);
                __syncthreads();
                slot ^= 1;
            }
            // ======================= backward sweep: mu, z, w, residuals (:422-620) =======================
            bool res = false;
            double mun[KX], u_keep[KS];
#pragma unroll
            for (int s = 0; s < KX; s++) mun[s] = 0.0;
            // z of one stage -> w, residual; returns clamp(w_new) (= v) in vn
            auto finish = [&](int t, const double (&w)[KS], const double (&z)[KS], const double (&cwo)[KS], const double *K,
                              int which_lb, int which_ub, double (&vn)[KS]) {
                double wnew[KS];
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    wnew[s] = z[s] + fz * (w[s] - cwo[s]);
                    vn[s] = fmin(fmax(wnew[s], SPCIES_RC(K, which_lb, s)), SPCIES_RC(K, which_ub, s));
                    res |= (fabs(__builtin_fma(fz, cwo[s], -vn[s])) > tol) | (fabs(z[s] - vn[s]) > tol);
                }
                if (active) {
#pragma unroll
                    for (int s = 0; s < KS; s++) Wt.st(t * KS + s, voff, wnew[s]);
                    if constexpr (WANT_SOL) {
                        const int off = (t == 0) ? -n : (m + (t - 1) * nm);
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            const int row = 4 * s + g;
                            const bool in = (t == 0) ? (row >= n && row < nm) : (t == N ? (TERMINAL && row < n) : row < nm);
                            if (in) z_out[inst * dim + off + row] = z[s];
                        }
                    }
                }
            };
            double yfp[KX], wfp[KS];  // y_l and w_{l+1}, in flight
#pragma unroll
            for (int s = 0; s < KX; s++) yfp[s] = Yt.ld((N - 1) * KX + s, voff);
#pragma unroll
            for (int s = 0; s < KS; s++) wfp[s] = Wt.ld(N * KS + s, voff);
            for (int l = N - 1; l >= 0; l--) {
                const int sq = 2 * N - 1 - l;
                asm volatile("" : "+v"(ao_l));
                stg.issue(seq + (long)((sq + 1 == 2 * N) ? 0 : sq + 1) * LY::CHD);
                const double *ch = ring + slot * LY::CHD;
                const double *K = ch + LY::NT_PAD * 16;
                double yf[KX], wt[KS];
#pragma unroll
                for (int s = 0; s < KX; s++) {
                    yf[s] = yfp[s];
                    if (l > 0) yfp[s] = Yt.ld((l - 1) * KX + s, voff);
                }
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    wt[s] = wfp[s];
                    wfp[s] = Wt.ld(l * KS + s, voff);  // stage l: next iteration, or stage 0 after the loop
                }
                double mu[KX];
#pragma unroll
                for (int s = 0; s < KX; s++) mu[s] = 0.0;
                {
                    int tix = 0;
                    double2 cur;
                    prod<KX, KX, UPPER>(mu, yf, ch, ao_l, tix, cur);
                    prod<KX, KX, DENSE>(mu, mun, ch, ao_l, tix, cur);
                }
                // stage t = l + 1
                const int t = l + 1;
                const bool lastt = (t == N);
                double cwo[KS], qh[KS], z[KS], vn[KS];
                qhat(wt, K, LY::K_LB, LY::K_UB, LY::K_RHO, lastt, fz, cwo, qh);
                if (lastt) {
#pragma unroll
                    for (int s = 0; s < KS; s++) z[s] = 0.0;
                    if constexpr (TERMINAL) {  // z_N = -Hi_N (q_hat_N - mu_{N-1})  (:477-485)
                        double d[KX], zx[KX];
#pragma unroll
                        for (int s = 0; s < KX; s++) {
                            d[s] = mu[s] - qh[s];
                            zx[s] = 0.0;
                        }
                        int tix = LY::T_HIN;
                        double2 cur;
                        prod<KX, KX, DENSE>(zx, d, lds, ao_l, tix, cur);
#pragma unroll
                        for (int s = 0; s < KX; s++) z[s] = zx[s];
                    }
                } else {  // z_t = -Hd_t (q_hat_t - [mu_{t-1}; 0] + AB' mu_t)  (:464-474)
                    double acc[KS];
#pragma unroll
                    for (int s = 0; s < KS; s++) acc[s] = qh[s] - ((s < KX) ? mu[s < KX ? s : 0] : 0.0);
                    int tix = LY::T_ABT;
                    double2 cur;
                    prod<KS, KX, DENSE>(acc, mun, lds, ao_l, tix, cur);
#pragma unroll
                    for (int s = 0; s < KS; s++) z[s] = -SPCIES_RC(K, LY::K_HD, s) * acc[s];
                }
                if (!lastt || TERMINAL) finish(t, wt, z, cwo, K, LY::K_LB, LY::K_UB, vn);
#pragma unroll
                for (int s = 0; s < KX; s++) mun[s] = mu[s];
                if (l == 0) {  // stage 0: z_0 = -Hd_0 (q_hat_0 + AB' mu_0)  (:456-461)
                    double cw0[KS], qh0[KS], acc[KS], z0[KS];
                    qhat(wfp, inv_rc, LY::C_LB0, LY::C_UB0, LY::C_RHO0, false, fz, cw0, qh0);
#pragma unroll
                    for (int s = 0; s < KS; s++) acc[s] = qh0[s];
                    int tix = LY::T_ABT;
                    double2 cur;
                    prod<KS, KX, DENSE>(acc, mun, lds, ao_l, tix, cur);
#pragma unroll
                    for (int s = 0; s < KS; s++) z0[s] = -SPCIES_RC(inv_rc, LY::C_HD0, s) * acc[s];
                    finish(0, wfp, z0, cw0, inv_rc, LY::C_LB0, LY::C_UB0, u_keep);
                }
                stg.commit(ring + (slot ^ 1) * LY::CHD);
                __syncthreads();
                slot ^= 1;
            }
            // ======================= exit (:624-631) =======================
            const bool res_inst = or_over_rows(res, c);
            const bool done_now = active && (!res_inst || kk >= p.k_max);
            if (done_now) {
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const int row = 4 * s + g;
                    if (row >= n && row < nm) u_out[inst * m + (row - n)] = u_keep[s];
                }
                if (g == 0) {
                    k_out[inst] = kk;
                    e_out[inst] = res_inst ? -1 : 1;
                }
                active = false;
            }
            if (!__syncthreads_or(active ? 1 : 0)) break;
        }
    }
#undef SPCIES_RC
}

// v = clamp(w), lambda = rho (w - v) from the frozen w, in the reference's flattened order (:659-684)
__global__ __launch_bounds__(256) void admm_g_record_kernel(const double *__restrict__ W, long B, int N, int KS, int n, int m,
                                                            int terminal, const double *__restrict__ lbS,
                                                            const double *__restrict__ ubS, const double *__restrict__ rhoS,
                                                            double *__restrict__ v_out, double *__restrict__ lam_out) {
    const int nm = n + m;
    const long dim = (long)N * nm - (terminal ? 0 : n);
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * dim) return;
    const long inst = i / dim;
    const int e = (int)(i % dim);
    int t, row;
    if (e < m) {
        t = 0;
        row = n + e;
    } else {
        t = 1 + (e - m) / nm;
        row = (e - m) % nm;
    }
    const long tile = inst / 16;
    const double w = W[((tile * (N + 1) + t) * KS + row / 4) * 64 + 16 * (row % 4) + (inst % 16)];
    const double v = fmin(fmax(w, lbS[t * nm + row]), ubS[t * nm + row]);
    if (v_out) v_out[i] = v;
    if (lam_out) lam_out[i] = rhoS[t * nm + row] * (w - v);
}

#define SPCIES_G4_ADMM_SHAPES(X) X(1, 1) X(1, 2) X(2, 2) X(2, 3) X(3, 3) X(3, 4) X(4, 4) X(4, 5) X(5, 5) X(5, 6) X(6, 6)

inline int admm_plan_build(Plan &p, const AdmmHost &a) {
    p.ok = false;
    const int KX = (a.n + 3) / 4, KS = (a.n + a.m + 3) / 4;
    if (a.N < 2) { p.why = "N < 2"; return 0; }
    for (int l = 1; l < a.N - 1; l++)
        for (int j = 0; j < a.n + a.m; j++)
            if (!std::isfinite(a.Hi[(size_t)l * (a.n + a.m) + j])) { p.why = "non-finite Hi"; return 0; }
#define X(KKX, KKS) \
    if (KX == KKX && KS == KKS) return admm_plan_build_shape<KKX, KKS>(p, a);
    SPCIES_G4_ADMM_SHAPES(X)
#undef X
    p.why = "MFMA4G ADMM kernel not instantiated for this (ceil(n/4), ceil((n+m)/4))";
    return 0;
}

inline size_t admm_state_bytes(const Plan &p, const AdmmHost &a, long B) {
    return (size_t)padded_tiles(B) * ((size_t)(a.N + 1) * p.KS + (size_t)a.N * p.KX) * 64 * sizeof(double);
}

template <int KX, int KS>
static int launch_admm_g_shape(Plan &pl, const AdmmHost &a, const Args &args, const double *x0, const double *xr,
                               const double *ur, double *state, double *u, int *k, int *e, double *z, double *v, double *lam,
                               hipStream_t st) {
    using LY = AdmmGLayout<KX, KS>;
    constexpr int WGS = (KS >= 4) ? 2 : 3;
    const long tiles = padded_tiles(args.B);
    const int N = a.N;
    double *W = state, *Y = W + tiles * (long)(N + 1) * KS * 64;
    const long wgs = std::min(tiles / 4, (long)pl.num_cu * pick_wgs(tiles / 4, pl.num_cu, WGS));
    const size_t shmem = LY::LDS_D * sizeof(double);
    dim3 grid((unsigned)wgs), block(256);
    SPCIES_HIP_CHECK(hipMemsetAsync(W, 0, (size_t)tiles * (size_t)(N + 1) * KS * 64 * sizeof(double), st));  // w = 0: cold start
    const AdmmGArgs ga{args, a.rho};
#define SPCIES_LAUNCH(TERM, SOL) \
    hipLaunchKernelGGL((admm_g_kernel<KX, KS, TERM, SOL, WGS>), grid, block, shmem, st, ga, pl.d_table, x0, xr, ur, W, Y, u, k, e, z)
    if (a.terminal) {
        if (z) SPCIES_LAUNCH(true, true); else SPCIES_LAUNCH(true, false);
    } else {
        if (z) SPCIES_LAUNCH(false, true); else SPCIES_LAUNCH(false, false);
    }
#undef SPCIES_LAUNCH
    SPCIES_HIP_CHECK(hipGetLastError());
    if (v || lam) {
        const long total = args.B * (long)a.dim();
        const double *lbS = pl.d_table + LY::chunks_end(N), *ubS = lbS + (size_t)(N + 1) * (a.n + a.m),
                     *rhoS = ubS + (size_t)(N + 1) * (a.n + a.m);
        hipLaunchKernelGGL(admm_g_record_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, args.B, N, KS, a.n,
                           a.m, a.terminal ? 1 : 0, lbS, ubS, rhoS, v, lam);
        SPCIES_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

inline int launch_admm_g(Plan &pl, const AdmmHost &a, const double *x0, const double *xr, const double *ur, int ref_stride,
                         long B, double *state, double *u, int *k, int *e, double *z, double *v, double *lam, hipStream_t st) {
    if (!pl.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4G variant unavailable: %s", pl.why.c_str());
    // (v, lambda come from the frozen state: they do not need the in-loop z stores)
    Args args{a.n, a.m, a.N, a.k_max, a.tol, B, ref_stride};
#define X(KKX, KKS)                   \
    if (pl.KX == KKX && pl.KS == KKS) \
        return launch_admm_g_shape<KKX, KKS>(pl, a, args, x0, xr, ur, state, u, k, e, z, v, lam, st);
    SPCIES_G4_ADMM_SHAPES(X)
#undef X
    return fail(SPCIES_HIP_ENOSUP, "MFMA4G ADMM kernel not instantiated for KX=%d KS=%d", pl.KX, pl.KS);
}

}  // namespace g4
}  // namespace spcies
