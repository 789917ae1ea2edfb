// Variant MFMA4G of the banded-Cholesky FISTA solver (laxMPC / equMPC), see mfma4g.hpp for the design.
// Algorithm: formulations/+laxMPC/code_laxMPC_FISTA_C.c:296-389 (iteration), :471-539 (z from the dual),
// :546-574 (residual), :582-649 (solve_W); equMPC: code_equMPC_FISTA_C.c (no terminal block, x_N = xr).
//
// In block form, with y_l (n rows) the dual of x_{l+1} = A x_l + B u_l, z_t = (x_t; u_t), Bi_l = Beta_l^-1:
//   z_t  = clamp( hd_t o ( q_t + [y_{t-1}; 0] - AB' y_t ) )          t = 0..N   (y_-1 = y_N = 0; hd_0 has no x rows)
//   r_l  = x_{l+1} - AB z_l (+ b for l = 0, b = -A x0)                l = 0..N-1 (x rows of z_0 are 0)
//   d_l  = Bi_l' r_l - Bi_l' Alpha_{l-1}' d_{l-1}                     forward substitution
//   d_l  = Bi_l  d_l - Bi_l  Alpha_l  d_{l+1}                         backward substitution
//   lambda_l = y_l + d_l,  y_l = lambda_l + (t_{k-1} - 1) / t_k (lambda_l - lambda_l^old)
// State in HBM: y, lambda, forward-substituted d: 3 N ceil(n/4) slab vectors per 16 instances;
// traffic per iteration: 7 N ceil(n/4) x 512 B per 16 instances (y twice, lambda, d read; y, lambda, d written).
#pragma once
#include "mfma4g.hpp"

namespace spcies {
namespace g4 {

#pragma clang fp contract(fast)  // (the STREAM headers included before this one switch contraction off)

template <int KX, int KS>
struct FistaGLayout {
    static constexpr int RC = 4 * KS;  // doubles per row-constant vector
    // stage-invariant region: -AB' (KS x KX blocks), -AB (KX x KS blocks), then row constants
    static constexpr int T_NABT = 0, T_NAB = KS * KX, INV_TILES = 2 * KS * KX;
    enum { C_HD0, C_LB0, C_UB0, C_QR, C_TD, C_COUNT };
    static constexpr int INV_D = INV_TILES * 16 + C_COUNT * RC;
    // one chunk per stage and sweep: lower/upper triangular Bi block + one dense block, then (forward
    // chunks only) hd, lb, ub of stage l + 1
    static constexpr int NT = blk_count(KX, KX, LOWER) + KX * KX, NT_PAD = (NT + 1) / 2 * 2;
    enum { K_HD, K_LB, K_UB, K_COUNT };
    static constexpr int CHD = NT_PAD * 16 + K_COUNT * RC;
    static constexpr int LDS_D = INV_D + 2 * CHD;
    static size_t table_doubles(int N) { return (size_t)INV_D + (size_t)2 * N * CHD; }
};

struct FistaGHost {  // what the packer needs beyond AdmmHost
    const std::vector<double> *QRi, *Td, *Ti;
};

template <int KX, int KS>
inline int fista_plan_build_shape(Plan &p, const AdmmHost &a, const FistaGHost &f) {
    using LY = FistaGLayout<KX, KS>;
    const int n = a.n, m = a.m, N = a.N, nm = n + m;
    std::vector<double> tab(LY::table_doubles(N), 0.0);
    DM AB(n, nm);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < nm; j++) AB(i, j) = a.AB[(size_t)i * nm + j];
    bool ok = true;
    {
        BlockWriter w(tab, 0);
        w.emit(neg(tr(AB)), KS, KX, DENSE);
        w.emit(neg(AB), KX, KS, DENSE);
        ok = ok && w.structure_ok && w.cursor == LY::INV_TILES;
        double *rc = tab.data() + LY::INV_TILES * 16;
        for (int j = 0; j < m; j++) {
            rc[LY::C_HD0 * LY::RC + n + j] = (*f.QRi)[n + j];
            rc[LY::C_LB0 * LY::RC + n + j] = a.LB[n + j];
            rc[LY::C_UB0 * LY::RC + n + j] = a.UB[n + j];
            rc[LY::C_QR * LY::RC + n + j] = a.R[j];
        }
        for (int j = 0; j < n; j++) {
            rc[LY::C_QR * LY::RC + j] = a.Q[j];
            rc[LY::C_TD * LY::RC + j] = a.terminal ? (*f.Td)[j] : 0.0;
        }
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
        if (s < N) {  // forward chunk of block l = s
            const int l = s;
            const DM BiT = tr(Bi[l]);
            w.emit(BiT, KX, KX, LOWER);
            w.emit(l >= 1 ? neg(mul(BiT, tr(Al[l - 1]))) : Zero, KX, KX, DENSE);
            double *rc = tab.data() + base + LY::NT_PAD * 16;
            const int t = l + 1;
            if (t < N) {
                for (int j = 0; j < nm; j++) {
                    rc[LY::K_HD * LY::RC + j] = (*f.QRi)[j];
                    rc[LY::K_LB * LY::RC + j] = a.LB[j];
                    rc[LY::K_UB * LY::RC + j] = a.UB[j];
                }
            } else if (a.terminal) {
                for (int j = 0; j < n; j++) {
                    rc[LY::K_HD * LY::RC + j] = (*f.Ti)[j];
                    rc[LY::K_LB * LY::RC + j] = a.LB[j];
                    rc[LY::K_UB * LY::RC + j] = a.UB[j];
                }
            }
        } else {  // backward chunk of block l = 2N-1-s
            const int l = 2 * N - 1 - s;
            w.emit(Bi[l], KX, KX, UPPER);
            w.emit(l < N - 1 ? neg(mul(Bi[l], Al[l])) : Zero, KX, KX, DENSE);
        }
        ok = ok && w.structure_ok && w.cursor == LY::NT;
    }
    if (!ok) { p.why = "MFMA4G packer: block structure mismatch"; return 0; }
    p.KX = KX;
    p.KS = KS;
    return plan_upload(p, tab);
}

// -------------------------------------------------------------------------------------------------
template <int KX, int KS, bool TERMINAL, bool WANT_SOL, int WG_PER_CU>
__global__ __launch_bounds__(256, WG_PER_CU) void fista_g_kernel(Args p, const double *__restrict__ tab,
                                                                 const double *__restrict__ x0g,
                                                                 const double *__restrict__ xrg,
                                                                 const double *__restrict__ urg, double *__restrict__ Yg,
                                                                 double *__restrict__ Lg, double *__restrict__ Dg,
                                                                 double *__restrict__ u_out, int *__restrict__ k_out,
                                                                 int *__restrict__ e_out, double *__restrict__ z_out) {
    using LY = FistaGLayout<KX, KS>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = p.n, m = p.m, nm = n + m, N = p.N;
    for (int i = threadIdx.x; i < LY::INV_D / 2; i += 256)
        reinterpret_cast<double2 *>(lds)[i] = reinterpret_cast<const double2 *>(tab)[i];
    double *ring = lds + LY::INV_D;
    const double *seq = tab + LY::INV_D;
    const double *inv_rc = lds + LY::INV_TILES * 16;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: state addresses are SGPR base + lane offset
    const int g = lane >> 4, c = lane & 15;
    const int ao = g * 4 + (lane & 3);
    const long n_tiles = (p.B + 15) / 16, n_groups = (n_tiles + 3) / 4;
    const double tol = p.tol;
    const int dim = TERMINAL ? N * nm : N * nm - n;
    const long NV = (long)N * KX;  // slab vectors per tile in each state array
    Stager<LY::CHD> stg;

    for (long group = blockIdx.x; group < n_groups; group += gridDim.x) {
        const long tile = group * 4 + wave;
        const long inst = tile * 16 + c;
        const bool valid = inst < p.B;
        const SlabBuf Yt(Yg + tile * NV * 64, NV), Lt(Lg + tile * NV * 64, NV), Dt(Dg + tile * NV * 64, NV);
        const int voff = lane * 8;
        // ---- per-instance setup (code_laxMPC_FISTA_C.c:274-289)
        double qm[KS], qN[KS], xrv[KS], bvec[KX];
        {
            double x0v[KS];
            const double *xrp = p.ref_stride ? xrg + inst * n : xrg;
            const double *urp = p.ref_stride ? urg + inst * m : urg;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int row = 4 * s + g;
                double xu = 0.0;
                x0v[s] = 0.0;
                xrv[s] = 0.0;
                if (valid && row < n) {
                    x0v[s] = x0g[inst * n + row];
                    xrv[s] = xrp[row];
                    xu = xrv[s];
                } else if (valid && row < nm) {
                    xu = urp[row - n];
                }
                qm[s] = inv_rc[LY::C_QR * LY::RC + row] * xu;
                qN[s] = inv_rc[LY::C_TD * LY::RC + row] * xrv[s];
            }
            __syncthreads();  // invariant region visible (first group) / previous group done with the ring
#pragma unroll
            for (int s = 0; s < KX; s++) bvec[s] = 0.0;
            int tix = LY::T_NAB;
            double2 cur;
            prod<KX, KS, DENSE>(bvec, x0v, lds, ao, tix, cur);  // b = -A x0
        }
        stg.issue(seq);
        stg.commit(ring);
        __syncthreads();
        int slot = 0;

        int ao_l = ao;
        bool active = valid, init = true;
        int kk = 0;
        double tk = 1.0, tk1 = 1.0;
        while (true) {
            if (!init) {
                kk += 1;
                tk1 = tk;
            }
            // ======================= forward sweep =======================
            bool res = false;
            double yc[KX], zc[KS], dprev[KX], z0[KS];
#pragma unroll
            for (int s = 0; s < KX; s++) {
                yc[s] = Yt.ld(s, voff);  // (y = lambda = 0 before the initial step: zero-filled by the launcher)
                dprev[s] = 0.0;
            }
            {  // z_0 (:474-491): only the u rows are free
                double acc[KS];
#pragma unroll
                for (int s = 0; s < KS; s++) acc[s] = qm[s];
                int tix = LY::T_NABT;
                double2 cur;
                prod<KS, KX, DENSE>(acc, yc, lds, ao_l, tix, cur);
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const int row = 4 * s + g;
                    zc[s] = fmin(fmax(inv_rc[LY::C_HD0 * LY::RC + row] * acc[s], inv_rc[LY::C_LB0 * LY::RC + row]),
                                 inv_rc[LY::C_UB0 * LY::RC + row]);
                    z0[s] = zc[s];
                    if constexpr (WANT_SOL) {
                        if (active && row >= n && row < nm) z_out[inst * dim + (row - n)] = zc[s];
                    }
                }
            }
            double ypre[KX];  // y_{l+2}, in flight during stage l
#pragma unroll
            for (int s = 0; s < KX; s++) ypre[s] = Yt.ld(KX + s, voff);
            for (int l = 0; l < N; l++) {
                asm volatile("" : "+v"(ao_l));  // keeps LICM from hoisting the stage-invariant LDS block reads
                stg.issue(seq + (long)(l + 1) * LY::CHD);
                const double *ch = ring + slot * LY::CHD;
                const double *rc = ch + LY::NT_PAD * 16;
                double yn[KX];
#pragma unroll
                for (int s = 0; s < KX; s++) {
                    yn[s] = (l + 1 < N) ? ypre[s] : 0.0;
                    ypre[s] = (l + 2 < N) ? Yt.ld((l + 2) * KX + s, voff) : 0.0;
                }
                // z_{l+1} (:494-537)
                double zn[KS];
                if (l + 1 == N && !TERMINAL) {
#pragma unroll
                    for (int s = 0; s < KS; s++) zn[s] = xrv[s];
                } else {
                    double acc[KS];
                    const bool lastt = (l + 1 == N);
#pragma unroll
                    for (int s = 0; s < KS; s++) acc[s] = (lastt ? qN[s] : qm[s]) + (s < KX ? yc[s < KX ? s : 0] : 0.0);
                    int tix = LY::T_NABT;
                    double2 cur;
                    prod<KS, KX, DENSE>(acc, yn, lds, ao_l, tix, cur);
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        const int row = 4 * s + g;
                        zn[s] = fmin(fmax(rc[LY::K_HD * LY::RC + row] * acc[s], rc[LY::K_LB * LY::RC + row]),
                                     rc[LY::K_UB * LY::RC + row]);
                    }
                }
                if constexpr (WANT_SOL) {
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        const int row = 4 * s + g;
                        const bool in = (l + 1 == N) ? (TERMINAL && row < n) : (row < nm);
                        if (active && in) z_out[inst * dim + m + (long)l * nm + row] = zn[s];
                    }
                }
                // residual block l (:546-574) and exit flag (:330-344)
                double r[KX];
#pragma unroll
                for (int s = 0; s < KX; s++) {
                    const int row = 4 * s + g;
                    r[s] = (row < n) ? zn[s] : 0.0;
                    if (l == 0) r[s] += bvec[s];
                }
                {
                    int tix = LY::T_NAB;
                    double2 cur;
                    prod<KX, KS, DENSE>(r, zc, lds, ao_l, tix, cur);
                }
#pragma unroll
                for (int s = 0; s < KX; s++) res |= fabs(r[s]) > tol;
                // forward substitution (:582-612)
                double d[KX];
#pragma unroll
                for (int s = 0; s < KX; s++) d[s] = 0.0;
                {
                    int tix = 0;
                    double2 cur;
                    prod<KX, KX, LOWER>(d, r, ch, ao_l, tix, cur);
                    prod<KX, KX, DENSE>(d, dprev, ch, ao_l, tix, cur);
                }
#pragma unroll
                for (int s = 0; s < KX; s++) {
                    Dt.st(l * KX + s, voff, d[s]);
                    dprev[s] = d[s];
                    yc[s] = yn[s];
                }
#pragma unroll
                for (int s = 0; s < KS; s++) zc[s] = zn[s];
                stg.commit(ring + (slot ^ 1) * LY::CHD);
                __syncthreads();
                slot ^= 1;
            }
            // ======================= exit (:346-353) =======================
            if (!init) {
                const bool res_inst = or_over_rows(res, c);
                const bool done_now = active && (!res_inst || kk >= p.k_max);
                if (done_now) {
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        const int row = 4 * s + g;
                        if (row >= n && row < nm) u_out[inst * m + (row - n)] = z0[s];
                    }
                    if (g == 0) {
                        k_out[inst] = kk;
                        e_out[inst] = res_inst ? -1 : 1;
                    }
                    active = false;
                }
                if (!__syncthreads_or(active ? 1 : 0)) break;
                tk = 0.5 * (1.0 + sqrt(1.0 + 4.0 * tk1 * tk1));
            }
            const double beta = init ? 0.0 : (tk1 - 1.0) / tk;
            // ======================= backward sweep: d = W^-1 r, lambda, y (:357-385) =======================
            double dn[KX], dfp[KX], yp[KX], lp[KX];
#pragma unroll
            for (int s = 0; s < KX; s++) {
                dn[s] = 0.0;
                dfp[s] = Dt.ld((N - 1) * KX + s, voff);
                yp[s] = Yt.ld((N - 1) * KX + s, voff);
                lp[s] = Lt.ld((N - 1) * KX + s, voff);
            }
            for (int l = N - 1; l >= 0; l--) {
                const int sq = 2 * N - 1 - l;
                asm volatile("" : "+v"(ao_l));
                stg.issue(seq + (long)((sq + 1 == 2 * N) ? 0 : sq + 1) * LY::CHD);
                const double *ch = ring + slot * LY::CHD;
                double df[KX], yv[KX], lv[KX];
#pragma unroll
                for (int s = 0; s < KX; s++) {
                    df[s] = dfp[s];
                    yv[s] = yp[s];
                    lv[s] = lp[s];
                    if (l > 0) {
                        dfp[s] = Dt.ld((l - 1) * KX + s, voff);
                        yp[s] = Yt.ld((l - 1) * KX + s, voff);
                        lp[s] = Lt.ld((l - 1) * KX + s, voff);
                    }
                }
                double d[KX];
#pragma unroll
                for (int s = 0; s < KX; s++) d[s] = 0.0;
                {
                    int tix = 0;
                    double2 cur;
                    prod<KX, KX, UPPER>(d, df, ch, ao_l, tix, cur);
                    prod<KX, KX, DENSE>(d, dn, ch, ao_l, tix, cur);
                }
                double lnw[KX], ynw[KX];
#pragma unroll
                for (int s = 0; s < KX; s++) {
                    lnw[s] = yv[s] + d[s];
                    ynw[s] = lnw[s] + beta * (lnw[s] - lv[s]);
                    dn[s] = d[s];
                }
                if (active) {
#pragma unroll
                    for (int s = 0; s < KX; s++) {
                        Lt.st(l * KX + s, voff, lnw[s]);
                        Yt.st(l * KX + s, voff, ynw[s]);
                    }
                }
                stg.commit(ring + (slot ^ 1) * LY::CHD);
                __syncthreads();
                slot ^= 1;
            }
            init = false;
        }
    }
}

#define SPCIES_G4_FISTA_SHAPES(X) X(1, 1) X(1, 2) X(2, 2) X(2, 3) X(3, 3) X(3, 4) X(4, 4) X(4, 5) X(5, 5) X(5, 6) X(6, 6)

inline int fista_plan_build(Plan &p, const AdmmHost &a, const FistaGHost &f) {
    p.ok = false;
    const int KX = (a.n + 3) / 4, KS = (a.n + a.m + 3) / 4;
    if (a.N < 2) { p.why = "N < 2"; return 0; }
#define X(KKX, KKS) \
    if (KX == KKX && KS == KKS) return fista_plan_build_shape<KKX, KKS>(p, a, f);
    SPCIES_G4_FISTA_SHAPES(X)
#undef X
    p.why = "MFMA4G FISTA kernel not instantiated for this (ceil(n/4), ceil((n+m)/4))";
    return 0;
}

inline size_t fista_state_bytes(const Plan &p, const AdmmHost &a, long B) {
    return (size_t)3 * padded_tiles(B) * a.N * p.KX * 64 * sizeof(double);
}

template <int KX, int KS>
static int launch_fista_g_shape(Plan &pl, const AdmmHost &a, const Args &args, const double *x0, const double *xr,
                                const double *ur, double *state, double *u, int *k, int *e, double *z, double *lam,
                                hipStream_t st) {
    using LY = FistaGLayout<KX, KS>;
    // compiled for up to 3 workgroups per CU (<= 170 registers) where the shape allows it; see pick_wgs()
    constexpr int WGS = (KS >= 5) ? 2 : 3;
    const long tiles = padded_tiles(args.B), NV = (long)a.N * KX;
    double *Y = state, *L = Y + tiles * NV * 64, *D = L + tiles * NV * 64;
    const long wgs = std::min(tiles / 4, (long)pl.num_cu * pick_wgs(tiles / 4, pl.num_cu, WGS));
    const size_t shmem = LY::LDS_D * sizeof(double);
    dim3 grid((unsigned)wgs), block(256);
    // y = lambda = 0 before the initial step (code_laxMPC_FISTA_C.c:296-318); Y and L are adjacent
    SPCIES_HIP_CHECK(hipMemsetAsync(Y, 0, (size_t)2 * tiles * NV * 64 * sizeof(double), st));
#define SPCIES_LAUNCH(TERM, SOL)                                                                                  \
    hipLaunchKernelGGL((fista_g_kernel<KX, KS, TERM, SOL, WGS>), grid, block, shmem, st, args, pl.d_table, x0, xr, ur, Y, \
                       L, D, u, k, e, z)
    if (a.terminal) {
        if (z) SPCIES_LAUNCH(true, true); else SPCIES_LAUNCH(true, false);
    } else {
        if (z) SPCIES_LAUNCH(false, true); else SPCIES_LAUNCH(false, false);
    }
#undef SPCIES_LAUNCH
    SPCIES_HIP_CHECK(hipGetLastError());
    if (lam) {  // the reference returns y as sol.lambda (code_laxMPC_FISTA_C.c:439-445)
        const long total = args.B * (long)a.N * a.n;
        hipLaunchKernelGGL(tile_state_to_aos_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, Y, args.B, a.N,
                           KX, a.n, lam);
        SPCIES_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

inline int launch_fista_g(Plan &pl, const AdmmHost &a, const double *x0, const double *xr, const double *ur, int ref_stride,
                          long B, double *state, double *u, int *k, int *e, double *z, double *lam, hipStream_t st) {
    if (!pl.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4G variant unavailable: %s", pl.why.c_str());
    Args args{a.n, a.m, a.N, a.k_max, a.tol, B, ref_stride};
#define X(KKX, KKS) \
    if (pl.KX == KKX && pl.KS == KKS) return launch_fista_g_shape<KKX, KKS>(pl, a, args, x0, xr, ur, state, u, k, e, z, lam, st);
    SPCIES_G4_FISTA_SHAPES(X)
#undef X
    return fail(SPCIES_HIP_ENOSUP, "MFMA4G FISTA kernel not instantiated for KX=%d KS=%d", pl.KX, pl.KS);
}

}  // namespace g4
}  // namespace spcies
