// Variant STREAM of the banded-Cholesky ADMM solver (laxMPC / equMPC): ONE LANE PER INSTANCE.
//
// 64 independent instances per wavefront run the reference's iteration in lock step; every
// floating-point accumulation is performed in the reference's own order
// (formulations/+laxMPC/code_laxMPC_ADMM_C.c:308-633, equMPC: code_equMPC_ADMM_C.c), so with
// EXACT = true (no FMA contraction) the results are bit-identical to the generated C built by
// gcc -O3 on x86-64.  Controller constants are wave-uniform: they are fetched with scalar loads
// (s_load) and fed to v_fma_f64 / v_mul_f64 as SGPR operands - no LDS, no lane idles.
//
// State that has to survive between iterations (v, lambda) and between the two sweeps of one
// iteration (the forward-substituted y) does not fit on chip at 64 instances per wave
// (600 doubles per instance), so it is streamed through HBM in a structure-of-arrays scratch
// [element][instance]: every access is a fully coalesced 512-byte wave transaction.
//
// Per iteration and instance (n=12, m=2, N=15): reads 2 x (v, lambda) + y, writes y + (v, lambda)
// = (4*420 + 2*180) * 8 B = 16.3 KB  ->  HBM-bound kernel.
#pragma once
#include "common.hpp"
#include "tv_update_kernel.inc"  // time-varying solvers: row layouts of their scratch (TvLayout, FistaTvLayout) and the update-phase kernels

namespace spcies {

#pragma clang fp contract(off)

template <bool EXACT>
__device__ __forceinline__ double msub(double acc, double a, double b) {  // acc - a*b
    if constexpr (EXACT)
        return acc - a * b;
    else
        return __builtin_fma(-a, b, acc);
}
template <bool EXACT>
__device__ __forceinline__ double madd(double acc, double a, double b) {  // acc + a*b
    if constexpr (EXACT)
        return acc + a * b;
    else
        return __builtin_fma(a, b, acc);
}

__device__ __forceinline__ double clamp_ref(double x, double lo, double hi) {
    x = (x > lo) ? x : lo;  // comparison sense of code_laxMPC_ADMM_C.c:500-501
    x = (x > hi) ? hi : x;
    return x;
}

__device__ __forceinline__ bool above(double a, double b, double tol) {
    double r = a - b;
    r = (r > 0.0) ? r : -r;
    return r > tol;
}

// A controller constant array: shared by all instances (pointer into the constants allocation, fetched with
// scalar loads) or, for the time-varying solvers, per instance in the structure-of-arrays scratch [row][Bp].
template <bool TV>
struct KArr;
template <>
struct KArr<false> {
    const double *p;
    __device__ __forceinline__ double operator[](long i) const { return p[i]; }
    __device__ __forceinline__ KArr operator+(long off) const { return KArr{p + off}; }
};
// Per-instance rows are read through a buffer resource (SGPR descriptor over the whole scratch, SGPR row
// offset, one VGPR lane offset): with flat addressing hipcc materialises one 64-bit VGPR pointer per
// constant of the unrolled sweeps and spills hundreds of them.  The scratch of one launch stays below 4 GB
// (the host splits larger batches).
template <>
struct KArr<true> {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t r;
    unsigned row, bp8, voff;  // first row, Bp * 8, 8 * instance
    __device__ __forceinline__ double operator[](long i) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, (row + (unsigned)i) * bp8, 0));
    }
    __device__ __forceinline__ KArr operator+(long off) const { return KArr{r, row + (unsigned)off, bp8, voff}; }
};


// Scratch: V, LAM are [dim][Bp], Y is [N*n][Bp], ZS (optional, only when the caller wants
// z / v / lambda back) is [dim][Bp].  Element order inside dim = the reference's flattened order
// (code_laxMPC_ADMM_C.c:659-684): m head entries, N-1 rows of n+m, n tail entries.
// time-varying instantiation: every constant is a per-lane global load; a compiler barrier per row keeps hipcc
// from hoisting a whole sweep's loads to the top (510 spilled registers without it)
#ifndef SPCIES_TV_BARRIER_EVERY
#define SPCIES_TV_BARRIER_EVERY 1
#endif
#define SPCIES_TV_ROW_BARRIER_AT(j)                                                            \
    do {                                                                                       \
        if constexpr (TV) {                                                                    \
            if ((j) % SPCIES_TV_BARRIER_EVERY == 0) asm volatile("" ::: "memory");             \
        }                                                                                      \
    } while (0)
#define SPCIES_TV_ROW_BARRIER() SPCIES_TV_ROW_BARRIER_AT(j)

template <int n, int m, bool TERMINAL, bool EXACT, bool TV = false, bool ELLIP = false, bool GEN = false>
__global__ __launch_bounds__(64) void admm_stream_kernel(AdmmDev c, const double *__restrict__ C,
                                                         const double *__restrict__ x0g,
                                                         const double *__restrict__ xrg,
                                                         const double *__restrict__ urg, int ref_stride, long B,
                                                         long Bp, double *__restrict__ V,
                                                         double *__restrict__ LAM, double *__restrict__ Y,
                                                         double *__restrict__ ZS, double *__restrict__ u_out,
                                                         int *__restrict__ k_out, int *__restrict__ e_out,
                                                         const double *__restrict__ TVS = nullptr) {
    constexpr int nm = n + m;
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= B) return;
    const int N = c.N;
    const double rho = c.rho, rho_i = c.rho_i, tol = c.tol;
    // time-varying: everything but Hi_N (= T_rho_i) and T comes from the instance's own rows (written by
    // admm_tv_update_kernel); otherwise from the shared constants
    const TvLayout tl = tv_layout(n, m, N);
    auto K = [&](int shared_off, int tv_row) {
        if constexpr (TV) {
            return KArr<true>{__builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(TVS), 0, -1, 0x00020000), (unsigned)tv_row,
                              (unsigned)(Bp * 8), (unsigned)(t * 8)};
        } else {
            return KArr<false>{C + shared_off};
        }
    };
    const KArr<TV> cAB = K(c.AB, tl.AB), cAlpha = K(c.Alpha, tl.Alpha), cBeta = K(c.Beta, tl.Beta), cHi = K(c.Hi, tl.Hi),
                   cHi_0 = K(c.Hi_0, tl.Hi_0), cQ = K(c.Q, tl.Q), cR = K(c.R, tl.R), cLB = K(c.LB, tl.LB), cUB = K(c.UB, tl.UB);
    const double *cHi_N = C + c.Hi_N, *cT = C + c.T;
    // ellipMPC ADMM (code_ellipMPC_ADMM_C.c): terminal ellipsoid constants and stage-wise bounds
    const double *cP = C + c.P, *cPh = C + c.P_half, *cPih = C + c.Pinv_half, *cCe = C + c.c_ell, *cLBz = C + c.LBz,
                 *cUBz = C + c.UBz, *cLBu0 = C + c.LBu0, *cUBu0 = C + c.UBu0;
    static_assert(!ELLIP || (TERMINAL && !TV), "ellipMPC ADMM: terminal block, constant model");
    static_assert(!GEN || !TV, "vector rho / VAR_BOUNDS: constant model (with ELLIP: vector rho, cons_ellipMPC_ADMM_C.m:111-117)");
    // GEN: stage-wise penalty and bounds (no SCALAR_RHO / VAR_BOUNDS, code_laxMPC_ADMM_C.c:323-348, 490-568)
    const double *gR0 = C + c.rho_0, *gRv = C + c.rho_v, *gRN = C + c.rho_N, *gRi0 = C + c.rho_i_0, *gRiv = C + c.rho_i_v,
                 *gRiN = C + c.rho_i_N, *gLBN = C + c.LBN, *gUBN = C + c.UBN;
    auto RH = [&](int j) { return GEN ? gR0[j] : rho; };
    auto RM = [&](int l, int j) { return GEN ? gRv[l * nm + j] : rho; };
    auto RT = [&](int j) { return GEN ? gRN[j] : rho; };
    auto RIH = [&](int j) { return GEN ? gRi0[j] : rho_i; };
    auto RIM = [&](int l, int j) { return GEN ? gRiv[l * nm + j] : rho_i; };
    auto RIT = [&](int j) { return GEN ? gRiN[j] : rho_i; };

    // ---- per-instance setup (code_laxMPC_ADMM_C.c:282-299)
    double xr[n], b[n], q[nm], qT[n];
    {
        double x0[n];
#pragma unroll
        for (int i = 0; i < n; i++) x0[i] = x0g[t * n + i];
        const double *xrp = ref_stride ? xrg + t * n : xrg;
        const double *urp = ref_stride ? urg + t * m : urg;
#pragma unroll
        for (int i = 0; i < n; i++) xr[i] = xrp[i];
#pragma unroll
        for (int j = 0; j < n; j++) {
            double acc = 0.0;
#pragma unroll
            for (int i = 0; i < n; i++) acc = msub<EXACT>(acc, cAB[j * nm + i], x0[i]);
            b[j] = acc;
        }
#pragma unroll
        for (int j = 0; j < n; j++) {
            q[j] = cQ[j] * xr[j];
            double acc = 0.0;
            if constexpr (TERMINAL) {
#pragma unroll
                for (int i = 0; i < n; i++) acc = madd<EXACT>(acc, cT[j * n + i], xr[i]);
            }
            qT[j] = acc;
        }
#pragma unroll
        for (int j = 0; j < m; j++) q[n + j] = cR[j] * urp[j];
    }

    // q_hat of the terminal block: qT + lambda - rho v (lax, :343-349) or qT + P_half lambda - P rho v (ellip, :146-156)
    auto qhat_tail = [&](const double (&lamv)[n], const double (&vold)[n], double (&out)[n]) {
#pragma unroll
        for (int j = 0; j < n; j++) {
            if constexpr (ELLIP) {
                double acc = qT[j];
#pragma unroll
                for (int i = 0; i < n; i++) acc = acc + cPh[j * n + i] * lamv[i] - cP[j * n + i] * RT(i) * vold[i];
                out[j] = acc;
            } else {
                out[j] = qT[j] + lamv[j] - RT(j) * vold[j];
            }
        }
    };
    const long off_mid = (long)m;                       // element offset of the middle rows
    const long off_tail = (long)m + (long)(N - 1) * nm; // element offset of the tail
    double *Vt = V + t, *Lt = LAM + t, *Yt = Y + t;
    double *Zt = ZS ? ZS + t : nullptr;

    int k = 0, flag = -1;
    double u0[m];
    while (true) {
        k += 1;
        const bool first = (k == 1);  // v = lambda = 0: skip the scratch reads (scratch is never pre-zeroed)

        // ================= forward sweep: q_hat, rhs, forward substitution (:323-417) =========
        double qp[nm];  // q_hat of the previous reference block
        double yp[n];
        // stage 0 head (m entries)
        double h0[m];
#pragma unroll
        for (int j = 0; j < m; j++) {
            double lam = first ? 0.0 : Lt[(long)j * Bp];
            double vv = first ? 0.0 : Vt[(long)j * Bp];
            h0[j] = q[n + j] + lam - RH(j) * vv;
        }
        for (int l = 0; l < N; l++) {
            // q_hat of reference block l  (= z[l][.] for l < N-1, z_N[.] for l = N-1)
            double qc[nm];
            const bool last = (l == N - 1);
            if (!last) {
                // (the block's lambda / v rows are gathered before they are used: hipcc otherwise waits for every pair of loads)
                double lamv[nm], vvv[nm];
#pragma unroll
                for (int j = 0; j < nm; j++) {
                    lamv[j] = 0.0;
                    vvv[j] = 0.0;
                }
                if (!first) {
#pragma unroll
                    for (int j = 0; j < nm; j++) {
                        long e = off_mid + (long)l * nm + j;
                        lamv[j] = Lt[e * Bp];
                        vvv[j] = Vt[e * Bp];
                    }
                }
                if constexpr (!TV) __builtin_amdgcn_sched_barrier(0);  // (time-varying: the registers go to the coefficient rows)
#pragma unroll
                for (int j = 0; j < nm; j++) qc[j] = q[j] + lamv[j] - RM(l, j) * vvv[j];
            } else if constexpr (TERMINAL) {
                double lamN[n], vN[n], qN[n];
#pragma unroll
                for (int j = 0; j < n; j++) {
                    long e = off_tail + j;
                    lamN[j] = first ? 0.0 : Lt[e * Bp];
                    vN[j] = first ? 0.0 : Vt[e * Bp];
                }
                qhat_tail(lamN, vN, qN);
#pragma unroll
                for (int j = 0; j < n; j++) qc[j] = qN[j];
            }
            // right-hand side (:355-381)
            double y[n];
#pragma unroll
            for (int j = 0; j < n; j++) {
                SPCIES_TV_ROW_BARRIER();
                double acc;
                if (l == 0) {
                    acc = cHi[j] * qc[j] - b[j];
#pragma unroll
                    for (int i = 0; i < m; i++) acc = acc - cAB[j * nm + n + i] * cHi_0[i] * h0[i];
                } else {
                    if (!last) {
                        acc = cHi[l * nm + j] * qc[j];
                    } else {
                        acc = 0.0;
                        if constexpr (TERMINAL) {
#pragma unroll
                            for (int i = 0; i < n; i++) acc = madd<EXACT>(acc, cHi_N[j * n + i], qc[i]);
                        }
                    }
                    if constexpr (TV) {  // all loads of the row first (the compiler otherwise waits for every pair: 2 loads in flight)
                        double ab_[nm], hi_[nm];
#pragma unroll
                        for (int i = 0; i < nm; i++) {
                            ab_[i] = cAB[j * nm + i];
                            hi_[i] = cHi[(l - 1) * nm + i];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < nm; i++) acc = acc - ab_[i] * hi_[i] * qp[i];
                    } else {
#pragma unroll
                        for (int i = 0; i < nm; i++) acc = acc - cAB[j * nm + i] * cHi[(l - 1) * nm + i] * qp[i];
                    }
                    if (last) {
                        if constexpr (!TERMINAL) acc = acc - xr[j];
                    }
                }
                y[j] = acc;
            }
            // forward substitution (:388-417)
            const KArr<TV> Bl = cBeta + (long)l * n * n;
            const KArr<TV> Al = cAlpha + (long)(l - 1) * n * n;
#pragma unroll
            for (int j = 0; j < n; j++) {
                SPCIES_TV_ROW_BARRIER();
                double acc = y[j];
                if constexpr (TV) {
                    double al_[n], bl_[n];
                    if (l > 0) {
#pragma unroll
                        for (int i = 0; i < n; i++) al_[i] = Al[i * n + j];
                    }
#pragma unroll
                    for (int i = 0; i <= j; i++) bl_[i] = Bl[i * n + j];
                    __builtin_amdgcn_sched_barrier(0);
                    if (l > 0) {
#pragma unroll
                        for (int i = 0; i < n; i++) acc = msub<EXACT>(acc, al_[i], yp[i]);
                    }
#pragma unroll
                    for (int i = 0; i < j; i++) acc = msub<EXACT>(acc, bl_[i], y[i]);
                    y[j] = bl_[j] * acc;
                } else {
                    if (l > 0) {
#pragma unroll
                        for (int i = 0; i < n; i++) acc = msub<EXACT>(acc, Al[i * n + j], yp[i]);
                    }
#pragma unroll
                    for (int i = 0; i < j; i++) acc = msub<EXACT>(acc, Bl[i * n + j], y[i]);
                    y[j] = Bl[j * n + j] * acc;
                }
            }
#pragma unroll
            for (int j = 0; j < n; j++) {
                Yt[((long)l * n + j) * Bp] = y[j];
                yp[j] = y[j];
            }
#pragma unroll
            for (int j = 0; j < nm; j++) qp[j] = qc[j];
        }

        // ================= backward sweep + z, v, lambda, residual (:422-620) ==================
        bool res = false;
        double mun[n];  // mu of block l+1
        for (int l = N - 1; l >= 0; l--) {
            const KArr<TV> Bl = cBeta + (long)l * n * n;
            const KArr<TV> Al = cAlpha + (long)l * n * n;
            double mu[n];
#pragma unroll
            for (int j = 0; j < n; j++) mu[j] = (l == N - 1) ? yp[j] : Yt[((long)l * n + j) * Bp];
#pragma unroll
            for (int j = n - 1; j >= 0; j--) {
                SPCIES_TV_ROW_BARRIER();
                double acc = mu[j];
                if constexpr (TV) {
                    double al_[n], bl_[n];
                    if (l < N - 1) {
#pragma unroll
                        for (int i = 0; i < n; i++) al_[i] = Al[j * n + i];
                    }
#pragma unroll
                    for (int i = j; i < n; i++) bl_[i] = Bl[j * n + i];
                    __builtin_amdgcn_sched_barrier(0);
                    if (l < N - 1) {
#pragma unroll
                        for (int i = n - 1; i >= 0; i--) acc = msub<EXACT>(acc, al_[i], mun[i]);
                    }
#pragma unroll
                    for (int i = n - 1; i > j; i--) acc = msub<EXACT>(acc, bl_[i], mu[i]);
                    mu[j] = bl_[j] * acc;
                } else {
                    if (l < N - 1) {
#pragma unroll
                        for (int i = n - 1; i >= 0; i--) acc = msub<EXACT>(acc, Al[j * n + i], mun[i]);
                    }
#pragma unroll
                    for (int i = n - 1; i > j; i--) acc = msub<EXACT>(acc, Bl[j * n + i], mu[i]);
                    mu[j] = Bl[j * n + j] * acc;
                }
            }
            if (l == N - 1) {
                if constexpr (TERMINAL) {
                    // z_N = -Hi_N (q_hat_N - mu_{N-1})  (:477-485)
                    double aux[n], lamv[n], vold[n];
#pragma unroll
                    for (int j = 0; j < n; j++) {
                        long e = off_tail + j;
                        lamv[j] = first ? 0.0 : Lt[e * Bp];
                        vold[j] = first ? 0.0 : Vt[e * Bp];
                    }
                    qhat_tail(lamv, vold, aux);
#pragma unroll
                    for (int j = 0; j < n; j++) aux[j] = aux[j] - mu[j];
                    if constexpr (ELLIP) {
                        // z_N, then the P-projection onto the ellipsoid (:318-352) and lambda_N through P_half (:374-386)
                        double zN[n], vn[n], pv[n];
#pragma unroll
                        for (int j = 0; j < n; j++) {
                            double zz = 0.0;
#pragma unroll
                            for (int i = 0; i < n; i++) zz = msub<EXACT>(zz, cHi_N[j * n + i], aux[i]);
                            zN[j] = zz;
                        }
#pragma unroll
                        for (int j = 0; j < n; j++) {
                            double acc = zN[j];
#pragma unroll
                            for (int i = 0; i < n; i++) acc = acc + cPih[j * n + i] * RIT(i) * lamv[i];
                            vn[j] = acc;
                        }
#pragma unroll
                        for (int j = 0; j < n; j++) {
                            double acc = 0.0;
#pragma unroll
                            for (int i = 0; i < n; i++) acc = acc + cP[j * n + i] * (vn[i] - cCe[i]);
                            pv[j] = acc;
                        }
                        double vPv = 0.0;
#pragma unroll
                        for (int j = 0; j < n; j++) vPv = vPv + (vn[j] - cCe[j]) * pv[j];
                        if (vPv > c.r_ell * c.r_ell) {
                            vPv = c.r_ell / sqrt(vPv);
#pragma unroll
                            for (int j = 0; j < n; j++) vn[j] = vPv * (vn[j] - cCe[j]) + cCe[j];
                        }
#pragma unroll
                        for (int j = 0; j < n; j++) pv[j] = RT(j) * (zN[j] - vn[j]);
#pragma unroll
                        for (int j = 0; j < n; j++) {
                            double ln = lamv[j];
#pragma unroll
                            for (int i = 0; i < n; i++) ln = ln + cPh[j * n + i] * pv[i];
                            res = res || above(vold[j], vn[j], tol) || above(zN[j], vn[j], tol);
                            long e = off_tail + j;
                            Vt[e * Bp] = vn[j];
                            Lt[e * Bp] = ln;
                            if (Zt) Zt[e * Bp] = zN[j];
                        }
                    } else {
#pragma unroll
                        for (int j = 0; j < n; j++) {
                            double zz = 0.0;
#pragma unroll
                            for (int i = 0; i < n; i++) zz = msub<EXACT>(zz, cHi_N[j * n + i], aux[i]);
                            double vn = GEN ? clamp_ref(zz + RIT(j) * lamv[j], gLBN[j], gUBN[j])
                                            : clamp_ref(zz + rho_i * lamv[j], cLB[j], cUB[j]);
                            double ln = lamv[j] + RT(j) * (zz - vn);
                            res = res || above(vold[j], vn, tol) || above(zz, vn, tol);
                            long e = off_tail + j;
                            Vt[e * Bp] = vn;
                            Lt[e * Bp] = ln;
                            if (Zt) Zt[e * Bp] = zz;
                        }
                    }
                }
            } else {
                // reference block l: z[l] = -Hi[l] (q_hat - [mu_l; 0] + AB' mu_{l+1})  (:464-474)
                double lamb[nm], vob[nm];  // lambda / v of the block, gathered before the rows are processed
#pragma unroll
                for (int j = 0; j < nm; j++) {
                    lamb[j] = 0.0;
                    vob[j] = 0.0;
                }
                if (!first) {
#pragma unroll
                    for (int j = 0; j < nm; j++) {
                        long e = off_mid + (long)l * nm + j;
                        lamb[j] = Lt[e * Bp];
                        vob[j] = Vt[e * Bp];
                    }
                }
                if constexpr (!TV) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < nm; j++) {
                    SPCIES_TV_ROW_BARRIER();
                    long e = off_mid + (long)l * nm + j;
                    double lam = lamb[j];
                    double vold = vob[j];
                    double zz = q[j] + lam - RM(l, j) * vold;
                    if (j < n) zz = zz - mu[j];
                    if constexpr (TV) {
                        double ab_[n];
#pragma unroll
                        for (int i = 0; i < n; i++) ab_[i] = cAB[i * nm + j];
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int i = 0; i < n; i++) zz = madd<EXACT>(zz, ab_[i], mun[i]);
                    } else {
#pragma unroll
                        for (int i = 0; i < n; i++) zz = madd<EXACT>(zz, cAB[i * nm + j], mun[i]);
                    }
                    zz = -cHi[l * nm + j] * zz;
                    double vn = (ELLIP || GEN) ? clamp_ref(zz + RIM(l, j) * lam, cLBz[l * nm + j], cUBz[l * nm + j])
                                               : clamp_ref(zz + rho_i * lam, cLB[j], cUB[j]);
                    double ln = lam + RM(l, j) * (zz - vn);
                    res = res || above(vold, vn, tol) || above(zz, vn, tol);
                    Vt[e * Bp] = vn;
                    Lt[e * Bp] = ln;
                    if (Zt) Zt[e * Bp] = zz;
                }
            }
#pragma unroll
            for (int j = 0; j < n; j++) mun[j] = mu[j];
        }
        // head: z_0 = -Hi_0 (q_hat_0 + B' mu_0)  (:456-461)
#pragma unroll
        for (int j = 0; j < m; j++) {
            SPCIES_TV_ROW_BARRIER();
            double lam = first ? 0.0 : Lt[(long)j * Bp];
            double vold = first ? 0.0 : Vt[(long)j * Bp];
            double zz = q[n + j] + lam - RH(j) * vold;
#pragma unroll
            for (int i = 0; i < n; i++) zz = madd<EXACT>(zz, cAB[i * nm + n + j], mun[i]);
            zz = -cHi_0[j] * zz;
            double vn = (ELLIP || GEN) ? clamp_ref(zz + RIH(j) * lam, cLBu0[j], cUBu0[j])
                                       : clamp_ref(zz + rho_i * lam, cLB[n + j], cUB[n + j]);
            double ln = lam + RH(j) * (zz - vn);
            res = res || above(vold, vn, tol) || above(zz, vn, tol);
            Vt[(long)j * Bp] = vn;
            Lt[(long)j * Bp] = ln;
            if (Zt) Zt[(long)j * Bp] = zz;
            u0[j] = vn;
        }
        // exit condition (:624-631)
        if (!res) {
            flag = 1;
            break;
        }
        if (k >= c.k_max) {
            flag = -1;
            break;
        }
    }
#pragma unroll
    for (int j = 0; j < m; j++) u_out[t * m + j] = u0[j];
    k_out[t] = k;
    e_out[t] = flag;
}



// [rows][Bp] structure-of-arrays scratch -> [B][rows] instance-contiguous output (the layout the
// reference's DEBUG copy-out produces per instance, code_laxMPC_ADMM_C.c:657-686).
static __global__ __launch_bounds__(256) void soa_to_aos_kernel(const double *__restrict__ S, long Bp, long B, int rows,
                                                          double *__restrict__ out) {
    __shared__ double tile[64][65];
    const long b0 = (long)blockIdx.x * 64;
    const int r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
    for (int rr = ty; rr < 64; rr += 4) {
        int r = r0 + rr;
        long bb = b0 + tx;
        tile[rr][tx] = (r < rows && bb < B) ? S[(long)r * Bp + bb] : 0.0;
    }
    __syncthreads();
    for (int bbi = ty; bbi < 64; bbi += 4) {
        long bb = b0 + bbi;
        int r = r0 + tx;
        if (bb < B && r < rows) out[bb * rows + r] = tile[tx][bbi];
    }
}

}  // namespace spcies
