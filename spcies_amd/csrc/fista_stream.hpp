// Variant STREAM of the banded-Cholesky FISTA solver (laxMPC / equMPC): ONE LANE PER INSTANCE, the
// reference's operation order (formulations/+laxMPC/code_laxMPC_FISTA_C.c:296-389 and helpers :471-651,
// equMPC: code_equMPC_FISTA_C.c), no FMA contraction with EXACT = true -> bit-identical results.
//
// One iteration is two sweeps over the N blocks:
//   forward : z(y) block by block (:471-539), residual r = b - G z (:546-574), exit test on |r|
//             (:330-344) and - speculatively, it only writes scratch - the forward substitution of
//             solve_W (:582-612);
//   backward: back substitution (:614-649), lambda = y + d, y = lambda + (t1-1)(lambda - lambda1)/t (:360-385).
// State streamed through the structure-of-arrays scratch [row][instance]: y, lambda, the forward-
// substituted d (N*n rows each) - (4 reads + 3 writes) * N*n * 8 B = 10 KB per iteration at C2.
#pragma once
#include "admm_stream.hpp"

namespace spcies {

struct FistaDev {
    int AB, Alpha, Beta, Q, R, QRi, T, Ti, LB, UB;  // offsets (doubles) into the constants allocation
    int N, k_max;
    double tol;
};

#pragma clang fp contract(off)


// TV: TIME_VARYING == 1 (code_laxMPC_FISTA_C.c:18, 42-56, 83-262): AB, Alpha, Beta, Q, R, QRi, LB, UB are this instance's own rows
// of the scratch TVS (written by fista_tv_update_kernel), read through a buffer resource; T, Ti stay controller constants.
template <int n, int m, bool TERMINAL, bool EXACT, bool TV = false>
__global__ __launch_bounds__(64) void fista_stream_kernel(FistaDev c, const double *__restrict__ C,
                                                          const double *__restrict__ x0g,
                                                          const double *__restrict__ xrg,
                                                          const double *__restrict__ urg, int ref_stride, long B,
                                                          long Bp, double *__restrict__ Y, double *__restrict__ LAM,
                                                          double *__restrict__ DL, double *__restrict__ ZS,
                                                          double *__restrict__ u_out, int *__restrict__ k_out,
                                                          int *__restrict__ e_out, const double *__restrict__ TVS = nullptr) {
    constexpr int nm = n + m;
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= B) return;
    const int N = c.N;
    const double tol = c.tol;
    const FistaTvLayout tl = fista_tv_layout(n, m, N);
    auto K = [&](int shared_off, int tv_row) {
        if constexpr (TV) {
            return KArr<true>{__builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(TVS), 0, -1, 0x00020000), (unsigned)tv_row,
                              (unsigned)(Bp * 8), (unsigned)(t * 8)};
        } else {
            return KArr<false>{C + shared_off};
        }
    };
    const KArr<TV> cAB = K(c.AB, tl.AB), cAlpha = K(c.Alpha, tl.Alpha), cBeta = K(c.Beta, tl.Beta), cQ = K(c.Q, tl.Q), cR = K(c.R, tl.R),
                   cQRi = K(c.QRi, tl.QRi), cLB = K(c.LB, tl.LB), cUB = K(c.UB, tl.UB);
    const double *cT = C + c.T, *cTi = C + c.Ti;

    // ---- per-instance setup (code_laxMPC_FISTA_C.c:274-289)
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
            qT[j] = TERMINAL ? cT[j] * xr[j] : 0.0;
        }
#pragma unroll
        for (int j = 0; j < m; j++) q[n + j] = cR[j] * urp[j];
    }
    double *Yt = Y + t, *Lt = LAM + t, *Dt = DL + t;
    double *Zt = ZS ? ZS + t : nullptr;
    const long off_mid = (long)m, off_tail = (long)m + (long)(N - 1) * nm;

    int k = 0, flag = -1;
    double tk = 1.0, tk1 = 1.0;
    double u0[m];
    bool init = true;  // the initial step (:296-318): y = lambda = 0, no exit test, no momentum
    while (true) {
        if (!init) {
            k += 1;
            tk1 = tk;
        }
        // ================= forward sweep =================
        bool res = false;
        double yc[n], yn[n];  // y_l, y_{l+1}
#pragma unroll
        for (int i = 0; i < n; i++) yc[i] = init ? 0.0 : Yt[(long)i * Bp];
        // z_0 (:474-491)
#pragma unroll
        for (int j = 0; j < m; j++) {
            double acc = q[n + j];
#pragma unroll
            for (int i = 0; i < n; i++) acc = msub<EXACT>(acc, cAB[i * nm + n + j], yc[i]);
            acc = acc * cQRi[n + j];
            u0[j] = clamp_ref(acc, cLB[n + j], cUB[n + j]);
            if (Zt) Zt[(long)j * Bp] = u0[j];
        }
        double zp[nm], dp[n];  // z of the previous block, forward-substituted d of the previous block
        for (int l = 0; l < N; l++) {
            const bool last = (l == N - 1);
            double zc[nm];
            if (!last) {
#pragma unroll
                for (int i = 0; i < n; i++) yn[i] = init ? 0.0 : Yt[((long)(l + 1) * n + i) * Bp];
                // z[l] (:494-520)
#pragma unroll
                for (int j = 0; j < nm; j++) {
                    double acc = q[j];
#pragma unroll
                    for (int i = 0; i < n; i++) acc = msub<EXACT>(acc, cAB[i * nm + j], yn[i]);
                    if (j < n) acc = acc + yc[j];
                    acc = acc * cQRi[j];
                    zc[j] = clamp_ref(acc, cLB[j], cUB[j]);
                    if (Zt) Zt[(off_mid + (long)l * nm + j) * Bp] = zc[j];
                }
            } else if constexpr (TERMINAL) {
                // z_N (:523-537)
#pragma unroll
                for (int j = 0; j < n; j++) {
                    double acc = qT[j] + yc[j];
                    acc = acc * cTi[j];
                    zc[j] = clamp_ref(acc, cLB[j], cUB[j]);
                    if (Zt) Zt[(off_tail + j) * Bp] = zc[j];
                }
            }
            // residual block l (:546-574), exit test (:330-344), forward substitution (:582-612)
            double d[n];
#pragma unroll
            for (int j = 0; j < n; j++) {
                double acc;
                if (l == 0) {
                    acc = b[j] + zc[j];
#pragma unroll
                    for (int i = 0; i < m; i++) acc = msub<EXACT>(acc, cAB[j * nm + n + i], u0[i]);
                } else {
                    if (!last) acc = zc[j];
                    else acc = TERMINAL ? zc[j] : xr[j];
#pragma unroll
                    for (int i = 0; i < nm; i++) acc = msub<EXACT>(acc, cAB[j * nm + i], zp[i]);
                }
                double a = (acc > 0.0) ? acc : -acc;
                res = res || (a > tol);
                d[j] = acc;
            }
            const KArr<TV> Bl = cBeta + (long)l * n * n;
            const KArr<TV> Al = cAlpha + (long)(l - 1) * n * n;
#pragma unroll
            for (int j = 0; j < n; j++) {
                if constexpr (TV) asm volatile("" ::: "memory");  // keeps a whole sweep's per-lane loads from being hoisted (and spilled)
                double acc = d[j];
                if (l > 0) {
#pragma unroll
                    for (int i = 0; i < n; i++) acc = msub<EXACT>(acc, Al[i * n + j], dp[i]);
                }
#pragma unroll
                for (int i = 0; i < j; i++) acc = msub<EXACT>(acc, Bl[i * n + j], d[i]);
                d[j] = Bl[j * n + j] * acc;
            }
#pragma unroll
            for (int j = 0; j < n; j++) {
                Dt[((long)l * n + j) * Bp] = d[j];
                dp[j] = d[j];
            }
#pragma unroll
            for (int j = 0; j < nm; j++) zp[j] = zc[j];
#pragma unroll
            for (int j = 0; j < n; j++) yc[j] = yn[j];
        }
        // ================= exit (:346-353) =================
        if (!init) {
            if (!res) {
                flag = 1;
                break;
            }
            if (k >= c.k_max) {
                flag = -1;
                break;
            }
            tk = 0.5 * (1 + sqrt(1 + 4 * tk1 * tk1));
        }
        // ================= backward sweep: d = W^-1 r, lambda, y (:357-385) =================
        double dn[n];
        for (int l = N - 1; l >= 0; l--) {
            const KArr<TV> Bl = cBeta + (long)l * n * n;
            const KArr<TV> Al = cAlpha + (long)l * n * n;
            double d[n];
#pragma unroll
            for (int j = 0; j < n; j++) d[j] = (l == N - 1) ? dp[j] : Dt[((long)l * n + j) * Bp];
#pragma unroll
            for (int j = n - 1; j >= 0; j--) {
                if constexpr (TV) asm volatile("" ::: "memory");
                double acc = d[j];
                if (l < N - 1) {
#pragma unroll
                    for (int i = n - 1; i >= 0; i--) acc = msub<EXACT>(acc, Al[j * n + i], dn[i]);
                }
#pragma unroll
                for (int i = n - 1; i > j; i--) acc = msub<EXACT>(acc, Bl[j * n + i], d[i]);
                d[j] = Bl[j * n + j] * acc;
            }
#pragma unroll
            for (int j = 0; j < n; j++) {
                const long e = ((long)l * n + j) * Bp;
                const double yo = init ? 0.0 : Yt[e];
                const double l1 = init ? 0.0 : Lt[e];
                const double ln = yo + d[j];
                Lt[e] = ln;
                Yt[e] = init ? ln : ln + (tk1 - 1) * (ln - l1) / tk;
                dn[j] = d[j];
            }
        }
        init = false;
    }
#pragma unroll
    for (int j = 0; j < m; j++) u_out[t * m + j] = u0[j];
    k_out[t] = k;
    e_out[t] = flag;
}


}  // namespace spcies
