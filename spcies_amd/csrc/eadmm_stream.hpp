// Variant STREAM of the MPCT EADMM solver (diagonal Q, R): ONE LANE PER INSTANCE, the reference's
// operation order (formulations/+MPCT/code_MPCT_EADMM_C.c:85-457), no FMA contraction -> bit-identical.
//
// One iteration = P1 (z1, clamp) + P2 (z2 = W2 q2) in one pass over the stages, then the P3 banded
// solve as a forward and a backward sweep; the backward sweep also forms z3, the residual, the dual
// update and the three-part exit test.  State streamed through the SoA scratch [row][instance]:
// z1, z3 ((N+1)(n+m) rows), lambda ((N+3)(n+m) rows), forward-substituted mu (N n rows); z2 stays in
// registers.
#pragma once
#include "admm_stream.hpp"

namespace spcies {

struct EadmmDev {
    int rho, rho_0, rho_s, LB, UB, LB_0, UB_0, LB_s, UB_s, AB, T, S, Alpha, Beta, H1i, W2, H3i;  // offsets (doubles)
    int N, k_max;
    double tol;
};

#pragma clang fp contract(off)

template <int n, int m>
__global__ __launch_bounds__(64) void eadmm_stream_kernel(EadmmDev c, const double *__restrict__ C,
                                                          const double *__restrict__ x0g,
                                                          const double *__restrict__ xrg,
                                                          const double *__restrict__ urg, int ref_stride, long B,
                                                          long Bp, double *__restrict__ Z1, double *__restrict__ Z3,
                                                          double *__restrict__ LAM, double *__restrict__ MUs,
                                                          double *__restrict__ z2_out, double *__restrict__ u_out,
                                                          int *__restrict__ k_out, int *__restrict__ e_out) {
    constexpr int nm = n + m;
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= B) return;
    const int N = c.N;
    const double tol = c.tol;
    const double *cRho = C + c.rho, *cRho0 = C + c.rho_0, *cRhos = C + c.rho_s, *cLB = C + c.LB, *cUB = C + c.UB,
                 *cLB0 = C + c.LB_0, *cUB0 = C + c.UB_0, *cLBs = C + c.LB_s, *cUBs = C + c.UB_s, *cAB = C + c.AB,
                 *cT = C + c.T, *cS = C + c.S, *cAlpha = C + c.Alpha, *cBeta = C + c.Beta, *cH1i = C + c.H1i,
                 *cW2 = C + c.W2, *cH3i = C + c.H3i;
    double x0[n], Txr[n], Sur[m];  // T xr and S ur enter q2 one product at a time (:128-137): keep xr, ur instead
    double xr[n], ur[m];
    {
        const double *xrp = ref_stride ? xrg + t * n : xrg;
        const double *urp = ref_stride ? urg + t * m : urg;
#pragma unroll
        for (int i = 0; i < n; i++) {
            x0[i] = x0g[t * n + i];
            xr[i] = xrp[i];
        }
#pragma unroll
        for (int i = 0; i < m; i++) ur[i] = urp[i];
        (void)Txr;
        (void)Sur;
    }
    double *Z1t = Z1 + t, *Z3t = Z3 + t, *Lt = LAM + t, *Mt = MUs + t;
#define EL(l, j) (((long)(l) * nm + (j)) * Bp)
    double z2[nm];
#pragma unroll
    for (int j = 0; j < nm; j++) z2[j] = 0.0;

    int k = 0, flag = -1;
    double u0[m];
    while (true) {
        k += 1;
        const bool first = (k == 1);  // z1 = z3 = lambda = 0: skip the scratch reads
        double z2p[nm];
#pragma unroll
        for (int j = 0; j < nm; j++) z2p[j] = z2[j];
        // ============ P1 (z1, :97-117) and the q2 accumulation of P2 (:123-143) ============
        double q2[nm];
        {  // stage N first: q2 starts from its terms
            double z1N[nm];
#pragma unroll
            for (int j = 0; j < nm; j++) {
                const double z3v = first ? 0.0 : Z3t[EL(N, j)];
                const double l1 = first ? 0.0 : Lt[EL(N + 1, j)], l2 = first ? 0.0 : Lt[EL(N + 2, j)];
                const double r = cRho[N * nm + j], rs = cRhos[j];
                double v = (r * z3v + (r + rs) * z2[j] + l1 + l2) * cH1i[N * nm + j];
                v = clamp_ref(v, cLBs[j], cUBs[j]);
                z1N[j] = v;
                Z1t[EL(N, j)] = v;
                q2[j] = r * z3v - (r + rs) * v + l1 + l2;
            }
#pragma unroll
            for (int j = 0; j < n; j++) {
#pragma unroll
                for (int i = 0; i < n; i++) q2[j] = q2[j] + cT[j * n + i] * xr[i];
            }
#pragma unroll
            for (int j = 0; j < m; j++) {
#pragma unroll
                for (int i = 0; i < m; i++) q2[j + n] = q2[j + n] + cS[j * m + i] * ur[i];
            }
        }
        for (int l = 0; l < N; l++) {
#pragma unroll
            for (int j = 0; j < nm; j++) {
                const double z3v = first ? 0.0 : Z3t[EL(l, j)];
                const double l1 = first ? 0.0 : Lt[EL(l + 1, j)];
                const double r = cRho[l * nm + j];
                double v;
                if (l == 0) {
                    const double l0 = first ? 0.0 : Lt[EL(0, j)];
                    const double x0j = (j < n) ? x0[j < n ? j : 0] : 0.0;
                    v = (r * (z3v + z2[j]) + cRho0[j] * x0j + l1 - l0) * cH1i[j];
                    v = clamp_ref(v, cLB0[j], cUB0[j]);
                } else {
                    v = (r * (z3v + z2[j]) + l1) * cH1i[l * nm + j];
                    v = clamp_ref(v, cLB[j], cUB[j]);
                }
                Z1t[EL(l, j)] = v;
                if (l == 0 && j >= n) u0[j >= n ? j - n : 0] = v;
                q2[j] = q2[j] + r * (z3v - v) + l1;
            }
        }
        // z2 = W2 q2 (:145-151)
#pragma unroll
        for (int j = 0; j < nm; j++) {
            double acc = 0;
#pragma unroll
            for (int i = 0; i < nm; i++) acc = acc + cW2[j * nm + i] * q2[i];
            z2[j] = acc;
        }
        // ============ P3 forward: q3, right-hand side (:157-183), forward substitution (:221-249) ============
        double q3c[nm], q3n[nm], yp[n];
#pragma unroll
        for (int j = 0; j < nm; j++) {
            const double l1 = first ? 0.0 : Lt[EL(1, j)];
            q3c[j] = cRho[j] * (z2[j] - Z1t[EL(0, j)]) + l1;
        }
        for (int l = 0; l < N; l++) {
#pragma unroll
            for (int j = 0; j < nm; j++) {
                const double l1 = first ? 0.0 : Lt[EL(l + 2, j)];
                q3n[j] = cRho[(l + 1) * nm + j] * (z2[j] - Z1t[EL(l + 1, j)]) + l1;
            }
            double y[n];
#pragma unroll
            for (int j = 0; j < n; j++) {
                double acc = cH3i[(l + 1) * nm + j] * q3n[j];
#pragma unroll
                for (int i = 0; i < nm; i++) acc = acc - cAB[j * nm + i] * cH3i[l * nm + i] * q3c[i];
                y[j] = acc;
            }
            const double *Bl = cBeta + (long)l * n * n;
            const double *Al = cAlpha + (long)(l - 1) * n * n;
#pragma unroll
            for (int j = 0; j < n; j++) {
                double acc = y[j];
                if (l > 0) {
#pragma unroll
                    for (int i = 0; i < n; i++) acc = acc - Al[i * n + j] * yp[i];
                }
#pragma unroll
                for (int i = 0; i < j; i++) acc = acc - Bl[i * n + j] * y[i];
                y[j] = Bl[j * n + j] * acc;
            }
#pragma unroll
            for (int j = 0; j < n; j++) {
                Mt[((long)l * n + j) * Bp] = y[j];
                yp[j] = y[j];
            }
#pragma unroll
            for (int j = 0; j < nm; j++) q3c[j] = q3n[j];
        }
        // ============ P3 backward (:253-285), z3 (:289-320), residual + lambda (:371-402), exit (:408-449) ============
        bool res = false;
#pragma unroll
        for (int j = 0; j < nm; j++) res = res || (fabs(z2p[j] - z2[j]) > tol);
        // one stage: z3_l from q3_l (recomputed), mu_{l-1} (x rows) and mu_l; then res_{l+1}, lambda_{l+1}
        auto finish_stage = [&](int l, const double *mu_lm1, const double *mu_l) {
#pragma unroll
            for (int j = 0; j < nm; j++) {
                const double lam = first ? 0.0 : Lt[EL(l + 1, j)];
                const double z1v = Z1t[EL(l, j)];
                double v = cRho[l * nm + j] * (z2[j] - z1v) + lam;  // q3
                if (mu_lm1 && j < n) v = v - mu_lm1[j < n ? j : 0];
                if (mu_l) {
#pragma unroll
                    for (int i = 0; i < n; i++) v = v + cAB[i * nm + j] * mu_l[i];
                }
                v = -cH3i[l * nm + j] * v;
                const double z3o = first ? 0.0 : Z3t[EL(l, j)];
                Z3t[EL(l, j)] = v;
                const double r = z2[j] + v - z1v;
                Lt[EL(l + 1, j)] = lam + cRho[l * nm + j] * r;
                res = res || (fabs(r) > tol) || (fabs(z3o - v) > tol);
            }
        };
        double mun[n];
        for (int l = N - 1; l >= 0; l--) {
            const double *Bl = cBeta + (long)l * n * n;
            const double *Al = cAlpha + (long)l * n * n;
            double mu[n];
#pragma unroll
            for (int j = 0; j < n; j++) mu[j] = (l == N - 1) ? yp[j] : Mt[((long)l * n + j) * Bp];
#pragma unroll
            for (int j = n - 1; j >= 0; j--) {
                double acc = mu[j];
                if (l < N - 1) {
#pragma unroll
                    for (int i = n - 1; i >= 0; i--) acc = acc - Al[j * n + i] * mun[i];
                }
#pragma unroll
                for (int i = n - 1; i > j; i--) acc = acc - Bl[j * n + i] * mu[i];
                mu[j] = Bl[j * n + j] * acc;
            }
            if (l == N - 1) finish_stage(N, mu, nullptr);   // z3_N = -H3i (q3_N - mu_{N-1})
            else finish_stage(l + 1, mu, mun);              // z3_{l+1}: - mu_l (x rows) + AB' mu_{l+1}
#pragma unroll
            for (int j = 0; j < n; j++) mun[j] = mu[j];
        }
        finish_stage(0, nullptr, mun);                      // z3_0 = -H3i (q3_0 + AB' mu_0)
        // first and last residual rows (:374-376, 386-388) and their multipliers (:391-393, 403-405)
#pragma unroll
        for (int j = 0; j < n; j++) {
            const double r = Z1t[EL(0, j)] - x0[j];
            const double l0 = first ? 0.0 : Lt[EL(0, j)];
            Lt[EL(0, j)] = l0 + cRho0[j] * r;
            res = res || (fabs(r) > tol);
        }
#pragma unroll
        for (int j = 0; j < nm; j++) {
            const double r = z2[j] - Z1t[EL(N, j)];
            const double l2 = first ? 0.0 : Lt[EL(N + 2, j)];
            Lt[EL(N + 2, j)] = l2 + cRhos[j] * r;
            res = res || (fabs(r) > tol);
            if (first && j >= n) Lt[EL(0, j)] = 0.0;  // rows n.. of lambda[0] are never touched by the solver: keep them 0
        }
        if (!res) {
            flag = 1;
            break;
        }
        if (k >= c.k_max) {
            flag = -1;
            break;
        }
    }
#undef EL
#pragma unroll
    for (int j = 0; j < m; j++) u_out[t * m + j] = u0[j];
    k_out[t] = k;
    e_out[t] = flag;
    if (z2_out) {
#pragma unroll
        for (int j = 0; j < nm; j++) z2_out[t * nm + j] = z2[j];
    }
}

// lambda copy-out with the reference's packing (code_MPCT_EADMM_C.c:495-513): the first n entries of every
// (n+m)-wide row, written contiguously; the rest of the (N+3)(n+m) buffer stays zero.
__global__ __launch_bounds__(256) void eadmm_pack_lambda_kernel(const double *__restrict__ LAM, long Bp, long B, int N,
                                                                 int n, int nm, double *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long rows = (long)(N + 3) * nm;
    if (i >= B * rows) return;
    const long b = i / rows;
    const int e = (int)(i % rows);
    double v = 0.0;
    if (e < (N + 3) * n) {
        const int l = e / n, j = e % n;
        v = LAM[((long)l * nm + j) * Bp + b];
    }
    out[i] = v;
}

}  // namespace spcies
