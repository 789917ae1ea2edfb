// Variant STREAM of the ellipMPC ADMM solver with the terminal ellipsoid as a second-order cone
// (formulations/+ellipMPC/code_ellipMPC_ADMM_soc_C.c:84-296): ONE LANE PER INSTANCE, reference
// operation order, no FMA contraction -> bit-identical.
//
// The sparsity pattern is the controller's, i.e. wave-uniform: CSR / CSC index arrays and values are
// read with scalar loads, every vector element access is the same row for all 64 lanes, so the
// per-instance vectors live in a structure-of-arrays scratch [row][instance] and each access is one
// coalesced 512-byte transaction.  Fully run-time sized (any n, m, N): no template parameters.
#pragma once
#include "admm_stream.hpp"

namespace spcies {

struct SocDev {
    // offsets (doubles) into the FP64 constants allocation
    int A, Q, R, T, LB, UB, PhiP, L_val, Dinv, GhHhi_val, HhiGh_val, Hhi_val;
    // offsets (ints) into the index allocation
    int L_col, L_row, GhHhi_col, GhHhi_row, HhiGh_col, HhiGh_row, Hhi_col, Hhi_row;
    int n, m, N, dim, n_s, n_eq, k_max;
    double tol_p, tol_d, rho, rho_i, sigma, sigma_i;
};

#pragma clang fp contract(off)

// Sparse row / column kernels of the STREAM variants: the reference's sums in its order, but the operands of eight terms are
// loaded before the first is used - hipcc otherwise waits for every load of such a chain (one load in flight per wavefront).
// acc (+/-)= sum_j val[j] * X[idx[j]],  j = r0 .. r1 - 1
template <bool SUB>
__device__ __forceinline__ double csr_dot(double acc, const double *__restrict__ val, const int *__restrict__ idx, int r0, int r1,
                                          const double *X, long Bp) {
    int j = r0;
    for (; j + 8 <= r1; j += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = X[(long)idx[j + u] * Bp];
#pragma unroll
        for (int u = 0; u < 8; u++) acc = SUB ? acc - val[j + u] * x[u] : acc + val[j + u] * x[u];
    }
    for (; j < r1; j++) acc = SUB ? acc - val[j] * X[(long)idx[j] * Bp] : acc + val[j] * X[(long)idx[j] * Bp];
    return acc;
}
// X[idx[j]] -= val[j] * xi,  j = r0 .. r1 - 1  (the targets of one column of L are distinct rows)
__device__ __forceinline__ void csc_scatter(const double *__restrict__ val, const int *__restrict__ idx, int r0, int r1, double xi,
                                            double *X, long Bp) {
    int j = r0;
    for (; j + 8 <= r1; j += 8) {
        double x[8];
#pragma unroll
        for (int u = 0; u < 8; u++) x[u] = X[(long)idx[j + u] * Bp];
#pragma unroll
        for (int u = 0; u < 8; u++) X[(long)idx[j + u] * Bp] = x[u] - val[j + u] * xi;
    }
    for (; j < r1; j++) X[(long)idx[j] * Bp] -= val[j] * xi;
}

// scratch rows: PR (dim+n_s) | PH (dim+n_s) | DU (dim+n_s) | QH (dim+n_s) | RH (n_eq+n_s) | BH (n_eq+n_s) | QV (dim)
__global__ __launch_bounds__(64) void soc_stream_kernel(SocDev c, const double *__restrict__ C, const int *__restrict__ I,
                                                        const double *__restrict__ x0g, const double *__restrict__ xrg,
                                                        const double *__restrict__ urg, int ref_stride,
                                                        const double *__restrict__ rg, int r_stride, long B, long Bp,
                                                        double *__restrict__ S, double *__restrict__ u_out,
                                                        int *__restrict__ k_out, int *__restrict__ e_out) {
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= B) return;
    const int n = c.n, m = c.m, nm = n + m, N = c.N, dim = c.dim, n_s = c.n_s, n_eq = c.n_eq;
    const int np = dim + n_s, nr = n_eq + n_s;
    double *PR = S + t, *PH = PR + (long)np * Bp, *DU = PH + (long)np * Bp, *QH = DU + (long)np * Bp;
    double *RH = QH + (long)np * Bp, *BH = RH + (long)nr * Bp, *QV = BH + (long)nr * Bp;
#define AT(P, i) (P)[(long)(i) * Bp]
    const double *x0 = x0g + t * n;
    const double *xr = ref_stride ? xrg + t * n : xrg;
    const double *ur = ref_stride ? urg + t * m : urg;
    const double r_ellip = rg[r_stride ? t : 0];
    const double *cA = C + c.A, *cQ = C + c.Q, *cR = C + c.R, *cT = C + c.T, *cLB = C + c.LB, *cUB = C + c.UB,
                 *cPhiP = C + c.PhiP;
    // ---- setup (:84-131): state = 0, bh, q
    for (int j = 0; j < np; j++) {
        AT(PR, j) = 0.0;
        AT(PH, j) = 0.0;
        AT(DU, j) = 0.0;
    }
    for (int j = 0; j < nr; j++) AT(BH, j) = 0.0;
    for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int i = 0; i < n; i++) acc -= cA[j * n + i] * x0[i];
        AT(BH, j) = acc;
    }
    AT(BH, n_eq - 1) = r_ellip;
    for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int i = 0; i < n; i++) acc -= cPhiP[j * n + i] * xr[i];
        AT(BH, n_eq + 1 + j) = acc;
    }
    for (int j = 0; j < dim; j++) AT(QV, j) = 0.0;
    for (int j = 0; j < m; j++) {
        double acc = 0.0;
        for (int i = 0; i < m; i++) acc += cR[j * m + i] * ur[i];
        AT(QV, j) = acc;
    }
    for (int kb = 0; kb < N - 1; kb++) {
        for (int j = 0; j < n; j++) {
            double acc = 0.0;
            for (int i = 0; i < n; i++) acc += cQ[j * n + i] * xr[i];
            AT(QV, m + kb * nm + j) = acc;
        }
        for (int j = 0; j < m; j++) {
            double acc = 0.0;
            for (int i = 0; i < m; i++) acc += cR[j * m + i] * ur[i];
            AT(QV, nm + kb * nm + j) = acc;
        }
    }
    for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int i = 0; i < n; i++) acc += cT[j * n + i] * xr[i];
        AT(QV, m + (N - 1) * nm + j) = acc;
    }
    const double *Lv = C + c.L_val, *Dinv = C + c.Dinv, *Gv = C + c.GhHhi_val, *HGv = C + c.HhiGh_val, *Hv = C + c.Hhi_val;
    const int *Lc = I + c.L_col, *Lr = I + c.L_row, *Gc = I + c.GhHhi_col, *Gr = I + c.GhHhi_row, *HGc = I + c.HhiGh_col,
              *HGr = I + c.HhiGh_row, *Hc = I + c.Hhi_col, *Hr = I + c.Hhi_row;
    const double rho = c.rho, rho_i = c.rho_i, sigma = c.sigma, sigma_i = c.sigma_i;

    int k = 0, flag = -1;
    while (true) {
        k += 1;
        // q_hat = [q + lambda - sigma z; mu - rho s]  (:144-149)
        for (int j = 0; j < dim; j++) AT(QH, j) = AT(QV, j) + AT(DU, j) - sigma * AT(PR, j);
        for (int j = 0; j < n_s; j++) AT(QH, dim + j) = AT(DU, dim + j) - rho * AT(PR, dim + j);
        // rhs = (-Gh Hh^-1) q_hat - bh  (:152-160)
        for (int i = 0; i < nr; i++) {
            const double acc = csr_dot<false>(0.0, Gv, Gc, Gr[i], Gr[i + 1], QH, Bp);
            AT(RH, i) = acc - AT(BH, i);
        }
        // W mu = rhs through L D L' (:166-188)
        for (int i = 0; i < nr; i++) {
            const double xi = AT(RH, i);
            csc_scatter(Lv, Lr, Lc[i], Lc[i + 1], xi, RH, Bp);
        }
        for (int j = 0; j < nr; j++) AT(RH, j) *= Dinv[j];
        for (int i = nr - 1; i >= 0; i--) AT(RH, i) = csr_dot<true>(AT(RH, i), Lv, Lr, Lc[i], Lc[i + 1], RH, Bp);
        // primal_hat = (-Hh^-1) q_hat + (-Hh^-1 Gh') mu  (:193-205)
        for (int i = 0; i < np; i++) {
            double acc = csr_dot<false>(0.0, Hv, Hc, Hr[i], Hr[i + 1], QH, Bp);
            acc = csr_dot<false>(acc, HGv, HGc, HGr[i], HGr[i + 1], RH, Bp);
            AT(PH, i) = acc;
        }
        // z: box on the first dim-n-1 entries (:209-217), lambda (:246-248), residuals (:256-267)
        bool res = false;
        for (int j = 0; j < dim; j++) {
            const double zh = AT(PH, j), lam = AT(DU, j), zo = AT(PR, j);
            double z = zh + sigma_i * lam;
            if (j < dim - n - 1) z = clamp_ref(z, cLB[j], cUB[j]);
            AT(PR, j) = z;
            AT(DU, j) = lam + sigma * (zh - z);
            res = res || (fabs(zo - z) > c.tol_d) || (fabs(z - zh) > c.tol_p);
        }
        // s: SOC projection (:220-242), mu (:251-253)
        double s_norm = 0.0, s0 = 0.0;
        for (int j = 0; j < n_s; j++) {
            const double v = AT(PH, dim + j) + rho_i * AT(DU, dim + j);
            AT(QH, dim + j) = v;  // q_hat's tail is free now: park the un-projected s there
            if (j == 0) s0 = v;
            else s_norm += v * v;
        }
        s_norm = sqrt(s_norm);
        for (int j = 0; j < n_s; j++) {
            double v = AT(QH, dim + j);
            if (s_norm <= s0) {
            } else if (s_norm <= -s0) {
                v = 0.0;
            } else {
                const double step = (s0 + s_norm) / (2 * s_norm);
                v = (j == 0) ? step * s_norm : step * v;
            }
            const double so = AT(PR, dim + j), sh = AT(PH, dim + j), mu = AT(DU, dim + j);
            AT(PR, dim + j) = v;
            AT(DU, dim + j) = mu + rho * (sh - v);
            res = res || (fabs(so - v) > c.tol_d) || (fabs(v - sh) > c.tol_p);
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
#undef AT
    for (int j = 0; j < m; j++) u_out[t * m + j] = PR[(long)j * Bp];
    k_out[t] = k;
    e_out[t] = flag;
}

}  // namespace spcies
