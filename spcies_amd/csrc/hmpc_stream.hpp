// Variant STREAM of the HMPC ADMM / SADMM solver with the (z_hat, s_hat) = (z, s) splitting, sparse KKT
// path, box constraints (formulations/+HMPC/code_HMPC_ADMM_split_C.c:102-333, snippets/proj_SOC3.c:4-35):
// ONE LANE PER INSTANCE, reference operation order, no FMA contraction -> bit-identical.
//
// Same memory design as soc_stream.hpp: the KKT factor's sparsity pattern is wave-uniform (scalar loads
// for indices / values), per-instance vectors live in a structure-of-arrays scratch [row][instance].
#pragma once
#include "admm_stream.hpp"

namespace spcies {

struct HmpcDev {
    int A, QQ, Te, Se, LB, UB, LBy, UBy, L_val, Dinv, bh;  // offsets (doubles)
    int L_col, L_row, idx_x0;                               // offsets (ints)
    int n, m, N, dim, n_s, n_eq, n_soc, nrow_M, k_max, use_soc, symmetric;
    double tol_p, tol_d, rho, rho_i, sigma, sigma_i, alpha;
    // COUPLED_CONSTRAINTS (code_HMPC_ADMM_split_C.c:65, 233-283): LBy <= E x + F u <= UBy - z is free, s = [N n_y box slacks of
    // the outputs (bounds LBy / UBy, stage after stage); cone rows]; n_y = rows of E / F (n + m with box constraints)
    int coupled, n_y;
    __host__ __device__ int cone0() const { return dim + (coupled ? N * n_y : 0); }       // first cone row
    __host__ __device__ int triples() const { return use_soc ? n_soc : (coupled ? n_y : n + m); }
};

#pragma clang fp contract(off)

// snippets/proj_SOC3.c:4-35 on three registers
__device__ __forceinline__ void proj_soc3(double &x0, double &x1, double &x2, double alpha, double d) {
    double x_norm = 0.0;
    x_norm += x1 * x1;
    x_norm += x2 * x2;
    x_norm = sqrt(x_norm);
    const double corrected = alpha * (x0 - d);
    if (x_norm <= corrected) {
    } else if (x_norm <= -corrected) {
        x0 = d;
        x1 = 0.0;
        x2 = 0.0;
    } else {
        const double step = (corrected + x_norm) / (2 * x_norm);
        x0 = step * x_norm * alpha + d;
        x1 = step * x1;
        x2 = step * x2;
    }
}

// scratch rows: PR (dim+n_s) | DU (dim+n_s) | RH (nrow_M) | BH (n_eq+n_s) | QV (dim)
__global__ __launch_bounds__(64) void hmpc_stream_kernel(HmpcDev c, const double *__restrict__ C, const int *__restrict__ I,
                                                         const double *__restrict__ x0g, const double *__restrict__ xrg,
                                                         const double *__restrict__ urg, int ref_stride, long B, long Bp,
                                                         double *__restrict__ S, double *__restrict__ u_out,
                                                         int *__restrict__ k_out, int *__restrict__ e_out) {
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= B) return;
    const int n = c.n, m = c.m, nm = n + m, N = c.N, dim = c.dim, n_s = c.n_s, n_eq = c.n_eq, nrow = c.nrow_M;
    const int np = dim + n_s, nc = n_eq + n_s;
    double *PR = S + t, *DU = PR + (long)np * Bp, *RH = DU + (long)np * Bp, *BH = RH + (long)nrow * Bp, *QV = BH + (long)nc * Bp;
#define AT(P, i) (P)[(long)(i) * Bp]
    const double *x0 = x0g + t * n;
    const double *xr = ref_stride ? xrg + t * n : xrg;
    const double *ur = ref_stride ? urg + t * m : urg;
    const double *cA = C + c.A, *cQQ = C + c.QQ, *cTe = C + c.Te, *cSe = C + c.Se, *cLB = C + c.LB, *cUB = C + c.UB,
                 *cLBy = C + c.LBy, *cUBy = C + c.UBy, *Lv = C + c.L_val, *Dinv = C + c.Dinv, *cbh = C + c.bh;
    const int *Lc = I + c.L_col, *Lr = I + c.L_row, *ix0 = I + c.idx_x0;
    // ---- setup (:102-129)
    for (int j = 0; j < np; j++) {
        AT(PR, j) = 0.0;
        AT(DU, j) = 0.0;
    }
    for (int j = 0; j < nc; j++) AT(BH, j) = cbh[j];
    for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int i = 0; i < n; i++) acc -= cA[j * n + i] * x0[i];
        AT(BH, ix0[j]) = acc;
    }
    for (int j = 0; j < dim; j++) AT(QV, j) = 0.0;
    for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int i = 0; i < n; i++) acc -= cTe[j * n + i] * xr[i] + cQQ[j * n + i] * x0[i];
        AT(QV, (N - 1) * nm + m + j) = acc;
    }
    for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int i = 0; i < n; i++) acc -= cQQ[j * n + i] * x0[i];
        AT(QV, (N - 1) * nm + 2 * n + m + j) = acc;
    }
    for (int j = 0; j < m; j++) {
        double acc = 0.0;
        for (int i = 0; i < m; i++) acc -= cSe[j * m + i] * ur[i];
        AT(QV, (N - 1) * nm + m + 3 * n + j) = acc;
    }
    const double rho = c.rho, rho_i = c.rho_i, sigma = c.sigma, sigma_i = c.sigma_i;
    const double as = c.alpha * c.sigma, ar = c.alpha * c.rho;
    const double gz = c.symmetric ? as : sigma, gs = c.symmetric ? ar : rho;  // step of the (second) dual update

    int k = 0, flag = -1;
    while (true) {
        k += 1;
        // rhs = [sigma z - q - lambda; rho s - mu; bh]  (:156-165)
        for (int j = 0; j < dim; j++) AT(RH, j) = sigma * AT(PR, j) - AT(QV, j) - AT(DU, j);
        for (int j = 0; j < n_s; j++) AT(RH, dim + j) = rho * AT(PR, dim + j) - AT(DU, dim + j);
        for (int j = 0; j < nc; j++) AT(RH, np + j) = AT(BH, j);
        // KKT solve through L D L' (:193-209)
        for (int i = 0; i < nrow; i++) {
            const double xi = AT(RH, i);
            csc_scatter(Lv, Lr, Lc[i], Lc[i + 1], xi, RH, Bp);
        }
        for (int j = 0; j < nrow; j++) AT(RH, j) *= Dinv[j];
        for (int i = nrow - 1; i >= 0; i--) AT(RH, i) = csr_dot<true>(AT(RH, i), Lv, Lr, Lc[i], Lc[i + 1], RH, Bp);
        bool res = false;
        // z (:215-238, 288-312, 318-333): half dual step (symmetric), box, dual step, residuals
        for (int j = 0; j < dim; j++) {
            const double zh = AT(RH, j), zo = AT(PR, j);
            double lam = AT(DU, j);
            if (c.symmetric) lam += as * (zh - zo);
            double z = zh + sigma_i * lam;
            if (!c.coupled && j < dim - 3 * nm) z = clamp_ref(z, cLB[j], cUB[j]);
            AT(PR, j) = z;
            AT(DU, j) = lam + gz * (zh - z);
            res = res || (fabs(zo - z) > c.tol_d) || (fabs(z - zh) > c.tol_p);
        }
        // coupled constraints: box on the output slacks (:262-269), same half step / dual step / residuals with rho
        const int cone0 = c.cone0();
        for (int j = dim; j < cone0; j++) {
            const double sh = AT(RH, j), so = AT(PR, j);
            double mu = AT(DU, j);
            if (c.symmetric) mu += ar * (sh - so);
            double sv = sh + rho_i * mu;
            sv = clamp_ref(sv, cLBy[(j - dim) % c.n_y], cUBy[(j - dim) % c.n_y]);
            AT(PR, j) = sv;
            AT(DU, j) = mu + gs * (sh - sv);
            res = res || (fabs(so - sv) > c.tol_d) || (fabs(sv - sh) > c.tol_p);
        }
        // s in triples (:241-259): diamond = two shifted cones per signal, or plain cones with use_soc
        const int triples = c.triples();
        for (int j = 0; j < triples; j++) {
            double sh[3], so[3], mu[3], s[3];
            for (int r = 0; r < 3; r++) {
                sh[r] = AT(RH, cone0 + 3 * j + r);
                so[r] = AT(PR, cone0 + 3 * j + r);
                mu[r] = AT(DU, cone0 + 3 * j + r);
                if (c.symmetric) mu[r] += ar * (sh[r] - so[r]);
                s[r] = sh[r] + rho_i * mu[r];
            }
            if (c.use_soc) {
                proj_soc3(s[0], s[1], s[2], 1.0, 0.0);
            } else {
                proj_soc3(s[0], s[1], s[2], 1.0, cLBy[j]);
                proj_soc3(s[0], s[1], s[2], -1.0, cUBy[j]);
            }
            for (int r = 0; r < 3; r++) {
                AT(PR, cone0 + 3 * j + r) = s[r];
                AT(DU, cone0 + 3 * j + r) = mu[r] + gs * (sh[r] - s[r]);
                res = res || (fabs(so[r] - s[r]) > c.tol_d) || (fabs(s[r] - sh[r]) > c.tol_p);
            }
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
