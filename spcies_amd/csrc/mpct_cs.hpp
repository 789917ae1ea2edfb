// MPCT ADMM on the extended state space ('cs' submethod; formulations/+MPCT/code_MPCT_ADMM_cs_C.c:18-248): the sparse
// solver family of soc_stream.hpp / sparse_tile.hpp without the (z, s) split - q_hat, one CSR product, the L D L' solve
// of W = Aeq Hhat^-1 Aeq', two CSR products, clamp, dual step, residuals.  Two variants:
//   STREAM: one lane per instance, reference operation order, no FMA contraction -> bit-identical;
//   TILE  : LPI lanes per instance, right-hand side and q_hat in LDS, host-built step streams (sparse_tile.hpp) -> 1e-10.
// Scalar or vector rho (cons_MPCT_ADMM_cs_C.m:76-83).  Record: z, v, lambda [2 N (n+m)] (header_MPCT_ADMM_cs_C.h:14-22).
#pragma once
#include "sparse_tile.hpp"

namespace spcies {

struct CsDev {
    // offsets (doubles) into the FP64 constants allocation
    int Tz, Sz, LB, UB, L_val, Dinv, AHi_val, HiA_val, Hi_val, rho_v, rho_i_v;
    // offsets (ints) into the index allocation
    int L_col, L_row, AHi_col, AHi_row, HiA_col, HiA_row, Hi_col, Hi_row;
    int n, m, N, dim, nrow, k_max, scalar_rho;
    double tol, rho, rho_i;
};

#pragma clang fp contract(off)

// scratch rows: Z (dim) | V (dim) | LAM (dim) | QH (dim) | MU (nrow) | QV (dnm)
__global__ __launch_bounds__(64) void cs_stream_kernel(CsDev c, const double *__restrict__ C, const int *__restrict__ I,
                                                       const double *__restrict__ x0g, const double *__restrict__ xrg,
                                                       const double *__restrict__ urg, int ref_stride, long B, long Bp,
                                                       double *__restrict__ S, double *__restrict__ u_out,
                                                       int *__restrict__ k_out, int *__restrict__ e_out) {
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= B) return;
    const int n = c.n, m = c.m, dnm = 2 * (n + m), dim = c.dim, nrow = c.nrow;
    double *Z = S + t, *V = Z + (long)dim * Bp, *LAM = V + (long)dim * Bp, *QH = LAM + (long)dim * Bp, *MU = QH + (long)dim * Bp,
           *QV = MU + (long)nrow * Bp;
#define AT(P, i) (P)[(long)(i) * Bp]
    const double *x0 = x0g + t * n;
    const double *xr = ref_stride ? xrg + t * n : xrg;
    const double *ur = ref_stride ? urg + t * m : urg;
    const double *cTz = C + c.Tz, *cSz = C + c.Sz, *cLB = C + c.LB, *cUB = C + c.UB, *cRho = C + c.rho_v, *cRhoi = C + c.rho_i_v;
    // ---- setup (:40-85): state = 0, q
    for (int j = 0; j < dim; j++) {
        AT(Z, j) = 0.0;
        AT(V, j) = 0.0;
        AT(LAM, j) = 0.0;
    }
    for (int j = 0; j < dnm; j++) AT(QV, j) = 0.0;
    for (int j = 0; j < n; j++) {
        double acc = 0.0;
        for (int i = 0; i < n; i++) acc += cTz[j * n + i] * xr[i];
        AT(QV, j + n) = acc;
    }
    for (int j = 0; j < m; j++) {
        double acc = 0.0;
        for (int i = 0; i < m; i++) acc += cSz[j * m + i] * ur[i];
        AT(QV, j + 2 * n + m) = acc;
    }
    const double *Lv = C + c.L_val, *Dinv = C + c.Dinv, *Av = C + c.AHi_val, *HAv = C + c.HiA_val, *Hv = C + c.Hi_val;
    const int *Lc = I + c.L_col, *Lr = I + c.L_row, *Ac = I + c.AHi_col, *Ar = I + c.AHi_row, *HAc = I + c.HiA_col,
              *HAr = I + c.HiA_row, *Hc = I + c.Hi_col, *Hr = I + c.Hi_row;
    const bool sr = c.scalar_rho != 0;
    const double rho = c.rho, rho_i = c.rho_i;

    int k = 0, flag = -1;
    while (true) {
        k += 1;
        // q_hat = q + lambda - rho v  (:103-109)
        for (int j = 0; j < dim; j++) AT(QH, j) = AT(QV, j % dnm) + AT(LAM, j) - (sr ? rho : cRho[j]) * AT(V, j);
        // rhs = (-Aeq Hhat^-1) q_hat - b  (:113-121)
        for (int i = 0; i < nrow; i++) {
            AT(MU, i) = csr_dot<false>(0.0, Av, Ac, Ar[i], Ar[i + 1], QH, Bp);
        }
        for (int j = 0; j < n; j++) AT(MU, j) -= x0[j];
        // W mu = rhs through L D L' (:126-146)
        for (int i = 0; i < nrow; i++) {
            const double xi = AT(MU, i);
            csc_scatter(Lv, Lr, Lc[i], Lc[i + 1], xi, MU, Bp);
        }
        for (int j = 0; j < nrow; j++) AT(MU, j) *= Dinv[j];
        for (int i = nrow - 1; i >= 0; i--) AT(MU, i) = csr_dot<true>(AT(MU, i), Lv, Lr, Lc[i], Lc[i + 1], MU, Bp);
        // z = (-Hhat^-1) q_hat + (-Hhat^-1 Aeq') mu (:152-164), v (:168-177), lambda (:181-188), residuals (:192-207)
        bool res = false;
        for (int i = 0; i < dim; i++) {
            double z = csr_dot<false>(0.0, Hv, Hc, Hr[i], Hr[i + 1], QH, Bp);
            z = csr_dot<false>(z, HAv, HAc, HAr[i], HAr[i + 1], MU, Bp);
            const double lam = AT(LAM, i), v1 = AT(V, i);
            double v = z + (sr ? rho_i : cRhoi[i]) * lam;
            v = clamp_ref(v, cLB[i], cUB[i]);
            AT(Z, i) = z;
            AT(V, i) = v;
            AT(LAM, i) = lam + (sr ? rho : cRho[i]) * (z - v);
            res = res || (fabs(v1 - v) > c.tol) || (fabs(z - v) > c.tol);
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
    for (int j = 0; j < m; j++) u_out[t * m + j] = AT(V, 2 * n + j);  // (:229-231)
#undef AT
    k_out[t] = k;
    e_out[t] = flag;
}

namespace tile {

#pragma clang fp contract(fast)

// Every global row is owned by lane group row % LPI.
// scratch rows per tile: V (dim) | LAM (dim) | Z (dim) | BH (n) | QV (dnm);  LDS rows: RH (nrow) | QH (dim) | PL (dim)
template <int LPI>
__global__ __launch_bounds__(64 * WAVES) void cs_tile_kernel(CsDev c, TileDev td, const double *__restrict__ C,
                                                             const int4 *__restrict__ recs, const double *__restrict__ x0g,
                                                             const double *__restrict__ xrg, const double *__restrict__ urg,
                                                             int ref_stride, long B, double *__restrict__ S,
                                                             int *__restrict__ k_out, int *__restrict__ e_out) {
    constexpr int T = 64 / LPI, U = 4;
    extern __shared__ __attribute__((aligned(16))) double lds_wg[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane / T, cc = lane % T;
    double *lds = lds_wg + (size_t)wave * td.lds_bytes / sizeof(double);
    const long tile = (long)blockIdx.x * WAVES + wave;
    const long t = tile * T + cc;
    const bool valid = t < B;
    const int n = c.n, m = c.m, dnm = 2 * (n + m), dim = c.dim, nrow = c.nrow;
    double *RH = lds, *QH = RH + nrow * T, *PL = QH + dim * T;
    const long rows_per_tile = 3L * dim + n + dnm;
    double *V = S + tile * rows_per_tile * T + cc, *LAM = V + (long)dim * T, *Z = LAM + (long)dim * T, *BH = Z + (long)dim * T,
           *QV = BH + (long)n * T;
#define AT(P, i) (P)[(i) * T]
    const long ti = valid ? t : 0;  // out-of-range lanes compute on instance 0's inputs and never store results
    const double *x0 = x0g + ti * n;
    const double *xr = ref_stride ? xrg + ti * n : xrg;
    const double *ur = ref_stride ? urg + ti * m : urg;
    const double *cTz = C + c.Tz, *cSz = C + c.Sz, *cLB = C + c.LB, *cUB = C + c.UB, *cRho = C + c.rho_v, *cRhoi = C + c.rho_i_v,
                 *Dinv = C + c.Dinv;
    const bool sr = c.scalar_rho != 0;
    for (int j = g; j < dim; j += LPI) {
        AT(V, j) = 0.0;
        AT(LAM, j) = 0.0;
    }
    for (int j = g; j < n; j += LPI) AT(BH, j) = x0[j];
    for (int j = g; j < dnm; j += LPI) {
        double v = 0.0;
        if (j >= n && j < 2 * n) {
            for (int i = 0; i < n; i++) v += cTz[(j - n) * n + i] * xr[i];
        } else if (j >= 2 * n + m) {
            for (int i = 0; i < m; i++) v += cSz[(j - 2 * n - m) * m + i] * ur[i];
        }
        AT(QV, j) = v;
    }
    // The setup above writes per-instance rows (q, b) that OTHER lane groups of the instance read in the first iteration: without
    // this fence such a read can overtake the store and pick up whatever the scratch allocation held (a NaN there ends the solve
    // at k = 2 with flag 1: clamp(NaN) is a bound, NaN > tol is false) - seen once in a full test run, never in isolation
    __syncthreads();
    int k = 0;
    bool active = valid;
    while (true) {
        k += 1;
        // q_hat = q + lambda - rho v
        for (int j0 = g; j0 < dim; j0 += U * LPI) {
            double q[U], du[U], pr[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * LPI;
                q[u] = (j < dim) ? AT(QV, j % dnm) : 0.0;
                du[u] = (j < dim) ? AT(LAM, j) : 0.0;
                pr[u] = (j < dim) ? AT(V, j) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * LPI;
                if (j < dim) QH[j * T + cc] = q[u] + du[u] - (sr ? c.rho : cRho[j]) * pr[u];
            }
        }
        // rhs = (-Aeq Hhat^-1) q_hat - b
        spmv_stream<LPI>(lds, RH, nrow, recs + td.rhs.off, td.rhs.steps, g, cc);
        for (int i = g; i < n; i += LPI) RH[i * T + cc] -= AT(BH, i);
        // W mu = rhs through L D L'
        scatter_stream<LPI>(RH, recs + td.fwd.off, td.fwd.steps, g, cc);
        for (int i = g; i < nrow; i += LPI) RH[i * T + cc] *= Dinv[i];
        scatter_stream<LPI>(RH, recs + td.bwd.off, td.bwd.steps, g, cc);
        // z = (-Hhat^-1) q_hat + (-Hhat^-1 Aeq') mu
        spmv_stream<LPI>(lds, PL, dim, recs + td.prim.off, td.prim.steps, g, cc);
        bool res = false;
        for (int i0 = g; i0 < dim; i0 += U * LPI) {
            double lam[U], vo[U], lb[U], ub[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + u * LPI;
                lam[u] = (i < dim) ? AT(LAM, i) : 0.0;
                vo[u] = (i < dim) ? AT(V, i) : 0.0;
                lb[u] = (i < dim) ? cLB[i] : 0.0;
                ub[u] = (i < dim) ? cUB[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + u * LPI;
                if (i < dim) {
                    const double z = PL[i * T + cc];
                    const double v = fmin(fmax(z + (sr ? c.rho_i : cRhoi[i]) * lam[u], lb[u]), ub[u]);
                    if (active) {
                        AT(Z, i) = z;
                        AT(V, i) = v;
                        AT(LAM, i) = lam[u] + (sr ? c.rho : cRho[i]) * (z - v);
                    }
                    res |= (fabs(vo[u] - v) > c.tol) | (fabs(z - v) > c.tol);
                }
            }
        }
        const bool res_inst = or_over_group<LPI>(res, cc);
        const bool done_now = active && (!res_inst || k >= c.k_max);
        if (done_now) {
            if (g == 0) {
                k_out[t] = k;
                e_out[t] = res_inst ? -1 : 1;
            }
            active = false;
        }
        if (!__syncthreads_or(active ? 1 : 0)) break;
    }
#undef AT
}

}  // namespace tile
}  // namespace spcies
