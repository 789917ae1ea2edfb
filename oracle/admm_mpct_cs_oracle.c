/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, FP64, one instance per call, run-time dimensions) of the MPCT ADMM solver on the extended
 * state space ('cs' submethod):
 *
 *   formulations/+MPCT/code_MPCT_ADMM_cs_C.c:18-248   (scalar or vector rho)
 *
 * Operation order as the reference's loops; build with -ffp-contract=off.  Parity pin: the reference test's z_opt
 * (tests/test_MPCT_ADMM.m) and bit-identity with the compiled reference template - tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n, m, N, nrow, k_max, scalar_rho;
    double tol, rho, rho_i;
    const double *rho_v, *rho_i_v;       /* [N dnm] (vector rho) */
    const double *Tz, *Sz;               /* [n][n], [m][m] */
    const double *LB, *UB;               /* [N dnm] */
    const double *L_val; const int *L_col, *L_row;    /* CSC of L - I [nrow] */
    const double *Dinv;
    const double *AHi_val; const int *AHi_col, *AHi_row;   /* CSR [nrow x N dnm] */
    const double *HiA_val; const int *HiA_col, *HiA_row;   /* CSR [N dnm x nrow] */
    const double *Hi_val; const int *Hi_col, *Hi_row;      /* CSR [N dnm x N dnm] */
} mpct_cs_data;

int oracle_mpct_cs_solve(const mpct_cs_data *D, const double *x0, const double *xr, const double *ur, double *u_opt, int *k_out,
                         int *e_flag, double *z_out, double *v_out, double *lam_out) {
    const int n = D->n, m = D->m, dnm = 2 * (n + m), dim = D->N * dnm, nrow = D->nrow;
    double *z = (double *)calloc((size_t)dim, sizeof(double)), *v = (double *)calloc((size_t)dim, sizeof(double));
    double *lambda = (double *)calloc((size_t)dim, sizeof(double)), *v1 = (double *)calloc((size_t)dim, sizeof(double));
    double *q_hat = (double *)calloc((size_t)dim, sizeof(double)), *mu = (double *)calloc((size_t)nrow, sizeof(double));
    double *q = (double *)calloc((size_t)dnm, sizeof(double)), *b = (double *)calloc((size_t)n, sizeof(double));
    for (int i = 0; i < n; i++) b[i] = x0[i];
    /* q (:76-85) */
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) q[j + n] += D->Tz[(size_t)j * n + i] * xr[i];
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) q[j + 2 * n + m] += D->Sz[(size_t)j * m + i] * ur[i];
    int k = 0, done = 0, flag = -1;
    while (!done) {
        k += 1;
        memcpy(v1, v, sizeof(double) * (size_t)dim);
        /* q_hat = q + lambda - rho v (:103-109) */
        for (int j = 0; j < dim; j++)
            q_hat[j] = D->scalar_rho ? q[j % dnm] + lambda[j] - D->rho * v[j] : q[j % dnm] + lambda[j] - D->rho_v[j] * v[j];
        /* rhs = (-Aeq Hhat^-1) q_hat - b (:113-121) */
        for (int i = 0; i < nrow; i++) {
            double r = 0.0;
            for (int j = D->AHi_row[i]; j < D->AHi_row[i + 1]; j++) r += D->AHi_val[j] * q_hat[D->AHi_col[j]];
            mu[i] = r;
        }
        for (int j = 0; j < n; j++) mu[j] -= b[j];
        /* L D L' solve (:126-146) */
        for (int i = 0; i < nrow; i++)
            for (int j = D->L_col[i]; j < D->L_col[i + 1]; j++) mu[D->L_row[j]] -= D->L_val[j] * mu[i];
        for (int j = 0; j < nrow; j++) mu[j] *= D->Dinv[j];
        for (int i = nrow - 1; i >= 0; i--)
            for (int j = D->L_col[i]; j < D->L_col[i + 1]; j++) mu[i] -= D->L_val[j] * mu[D->L_row[j]];
        /* z = (-Hhat^-1) q_hat + (-Hhat^-1 Aeq') mu (:152-164) */
        for (int i = 0; i < dim; i++) {
            z[i] = 0.0;
            for (int j = D->Hi_row[i]; j < D->Hi_row[i + 1]; j++) z[i] += D->Hi_val[j] * q_hat[D->Hi_col[j]];
        }
        for (int i = 0; i < dim; i++)
            for (int j = D->HiA_row[i]; j < D->HiA_row[i + 1]; j++) z[i] += D->HiA_val[j] * mu[D->HiA_col[j]];
        /* v, lambda (:168-188) */
        for (int j = 0; j < dim; j++) {
            v[j] = D->scalar_rho ? z[j] + D->rho_i * lambda[j] : z[j] + D->rho_i_v[j] * lambda[j];
            v[j] = (v[j] > D->LB[j]) ? v[j] : D->LB[j];
            v[j] = (v[j] > D->UB[j]) ? D->UB[j] : v[j];
        }
        for (int j = 0; j < dim; j++)
            lambda[j] = D->scalar_rho ? lambda[j] + D->rho * (z[j] - v[j]) : lambda[j] + D->rho_v[j] * (z[j] - v[j]);
        /* residuals, exit (:192-217) */
        int rf = 0;
        for (int j = 0; j < dim; j++) {
            double r1 = v1[j] - v[j], r2 = z[j] - v[j];
            r1 = (r1 > 0.0) ? r1 : -r1;
            r2 = (r2 > 0.0) ? r2 : -r2;
            if (r1 > D->tol || r2 > D->tol) { rf = 1; break; }
        }
        if (!rf) { done = 1; flag = 1; }
        else if (k >= D->k_max) { done = 1; flag = -1; }
    }
    for (int j = 0; j < m; j++) u_opt[j] = v[2 * n + j];
    *k_out = k;
    *e_flag = flag;
    if (z_out) memcpy(z_out, z, sizeof(double) * (size_t)dim);
    if (v_out) memcpy(v_out, v, sizeof(double) * (size_t)dim);
    if (lam_out) memcpy(lam_out, lambda, sizeof(double) * (size_t)dim);
    free(z); free(v); free(lambda); free(v1); free(q_hat); free(mu); free(q); free(b);
    return 0;
}

int oracle_mpct_cs_batch(const mpct_cs_data *D, long B, const double *x0, const double *xr, const double *ur, int ref_stride,
                         double *u, int *k, int *e_flag, double *z, double *v, double *lam) {
    const size_t dim = (size_t)D->N * 2 * (size_t)(D->n + D->m);
    for (long i = 0; i < B; i++) {
        const double *xri = ref_stride ? xr + (size_t)i * D->n : xr, *uri = ref_stride ? ur + (size_t)i * D->m : ur;
        int rc = oracle_mpct_cs_solve(D, x0 + (size_t)i * D->n, xri, uri, u + (size_t)i * D->m, k + i, e_flag + i,
                                      z ? z + i * dim : NULL, v ? v + i * dim : NULL, lam ? lam + i * dim : NULL);
        if (rc) return rc;
    }
    return 0;
}
