/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, FP64, one instance per call, runtime dimensions) of the ADMM solver the
 * reference generates for ellipMPC with the terminal ellipsoid as a second-order cone:
 *
 *   formulations/+ellipMPC/code_ellipMPC_ADMM_soc_C.c:20-329
 *
 * CSR mat-vecs (:157-165, :193-205), sparse L D L' solve in the QDLDL style (:172-188), box on
 * z[0 : dim-n-1] (:214-217), SOC projection of s (:223-242), dual update and exit test (:246-272).
 * Accumulation order follows the reference; built with -ffp-contract=off.
 *
 * Parity pin: tests/test_oracle_golden.py (z[0:dim-1] vs z_opt of tests/test_ellipMPC_ADMM_soc.m:38,
 * tolerance 1e-4) and tests/golden/template_*soc*.npz (bit-exact).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n, m, N, dim, n_s, n_eq, k_max;
    double tol_p, tol_d, rho, rho_i, sigma, sigma_i;
    const double *A;      /* [n][n] */
    const double *Q;      /* [n][n] negated */
    const double *R;      /* [m][m] negated */
    const double *T;      /* [n][n] negated */
    const double *LB, *UB;/* [dim-n-1] */
    const double *PhiP;   /* [n][n] */
    const double *L_val; const int *L_col, *L_row; const double *Dinv;   /* CSC of L - I (n_eq+n_s cols), 0-based */
    const double *GhHhi_val; const int *GhHhi_col, *GhHhi_row;           /* CSR, n_eq+n_s rows */
    const double *HhiGh_val; const int *HhiGh_col, *HhiGh_row;           /* CSR, dim+n_s rows  */
    const double *Hhi_val; const int *Hhi_col, *Hhi_row;                 /* CSR, dim+n_s rows  */
} admm_soc_data;

static inline double absd(double x) { return (x > 0.0) ? x : -x; }

/* outputs (any may be NULL): z, z_hat, lambda [dim]; s, s_hat, mu [n_s] */
int oracle_admm_soc_solve(const admm_soc_data *d, const double *x0, const double *xr, const double *ur, double r_ellip,
                          double *u_opt, int *k_out, int *e_flag, double *z_out, double *s_out, double *zh_out,
                          double *sh_out, double *lam_out, double *mu_out) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N, dim = d->dim, n_s = d->n_s, n_eq = d->n_eq;
    const int np_ = dim + n_s, nr = n_eq + n_s;
    double *primal = (double *)calloc((size_t)np_, sizeof(double)), *primal_ant = (double *)calloc((size_t)np_, sizeof(double));
    double *primal_hat = (double *)calloc((size_t)np_, sizeof(double)), *dual = (double *)calloc((size_t)np_, sizeof(double));
    double *bh = (double *)calloc((size_t)nr, sizeof(double)), *q_hat = (double *)calloc((size_t)np_, sizeof(double));
    double *q = (double *)calloc((size_t)dim, sizeof(double)), *rhs = (double *)calloc((size_t)nr, sizeof(double));
    double *z = primal, *s = primal + dim, *z_hat = primal_hat, *s_hat = primal_hat + dim, *lambda = dual, *mu = dual + dim;

    /* setup (:84-131) */
    for (int j = 0; j < n; j++) {
        bh[j] = 0.0;
        for (int i = 0; i < n; i++) bh[j] -= d->A[(size_t)j * n + i] * x0[i];
    }
    bh[n_eq - 1] = r_ellip;
    for (int j = 0; j < n; j++) {
        bh[n_eq + 1 + j] = 0.0;
        for (int i = 0; i < n; i++) bh[n_eq + 1 + j] -= d->PhiP[(size_t)j * n + i] * xr[i];
    }
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) q[j] += d->R[(size_t)j * m + i] * ur[i];
    for (int k = 0; k < N - 1; k++) {
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) q[m + k * nm + j] += d->Q[(size_t)j * n + i] * xr[i];
        for (int j = 0; j < m; j++)
            for (int i = 0; i < m; i++) q[nm + k * nm + j] += d->R[(size_t)j * m + i] * ur[i];
    }
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) q[m + (N - 1) * nm + j] += d->T[(size_t)j * n + i] * xr[i];

    int k = 0, done = 0, flag = -1;
    while (!done) {
        k += 1;
        memcpy(primal_ant, primal, sizeof(double) * (size_t)np_);
        for (int j = 0; j < dim; j++) q_hat[j] = q[j] + lambda[j] - d->sigma * z[j];
        for (int j = 0; j < n_s; j++) q_hat[j + dim] = mu[j] - d->rho * s[j];
        for (int i = 0; i < nr; i++) {
            rhs[i] = 0.0;
            for (int j = d->GhHhi_row[i]; j < d->GhHhi_row[i + 1]; j++) rhs[i] += d->GhHhi_val[j] * q_hat[d->GhHhi_col[j]];
        }
        for (int j = 0; j < nr; j++) rhs[j] -= bh[j];
        for (int i = 0; i < nr; i++)
            for (int j = d->L_col[i]; j < d->L_col[i + 1]; j++) rhs[d->L_row[j]] -= d->L_val[j] * rhs[i];
        for (int j = 0; j < nr; j++) rhs[j] *= d->Dinv[j];
        for (int i = nr - 1; i >= 0; i--)
            for (int j = d->L_col[i]; j < d->L_col[i + 1]; j++) rhs[i] -= d->L_val[j] * rhs[d->L_row[j]];
        for (int i = 0; i < np_; i++) {
            primal_hat[i] = 0.0;
            for (int j = d->Hhi_row[i]; j < d->Hhi_row[i + 1]; j++) primal_hat[i] += d->Hhi_val[j] * q_hat[d->Hhi_col[j]];
        }
        for (int i = 0; i < np_; i++)
            for (int j = d->HhiGh_row[i]; j < d->HhiGh_row[i + 1]; j++) primal_hat[i] += d->HhiGh_val[j] * rhs[d->HhiGh_col[j]];
        for (int j = 0; j < dim; j++) z[j] = z_hat[j] + d->sigma_i * lambda[j];
        for (int j = 0; j < dim - n - 1; j++) {
            z[j] = (z[j] > d->LB[j]) ? z[j] : d->LB[j];
            z[j] = (z[j] > d->UB[j]) ? d->UB[j] : z[j];
        }
        for (int j = 0; j < n_s; j++) s[j] = s_hat[j] + d->rho_i * mu[j];
        double s_norm = 0.0;
        for (int j = 1; j < n_s; j++) s_norm += s[j] * s[j];
        s_norm = sqrt(s_norm);
        if (s_norm <= s[0]) {
        } else if (s_norm <= -s[0]) {
            for (int j = 0; j < n_s; j++) s[j] = 0.0;
        } else {
            double step = (s[0] + s_norm) / (2 * s_norm);
            s[0] = step * s_norm;
            for (int j = 1; j < n_s; j++) s[j] = step * s[j];
        }
        for (int j = 0; j < dim; j++) lambda[j] += d->sigma * (z_hat[j] - z[j]);
        for (int j = 0; j < n_s; j++) mu[j] += d->rho * (s_hat[j] - s[j]);
        int rf = 0;
        for (int j = 0; j < np_; j++)
            if (absd(primal_ant[j] - primal[j]) > d->tol_d || absd(primal[j] - primal_hat[j]) > d->tol_p) { rf = 1; break; }
        if (!rf) { done = 1; flag = 1; }
        else if (k >= d->k_max) { done = 1; flag = -1; }
    }
    for (int j = 0; j < m; j++) u_opt[j] = z[j];
    *k_out = k;
    *e_flag = flag;
    if (z_out) memcpy(z_out, z, sizeof(double) * (size_t)dim);
    if (zh_out) memcpy(zh_out, z_hat, sizeof(double) * (size_t)dim);
    if (lam_out) memcpy(lam_out, lambda, sizeof(double) * (size_t)dim);
    if (s_out) memcpy(s_out, s, sizeof(double) * (size_t)n_s);
    if (sh_out) memcpy(sh_out, s_hat, sizeof(double) * (size_t)n_s);
    if (mu_out) memcpy(mu_out, mu, sizeof(double) * (size_t)n_s);
    free(primal); free(primal_ant); free(primal_hat); free(dual); free(bh); free(q_hat); free(q); free(rhs);
    return 0;
}

int oracle_admm_soc_batch(const admm_soc_data *d, long B, const double *x0, const double *xr, const double *ur,
                          int ref_stride, const double *r, int r_stride, double *u, int *k, int *e_flag, double *z,
                          double *s, double *zh, double *sh, double *lam, double *mu) {
    for (long i = 0; i < B; i++) {
        const double *xri = ref_stride ? xr + (size_t)i * d->n : xr;
        const double *uri = ref_stride ? ur + (size_t)i * d->m : ur;
        int rc = oracle_admm_soc_solve(d, x0 + (size_t)i * d->n, xri, uri, r[r_stride ? i : 0], u + (size_t)i * d->m, k + i,
                                       e_flag + i, z ? z + (size_t)i * d->dim : NULL, s ? s + (size_t)i * d->n_s : NULL,
                                       zh ? zh + (size_t)i * d->dim : NULL, sh ? sh + (size_t)i * d->n_s : NULL,
                                       lam ? lam + (size_t)i * d->dim : NULL, mu ? mu + (size_t)i * d->n_s : NULL);
        if (rc) return rc;
    }
    return 0;
}
