/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, FP64, one instance per call, run-time dimensions) of the HMPC ADMM / SADMM solver
 * WITHOUT the splitting - the reference's default HMPC solver:
 *
 *   formulations/+HMPC/code_HMPC_ADMM_C.c:18-310   (box constraints; diamond or USE_SOC cones; IS_SYMMETRIC = SADMM)
 *
 * z = M2 b + M1 q_hat is the dense product of :145-157; C and C' arrive in CSR as the generator prints them
 * (cons_HMPC_ADMM_C.m:123-131).  Operation order as the reference's loops; build with -ffp-contract=off.
 * Parity pin: bit-identical to the compiled reference template (oracle/ref_template.py) and the reference test's
 * z_opt (tests/test_HMPC_ADMM.m:24) - see tests/test_oracle_golden.py.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n, m, N, dim, n_s, n_box, n_soc, k_max, use_soc, symmetric;
    double tol_p, tol_d, rho, rho_i, alpha;
    const double *A, *QQ, *Te, *Se;      /* [n][n] x3, [m][m] */
    const double *LB, *UB;               /* [n_box] */
    const double *LBy, *UBy;             /* [n+m] */
    const double *d;                     /* [n_s] (read with USE_SOC only) */
    const double *C_val; const int *C_col, *C_row;     /* CSR of C  [n_s x dim] */
    const double *Ct_val; const int *Ct_col, *Ct_row;  /* CSR of C' [dim x n_s] */
    const double *M1;                    /* [dim][dim] */
    const double *M2;                    /* [dim][n] */
} hmpc_dense_data;

static inline double absd(double x) { return (x > 0.0) ? x : -x; }

/* snippets/proj_SOC3.c:4-35 */
static void proj_SOC3(double *x, double alpha, double d) {
    double x_0 = x[0], x_norm = 0.0;
    for (int j = 1; j < 3; j++) x_norm += x[j] * x[j];
    x_norm = sqrt(x_norm);
    const double corrected = alpha * (x_0 - d);
    if (x_norm <= corrected) {
    } else if (x_norm <= -corrected) {
        x[0] = d; x[1] = 0.0; x[2] = 0.0;
    } else {
        const double step = (corrected + x_norm) / (2 * x_norm);
        x[0] = step * x_norm * alpha + d;
        for (int j = 1; j < 3; j++) x[j] = step * x[j];
    }
}

int oracle_hmpc_dense_solve(const hmpc_dense_data *D, const double *x0, const double *xr, const double *ur, double *u_opt,
                            int *k_out, int *e_flag, double *z_out, double *s_out, double *lam_out) {
    const int n = D->n, m = D->m, nm = n + m, N = D->N, dim = D->dim, n_s = D->n_s;
    double *q = (double *)calloc((size_t)dim, sizeof(double)), *z = (double *)calloc((size_t)dim, sizeof(double));
    double *q_hat = (double *)calloc((size_t)dim, sizeof(double)), *s = (double *)calloc((size_t)n_s, sizeof(double));
    double *Cz = (double *)calloc((size_t)n_s, sizeof(double)), *s_ant = (double *)calloc((size_t)n_s, sizeof(double));
    double *lambda = (double *)calloc((size_t)n_s, sizeof(double)), *b = (double *)calloc((size_t)n, sizeof(double));
    double *s_cone = s + D->n_box;
    /* setup (:82-107) */
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) b[j] -= D->A[(size_t)j * n + i] * x0[i];
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) q[(N - 1) * nm + m + j] -= D->Te[(size_t)j * n + i] * xr[i] + D->QQ[(size_t)j * n + i] * x0[i];
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) q[(N - 1) * nm + 2 * n + m + j] -= D->QQ[(size_t)j * n + i] * x0[i];
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) q[(N - 1) * nm + 3 * n + m + j] -= D->Se[(size_t)j * m + i] * ur[i];

    int k = 0, done = 0, flag = -1;
    while (!done) {
        k += 1;
        memcpy(s_ant, s, sizeof(double) * (size_t)n_s);
        /* q_hat = q + C'(rho (s - d) + lambda)  (:123-137) */
        for (int i = 0; i < n_s; i++) Cz[i] = D->use_soc ? D->rho * (s[i] - D->d[i]) + lambda[i] : D->rho * s[i] + lambda[i];
        for (int i = 0; i < dim; i++) {
            q_hat[i] = q[i];
            for (int j = D->Ct_row[i]; j < D->Ct_row[i + 1]; j++) q_hat[i] += D->Ct_val[j] * Cz[D->Ct_col[j]];
        }
        /* z = M2 b + M1 q_hat  (:145-157) */
        for (int i = 0; i < dim; i++) z[i] = 0.0;
        for (int i = 0; i < dim; i++)
            for (int j = 0; j < n; j++) z[i] += D->M2[(size_t)i * n + j] * b[j];
        for (int i = 0; i < dim; i++)
            for (int j = 0; j < dim; j++) z[i] += D->M1[(size_t)i * dim + j] * q_hat[j];
        /* C z (- d)  (:161-170) */
        for (int i = 0; i < n_s; i++) {
            Cz[i] = D->use_soc ? -D->d[i] : 0.0;
            for (int j = D->C_row[i]; j < D->C_row[i + 1]; j++) Cz[i] += D->C_val[j] * z[D->C_col[j]];
        }
        if (D->symmetric)
            for (int j = 0; j < n_s; j++) lambda[j] += D->alpha * D->rho * (Cz[j] + s[j]);
        for (int j = 0; j < n_s; j++) s[j] = -Cz[j] - D->rho_i * lambda[j];
        for (int j = 0; j < D->n_box; j++) {
            s[j] = (s[j] > D->LB[j]) ? s[j] : D->LB[j];
            s[j] = (s[j] > D->UB[j]) ? D->UB[j] : s[j];
        }
        if (D->use_soc) {
            for (int j = 0; j < D->n_soc; j++) proj_SOC3(&s_cone[3 * j], 1.0, 0.0);
        } else {
            for (int j = 0; j < D->n_soc; j++) { /* n_y outputs (:200): n + m with box constraints, the rows of E / F when coupled */
                proj_SOC3(&s_cone[3 * j], 1.0, D->LBy[j]);
                proj_SOC3(&s_cone[3 * j], -1.0, D->UBy[j]);
            }
        }
        for (int j = 0; j < n_s; j++) Cz[j] += s[j];
        if (D->symmetric)
            for (int j = 0; j < n_s; j++) lambda[j] += D->alpha * D->rho * Cz[j];
        else
            for (int j = 0; j < n_s; j++) lambda[j] += D->rho * Cz[j];
        int rf = 0;
        for (int j = 0; j < n_s; j++)
            if (absd(Cz[j]) > D->tol_p || absd(s[j] - s_ant[j]) > D->tol_d) { rf = 1; break; }
        if (!rf) { done = 1; flag = 1; }
        else if (k >= D->k_max) { done = 1; flag = -1; }
    }
    for (int j = 0; j < m; j++) u_opt[j] = z[j];
    *k_out = k;
    *e_flag = flag;
    if (z_out) memcpy(z_out, z, sizeof(double) * (size_t)dim);
    if (s_out) memcpy(s_out, s, sizeof(double) * (size_t)n_s);
    if (lam_out) memcpy(lam_out, lambda, sizeof(double) * (size_t)n_s);
    free(q); free(z); free(q_hat); free(s); free(Cz); free(s_ant); free(lambda); free(b);
    return 0;
}

int oracle_hmpc_dense_batch(const hmpc_dense_data *D, long B, const double *x0, const double *xr, const double *ur, int ref_stride,
                            double *u, int *k, int *e_flag, double *z, double *s, double *lam) {
    for (long i = 0; i < B; i++) {
        const double *xri = ref_stride ? xr + (size_t)i * D->n : xr, *uri = ref_stride ? ur + (size_t)i * D->m : ur;
        int rc = oracle_hmpc_dense_solve(D, x0 + (size_t)i * D->n, xri, uri, u + (size_t)i * D->m, k + i, e_flag + i,
                                         z ? z + (size_t)i * D->dim : NULL, s ? s + (size_t)i * D->n_s : NULL,
                                         lam ? lam + (size_t)i * D->n_s : NULL);
        if (rc) return rc;
    }
    return 0;
}
