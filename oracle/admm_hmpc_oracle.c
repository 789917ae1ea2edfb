/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, FP64, one instance per call, runtime dimensions) of the ADMM / SADMM solver
 * the reference generates for HMPC with the (z_hat, s_hat) = (z, s) splitting, sparse KKT path, box
 * constraints:
 *
 *   formulations/+HMPC/code_HMPC_ADMM_split_C.c:19-393   and   snippets/proj_SOC3.c:4-35
 *
 * rhs = [sigma z - q - lambda; rho s - mu; bh] (:156-165), KKT solve through L D L' (:193-209), optional
 * half dual step of the symmetric variant (:215-225), box on z[0 : dim-3(n+m)] (:235-238), diamond or SOC
 * projection of s (:248-259), dual step (:288-312), exit test (:318-333).  Built with -ffp-contract=off.
 *
 * Parity pin: the reference tests' HMPC golden is stale, so this oracle is pinned by
 * tests/golden/template_*HMPC*.npz (bit-exact against the reference's compiled template) and by the
 * consistency / optimality checks of tests/test_oracle_golden.py.  "parity pinned by template + KKT".
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n, m, N, dim, n_s, n_eq, n_soc, nrow_M, k_max, use_soc, symmetric;
    double tol_p, tol_d, rho, rho_i, sigma, sigma_i, alpha;
    const double *A;      /* [n][n] */
    const double *QQ;     /* [n][n] */
    const double *Te;     /* [n][n] */
    const double *Se;     /* [m][m] */
    const double *LB, *UB;/* [dim - 3(n+m)] */
    const double *LBy, *UBy; /* [n+m] */
    const double *L_val; const int *L_col, *L_row; const double *Dinv;  /* CSC of L - I, nrow_M columns */
    const int *idx_x0;    /* [n] position of the x0 rows inside bh */
    const double *bh;     /* [n_eq + n_s] (copied: its x0 rows are rewritten per call) */
    /* NON_SPARSE (option sparse = false, the reference's default: def_options_HMPC_ADMM.m:31):
     * primal_hat = M2 bh - M1 q_hat (code_HMPC_ADMM_split_C.c:174-190) instead of the L D L' sweeps       */
    int non_sparse, dim_M2;   /* dim_M2 = n (diamond: only the x0 rows of bh are non-zero) or n_eq + n_s (use_soc)     */
    const double *M1;         /* [dim + n_s][dim + n_s]                                               */
    const double *M2;         /* [dim + n_s][dim_M2]                                                  */
    const double *bh_nat;     /* [n_eq + n_s] bh in its natural order (x0 rows first)                 */
    /* COUPLED_CONSTRAINTS (:65, 233-283): LBy <= E x + F u <= UBy.  s = [N n_y box slacks of the outputs; cone rows], z is free */
    int coupled, n_y;
} admm_hmpc_data;

static inline double absd(double x) { return (x > 0.0) ? x : -x; }

/* snippets/proj_SOC3.c:4-35 */
static void proj_SOC3(double *x, double alpha, double d) {
    double x_0 = x[0], x_norm = 0.0;
    for (int j = 1; j < 3; j++) x_norm += x[j] * x[j];
    x_norm = sqrt(x_norm);
    double corrected = alpha * (x_0 - d);
    if (x_norm <= corrected) {
    } else if (x_norm <= -corrected) {
        x[0] = d;
        for (int j = 1; j < 3; j++) x[j] = 0.0;
    } else {
        double step = (corrected + x_norm) / (2 * x_norm);
        x[0] = step * x_norm * alpha + d;
        for (int j = 1; j < 3; j++) x[j] = step * x[j];
    }
}

int oracle_admm_hmpc_solve(const admm_hmpc_data *d, const double *x0, const double *xr, const double *ur, double *u_opt,
                           int *k_out, int *e_flag, double *z_out, double *s_out, double *zh_out, double *sh_out,
                           double *lam_out, double *mu_out) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N, dim = d->dim, n_s = d->n_s, n_eq = d->n_eq, nrow = d->nrow_M;
    const int np_ = dim + n_s;
    double *q = (double *)calloc((size_t)dim, sizeof(double)), *rhs = (double *)calloc((size_t)nrow, sizeof(double));
    double *primal = (double *)calloc((size_t)np_, sizeof(double)), *primal_ant = (double *)calloc((size_t)np_, sizeof(double));
    double *dual = (double *)calloc((size_t)np_, sizeof(double)), *bh = (double *)malloc(sizeof(double) * (size_t)(n_eq + n_s));
    double *z = primal, *s = primal + dim, *lambda = dual, *mu = dual + dim, *z_hat = rhs, *s_hat = rhs + dim;
    memcpy(bh, d->non_sparse ? d->bh_nat : d->bh, sizeof(double) * (size_t)(n_eq + n_s));
    /* setup (:97-129) */
    for (int j = 0; j < n; j++) {
        const int at = d->non_sparse ? j : d->idx_x0[j];
        bh[at] = 0.0;
        for (int i = 0; i < n; i++) bh[at] -= d->A[(size_t)j * n + i] * x0[i];
    }
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) q[(N - 1) * nm + m + j] -= d->Te[(size_t)j * n + i] * xr[i] + d->QQ[(size_t)j * n + i] * x0[i];
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) q[(N - 1) * nm + 2 * n + m + j] -= d->QQ[(size_t)j * n + i] * x0[i];
    for (int j = 0; j < m; j++)
        for (int i = 0; i < m; i++) q[(N - 1) * nm + m + 3 * n + j] -= d->Se[(size_t)j * m + i] * ur[i];

    const double as = d->alpha * d->sigma, ar = d->alpha * d->rho;
    int k = 0, done = 0, flag = -1;
    while (!done) {
        k += 1;
        memcpy(primal_ant, primal, sizeof(double) * (size_t)np_);
        for (int j = 0; j < dim; j++) rhs[j] = d->sigma * z[j] - q[j] - lambda[j];
        for (int j = 0; j < n_s; j++) rhs[j + dim] = d->rho * s[j] - mu[j];
        if (d->non_sparse) { /* (:174-190): q_hat sits in rhs[0..np), the result replaces it */
            double *qh = (double *)malloc(sizeof(double) * (size_t)np_);
            memcpy(qh, rhs, sizeof(double) * (size_t)np_);
            for (int i = 0; i < np_; i++) rhs[i] = 0.0;
            for (int i = 0; i < np_; i++)
                for (int j = 0; j < d->dim_M2; j++) rhs[i] += d->M2[(size_t)i * d->dim_M2 + j] * bh[j];
            for (int i = 0; i < np_; i++)
                for (int j = 0; j < np_; j++) rhs[i] -= d->M1[(size_t)i * np_ + j] * qh[j];
            free(qh);
        } else {
        for (int j = 0; j < n_s + n_eq; j++) rhs[dim + n_s + j] = bh[j];
        for (int i = 0; i < nrow; i++)
            for (int j = d->L_col[i]; j < d->L_col[i + 1]; j++) rhs[d->L_row[j]] -= d->L_val[j] * rhs[i];
        for (int j = 0; j < nrow; j++) rhs[j] *= d->Dinv[j];
        for (int i = nrow - 1; i >= 0; i--)
            for (int j = d->L_col[i]; j < d->L_col[i + 1]; j++) rhs[i] -= d->L_val[j] * rhs[d->L_row[j]];
        }
        if (d->symmetric) { /* (:215-225) alpha_SADMM*sigma*(...) evaluates left to right */
            for (int j = 0; j < dim; j++) lambda[j] += as * (z_hat[j] - z[j]);
            for (int j = 0; j < n_s; j++) mu[j] += ar * (s_hat[j] - s[j]);
        }
        for (int j = 0; j < dim; j++) z[j] = z_hat[j] + d->sigma_i * lambda[j];
        if (!d->coupled)
            for (int j = 0; j < dim - 3 * n - 3 * m; j++) {
                z[j] = (z[j] > d->LB[j]) ? z[j] : d->LB[j];
                z[j] = (z[j] > d->UB[j]) ? d->UB[j] : z[j];
            }
        for (int j = 0; j < n_s; j++) s[j] = s_hat[j] + d->rho_i * mu[j];
        double *s_cone = s;
        if (d->coupled) { /* (:262-283) box on the output slacks, stage by stage; the cones sit behind them */
            for (int j = 0; j < N; j++)
                for (int i = 0; i < d->n_y; i++) {
                    double *e = &s[j * d->n_y + i];
                    *e = (*e > d->LBy[i]) ? *e : d->LBy[i];
                    *e = (*e > d->UBy[i]) ? d->UBy[i] : *e;
                }
            s_cone = s + N * d->n_y;
        }
        if (d->use_soc) {
            for (int j = 0; j < d->n_soc; j++) proj_SOC3(&s_cone[3 * j], 1.0, 0.0);
        } else {
            for (int j = 0; j < (d->coupled ? d->n_y : nm); j++) {
                proj_SOC3(&s_cone[3 * j], 1.0, d->LBy[j]);
                proj_SOC3(&s_cone[3 * j], -1.0, d->UBy[j]);
            }
        }
        if (d->symmetric) {
            for (int j = 0; j < dim; j++) lambda[j] += as * (z_hat[j] - z[j]);
            for (int j = 0; j < n_s; j++) mu[j] += ar * (s_hat[j] - s[j]);
        } else {
            for (int j = 0; j < dim; j++) lambda[j] += d->sigma * (z_hat[j] - z[j]);
            for (int j = 0; j < n_s; j++) mu[j] += d->rho * (s_hat[j] - s[j]);
        }
        int rf = 0;
        for (int j = 0; j < np_; j++)
            if (absd(primal_ant[j] - primal[j]) > d->tol_d || absd(primal[j] - rhs[j]) > d->tol_p) { rf = 1; break; }
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
    free(q); free(rhs); free(primal); free(primal_ant); free(dual); free(bh);
    return 0;
}

int oracle_admm_hmpc_batch(const admm_hmpc_data *d, long B, const double *x0, const double *xr, const double *ur,
                           int ref_stride, double *u, int *k, int *e_flag, double *z, double *s, double *zh, double *sh,
                           double *lam, double *mu) {
    for (long i = 0; i < B; i++) {
        const double *xri = ref_stride ? xr + (size_t)i * d->n : xr;
        const double *uri = ref_stride ? ur + (size_t)i * d->m : ur;
        int rc = oracle_admm_hmpc_solve(d, x0 + (size_t)i * d->n, xri, uri, u + (size_t)i * d->m, k + i, e_flag + i,
                                        z ? z + (size_t)i * d->dim : NULL, s ? s + (size_t)i * d->n_s : NULL,
                                        zh ? zh + (size_t)i * d->dim : NULL, sh ? sh + (size_t)i * d->n_s : NULL,
                                        lam ? lam + (size_t)i * d->dim : NULL, mu ? mu + (size_t)i * d->n_s : NULL);
        if (rc) return rc;
    }
    return 0;
}
