/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, FP64, one instance per call, runtime dimensions) of the EADMM solver the
 * reference generates for the MPCT formulation, both the diagonal-Q/R path (IS_DIAG == 1) and the general one
 * (IS_DIAG == 0, :184-217 and :321-366: dense inverses Q_bi / Q_mi / R_bi / R_mi of the blocks of H3, AB_bi / AB_mi):
 *
 *   formulations/+MPCT/code_MPCT_EADMM_C.c:18-525
 *
 * Three-block extended ADMM: P1 (z1, clamp, :97-117), P2 (z2 = W2 q2, :123-149), P3 (z3 through the
 * banded Cholesky solve, :157-320), residual / dual update (:371-402), three-part exit test (:408-457).
 * Accumulation order follows the reference loop nests; built with -ffp-contract=off.
 *
 * Parity pin: tests/test_oracle_golden.py (z1 vs z_opt of tests/test_MPCT_EADMM.m:32, tolerance 1e-4)
 * and tests/golden/template_*MPCT*.npz (bit-exact).
 */
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n, m, N, k_max;
    double tol;
    const double *rho;    /* [N+1][n+m] */
    const double *rho_0;  /* [n+m] (last m entries zero) */
    const double *rho_s;  /* [n+m] */
    const double *LB, *UB, *LB_0, *UB_0, *LB_s, *UB_s; /* [n+m] each */
    const double *AB;     /* [n][n+m] */
    const double *T;      /* [n][n]  negated */
    const double *S;      /* [m][m]  negated */
    const double *Alpha;  /* [N-1][n][n] */
    const double *Beta;   /* [N][n][n] */
    const double *H1i;    /* [N+1][n+m] */
    const double *W2;     /* [n+m][n+m] */
    const double *H3i;    /* [N+1][n+m]  (IS_DIAG == 1) */
    int is_diag;
    const double *Q_bi, *Q_mi; /* [n][n]    (IS_DIAG == 0; cons_MPCT_EADMM_C.m:102-107) */
    const double *R_bi, *R_mi; /* [m][m]   */
    const double *AB_bi, *AB_mi; /* [n][n+m] */
} eadmm_mpct_data;

#define M2(a, l, j) ((a)[(size_t)(l) * nm + (j)])
#define MU(l, j) (mu[(size_t)(l) * n + (j)])
#define ABij(i, j) (d->AB[(size_t)(i) * nm + (j)])
#define ALPHA(l, i, j) (d->Alpha[((size_t)(l) * n + (i)) * n + (j)])
#define BETA(l, i, j) (d->Beta[((size_t)(l) * n + (i)) * n + (j)])

static inline double clampd(double x, double lo, double hi) {
    x = (x > lo) ? x : lo;
    x = (x > hi) ? hi : x;
    return x;
}
static inline double absd(double x) { return (x > 0.0) ? x : -x; }

/* Outputs: u [m]; z1, z3 [(N+1)(n+m)]; z2 [n+m]; lambda [(N+3)(n+m)] packed as the DEBUG copy-out
 * of the reference does (:495-513: only the first n entries of each row are copied, contiguously;
 * the remainder of the buffer stays zero).  Any of z1/z2/z3/lam may be NULL. */
int oracle_eadmm_mpct_solve(const eadmm_mpct_data *d, const double *x0_in, const double *xr, const double *ur,
                            double *u_opt, int *k_out, int *e_flag, double *z1_out, double *z2_out, double *z3_out,
                            double *lam_out) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    if (n <= 0 || m <= 0 || N < 2) return -1;
    double *x0 = (double *)calloc((size_t)nm, sizeof(double));
    double *z1 = (double *)calloc((size_t)(N + 1) * nm, sizeof(double));
    double *z3 = (double *)calloc((size_t)(N + 1) * nm, sizeof(double));
    double *z3p = (double *)calloc((size_t)(N + 1) * nm, sizeof(double));
    double *z2 = (double *)calloc((size_t)nm, sizeof(double)), *z2p = (double *)calloc((size_t)nm, sizeof(double));
    double *q2 = (double *)calloc((size_t)nm, sizeof(double));
    double *lam = (double *)calloc((size_t)(N + 3) * nm, sizeof(double));
    double *res = (double *)calloc((size_t)(N + 3) * nm, sizeof(double));
    double *mu = (double *)calloc((size_t)N * n, sizeof(double));
    for (int i = 0; i < n; i++) x0[i] = x0_in[i];

    int k = 0, done = 0, flag = -1;
    while (!done) {
        k += 1;
        memcpy(z2p, z2, sizeof(double) * (size_t)nm);
        memcpy(z3p, z3, sizeof(double) * (size_t)(N + 1) * nm);
        /* ---- P1 (:97-117) */
        for (int j = 0; j < nm; j++) {
            double v = (M2(d->rho, 0, j) * (M2(z3, 0, j) + z2[j]) + d->rho_0[j] * x0[j] + M2(lam, 1, j) - M2(lam, 0, j)) *
                       M2(d->H1i, 0, j);
            M2(z1, 0, j) = clampd(v, d->LB_0[j], d->UB_0[j]);
        }
        for (int l = 1; l < N; l++)
            for (int j = 0; j < nm; j++) {
                double v = (M2(d->rho, l, j) * (M2(z3, l, j) + z2[j]) + M2(lam, l + 1, j)) * M2(d->H1i, l, j);
                M2(z1, l, j) = clampd(v, d->LB[j], d->UB[j]);
            }
        for (int j = 0; j < nm; j++) {
            double v = (M2(d->rho, N, j) * M2(z3, N, j) + (M2(d->rho, N, j) + d->rho_s[j]) * z2[j] + M2(lam, N + 1, j) +
                        M2(lam, N + 2, j)) * M2(d->H1i, N, j);
            M2(z1, N, j) = clampd(v, d->LB_s[j], d->UB_s[j]);
        }
        /* ---- P2 (:123-149) */
        for (int j = 0; j < nm; j++)
            q2[j] = M2(d->rho, N, j) * M2(z3, N, j) - (M2(d->rho, N, j) + d->rho_s[j]) * M2(z1, N, j) + M2(lam, N + 1, j) +
                    M2(lam, N + 2, j);
        for (int j = 0; j < n; j++)
            for (int i = 0; i < n; i++) q2[j] = q2[j] + d->T[(size_t)j * n + i] * xr[i];
        for (int j = 0; j < m; j++)
            for (int i = 0; i < m; i++) q2[j + n] = q2[j + n] + d->S[(size_t)j * m + i] * ur[i];
        for (int l = 0; l < N; l++)
            for (int j = 0; j < nm; j++) q2[j] = q2[j] + M2(d->rho, l, j) * (M2(z3, l, j) - M2(z1, l, j)) + M2(lam, l + 1, j);
        for (int j = 0; j < nm; j++) {
            double acc = 0;
            for (int i = 0; i < nm; i++) acc = acc + d->W2[(size_t)j * nm + i] * q2[i];
            z2[j] = acc;
        }
        /* ---- P3: q3 stored in z3 (:157-172) */
        for (int l = 0; l <= N; l++)
            for (int j = 0; j < nm; j++) M2(z3, l, j) = M2(d->rho, l, j) * (z2[j] - M2(z1, l, j)) + M2(lam, l + 1, j);
        /* rhs (:176-183; general Q, R :184-217) */
        if (d->is_diag) {
            for (int l = 0; l < N; l++)
                for (int j = 0; j < n; j++) {
                    double acc = M2(d->H3i, l + 1, j) * M2(z3, l + 1, j);
                    for (int i = 0; i < nm; i++) acc = acc - ABij(j, i) * M2(d->H3i, l, i) * M2(z3, l, i);
                    MU(l, j) = acc;
                }
        } else {
            for (int l = 0; l < N - 1; l++)
                for (int j = 0; j < n; j++) {
                    MU(l, j) = 0.0;
                    for (int i = 0; i < n; i++) MU(l, j) += d->Q_bi[(size_t)j * n + i] * M2(z3, l + 1, i);
                }
            for (int j = 0; j < n; j++) {
                MU(N - 1, j) = 0.0;
                for (int i = 0; i < n; i++) MU(N - 1, j) += d->Q_mi[(size_t)j * n + i] * M2(z3, N, i);
            }
            for (int j = 0; j < n; j++)
                for (int i = 0; i < nm; i++) MU(0, j) -= d->AB_mi[(size_t)j * nm + i] * M2(z3, 0, i);
            for (int l = 1; l < N; l++)
                for (int j = 0; j < n; j++)
                    for (int i = 0; i < nm; i++) MU(l, j) -= d->AB_bi[(size_t)j * nm + i] * M2(z3, l, i);
        }
        /* banded solve (:221-285) */
        for (int l = 0; l < N; l++)
            for (int j = 0; j < n; j++) {
                double acc = MU(l, j);
                if (l > 0)
                    for (int i = 0; i < n; i++) acc = acc - ALPHA(l - 1, i, j) * MU(l - 1, i);
                for (int i = 0; i < j; i++) acc = acc - BETA(l, i, j) * MU(l, i);
                MU(l, j) = BETA(l, j, j) * acc;
            }
        for (int l = N - 1; l >= 0; l--)
            for (int j = n - 1; j >= 0; j--) {
                double acc = MU(l, j);
                if (l < N - 1)
                    for (int i = n - 1; i >= 0; i--) acc = acc - ALPHA(l, j, i) * MU(l + 1, i);
                for (int i = n - 1; i > j; i--) acc = acc - BETA(l, j, i) * MU(l, i);
                MU(l, j) = BETA(l, j, j) * acc;
            }
        /* z3 (:289-320) */
        for (int j = 0; j < nm; j++)
            for (int i = 0; i < n; i++) M2(z3, 0, j) = M2(z3, 0, j) + ABij(i, j) * MU(0, i);
        for (int l = 1; l < N; l++) {
            for (int j = 0; j < n; j++) M2(z3, l, j) = M2(z3, l, j) - MU(l - 1, j);
            for (int j = 0; j < nm; j++)
                for (int i = 0; i < n; i++) M2(z3, l, j) = M2(z3, l, j) + ABij(i, j) * MU(l, i);
        }
        for (int j = 0; j < n; j++) M2(z3, N, j) = M2(z3, N, j) - MU(N - 1, j);
        if (d->is_diag) {
            for (int l = 0; l <= N; l++)
                for (int j = 0; j < nm; j++) M2(z3, l, j) = -M2(d->H3i, l, j) * M2(z3, l, j);
        } else { /* (:321-366): z3 = -H3^-1 z3 block by block; z3p has been compared against already?  no: keep it, use a scratch copy */
            double *aux = (double *)malloc(sizeof(double) * (size_t)(N + 1) * nm);
            memcpy(aux, z3, sizeof(double) * (size_t)(N + 1) * nm);
            memset(z3, 0, sizeof(double) * (size_t)(N + 1) * nm);
            for (int l = 0; l <= N; l++) {
                const double *Qi = (l == 0 || l == N) ? d->Q_mi : d->Q_bi;
                const double *Ri = (l == N) ? d->R_mi : d->R_bi;
                for (int j = 0; j < n; j++)
                    for (int i = 0; i < n; i++) M2(z3, l, j) -= Qi[(size_t)j * n + i] * aux[(size_t)l * nm + i];
                for (int j = 0; j < m; j++)
                    for (int i = 0; i < m; i++) M2(z3, l, j + n) -= Ri[(size_t)j * m + i] * aux[(size_t)l * nm + i + n];
            }
            free(aux);
        }
        /* ---- residual and lambda (:371-402) */
        for (int j = 0; j < n; j++) M2(res, 0, j) = M2(z1, 0, j) - x0[j];
        for (int l = 0; l <= N; l++)
            for (int j = 0; j < nm; j++) M2(res, l + 1, j) = z2[j] + M2(z3, l, j) - M2(z1, l, j);
        for (int j = 0; j < nm; j++) M2(res, N + 2, j) = z2[j] - M2(z1, N, j);
        for (int j = 0; j < n; j++) M2(lam, 0, j) = M2(lam, 0, j) + d->rho_0[j] * M2(res, 0, j);
        for (int l = 1; l < N + 2; l++)
            for (int j = 0; j < nm; j++) M2(lam, l, j) = M2(lam, l, j) + M2(d->rho, l - 1, j) * M2(res, l, j);
        for (int j = 0; j < nm; j++) M2(lam, N + 2, j) = M2(lam, N + 2, j) + d->rho_s[j] * M2(res, N + 2, j);
        /* ---- exit (:408-457) */
        int rf = 0;
        for (int j = 0; j < nm && !rf; j++)
            if (absd(z2p[j] - z2[j]) > d->tol) rf = 1;
        for (size_t i = 0; i < (size_t)(N + 3) * nm && !rf; i++)
            if (absd(res[i]) > d->tol) rf = 1;
        for (size_t i = 0; i < (size_t)(N + 1) * nm && !rf; i++)
            if (absd(z3p[i] - z3[i]) > d->tol) rf = 1;
        if (!rf) { done = 1; flag = 1; }
        else if (k >= d->k_max) { done = 1; flag = -1; }
    }
    for (int j = 0; j < m; j++) u_opt[j] = M2(z1, 0, n + j);
    *k_out = k;
    *e_flag = flag;
    if (z1_out) memcpy(z1_out, z1, sizeof(double) * (size_t)(N + 1) * nm);
    if (z3_out) memcpy(z3_out, z3, sizeof(double) * (size_t)(N + 1) * nm);
    if (z2_out) memcpy(z2_out, z2, sizeof(double) * (size_t)nm);
    if (lam_out) {
        memset(lam_out, 0, sizeof(double) * (size_t)(N + 3) * nm);
        size_t c = 0;
        for (int l = 0; l < N + 3; l++)
            for (int j = 0; j < n; j++) lam_out[c++] = M2(lam, l, j);
    }
    free(x0); free(z1); free(z3); free(z3p); free(z2); free(z2p); free(q2); free(lam); free(res); free(mu);
    return 0;
}

int oracle_eadmm_mpct_batch(const eadmm_mpct_data *d, long B, const double *x0, const double *xr, const double *ur,
                            int ref_stride, double *u, int *k, int *e_flag, double *z1, double *z2, double *z3,
                            double *lam) {
    const size_t nm = (size_t)(d->n + d->m), dz = (size_t)(d->N + 1) * nm, dl = (size_t)(d->N + 3) * nm;
    for (long i = 0; i < B; i++) {
        const double *xri = ref_stride ? xr + (size_t)i * d->n : xr;
        const double *uri = ref_stride ? ur + (size_t)i * d->m : ur;
        int rc = oracle_eadmm_mpct_solve(d, x0 + (size_t)i * d->n, xri, uri, u + (size_t)i * d->m, k + i, e_flag + i,
                                         z1 ? z1 + (size_t)i * dz : NULL, z2 ? z2 + (size_t)i * nm : NULL,
                                         z3 ? z3 + (size_t)i * dz : NULL, lam ? lam + (size_t)i * dl : NULL);
        if (rc) return rc;
    }
    return 0;
}
