/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, FP64, one instance per call, runtime dimensions) of the FISTA solver the
 * reference generates for the laxMPC and equMPC formulations:
 *
 *   formulations/+laxMPC/code_laxMPC_FISTA_C.c:21-651   (`terminal = 1`)
 *   formulations/+equMPC/code_equMPC_FISTA_C.c:21-632   (`terminal = 0`)
 *
 * Dual fast-gradient iteration: z(y) = clamp(H^-1 (q - G'y)), residual r = b - G z, d = W^-1 r through
 * the banded Cholesky factor (Alpha / Beta), lambda = y + d, Nesterov step on y.  Accumulation order
 * follows the reference loop nests (cited per function); built with -ffp-contract=off.
 *
 * Parity pin: tests/test_oracle_golden.py (reference tests' z_opt of tests/test_laxMPC_FISTA.m:34 and
 * tests/test_equMPC_FISTA.m:32, tolerance 1e-4) and tests/golden/template_*FISTA*.npz (bit-exact).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n, m, N, k_max, terminal;
    double tol;
    const double *AB;     /* [n][n+m]                                   */
    const double *Alpha;  /* [N-1][n][n]                                */
    const double *Beta;   /* [N][n][n] upper triangle, inverted diagonal */
    const double *Q, *R;  /* negated diagonals                          */
    const double *QRi;    /* [n+m]  -1/diag([Q, R])                     */
    const double *T, *Ti; /* [n] -diag(T), -1/diag(T) (terminal only)   */
    const double *LB, *UB;/* [n+m]                                      */
} fista_banded_data;

#define ABij(i, j) (d->AB[(size_t)(i) * nm + (j)])
#define ALPHA(l, i, j) (d->Alpha[((size_t)(l) * n + (i)) * n + (j)])
#define BETA(l, i, j) (d->Beta[((size_t)(l) * n + (i)) * n + (j)])
#define Zm(l, j) (z_mid[(size_t)(l) * nm + (j)])
#define V(a, l, j) ((a)[(size_t)(l) * n + (j)])

static inline double clampd(double x, double lo, double hi) {
    x = (x > lo) ? x : lo;
    x = (x > hi) ? hi : x;
    return x;
}

/* z(lam) = clamp(H^-1 (q - G' lam))   (code_laxMPC_FISTA_C.c:471-539) */
static void z_of_lambda(const fista_banded_data *d, const double *lam, const double *q, const double *qT,
                        double *z_0, double *z_mid, double *z_N) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    for (int j = 0; j < m; j++) {
        double acc = q[n + j];
        for (int i = 0; i < n; i++) acc = acc - ABij(i, n + j) * V(lam, 0, i);
        acc = acc * d->QRi[n + j];
        z_0[j] = clampd(acc, d->LB[n + j], d->UB[n + j]);
    }
    for (int l = 0; l < N - 1; l++) {
        for (int j = 0; j < nm; j++) {
            double acc = q[j];
            for (int i = 0; i < n; i++) acc = acc - ABij(i, j) * V(lam, l + 1, i);
            Zm(l, j) = acc;
        }
        for (int j = 0; j < n; j++) Zm(l, j) = Zm(l, j) + V(lam, l, j);
        for (int j = 0; j < nm; j++) Zm(l, j) = clampd(Zm(l, j) * d->QRi[j], d->LB[j], d->UB[j]);
    }
    if (d->terminal)
        for (int j = 0; j < n; j++) {
            double acc = qT[j] + V(lam, N - 1, j);
            acc = acc * d->Ti[j];
            z_N[j] = clampd(acc, d->LB[j], d->UB[j]);
        }
}

/* r = b - G z   (code_laxMPC_FISTA_C.c:546-574; equMPC: last block starts from xr, code_equMPC_FISTA_C.c:549) */
static void residual(const fista_banded_data *d, const double *z_0, const double *z_mid, const double *z_N,
                     const double *b, const double *xr, double *r) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    for (int j = 0; j < n; j++) {
        double acc = b[j] + Zm(0, j);
        for (int i = 0; i < m; i++) acc = acc - ABij(j, n + i) * z_0[i];
        V(r, 0, j) = acc;
    }
    for (int l = 1; l < N - 1; l++)
        for (int j = 0; j < n; j++) {
            double acc = Zm(l, j);
            for (int i = 0; i < nm; i++) acc = acc - ABij(j, i) * Zm(l - 1, i);
            V(r, l, j) = acc;
        }
    for (int j = 0; j < n; j++) {
        double acc = d->terminal ? z_N[j] : xr[j];
        for (int i = 0; i < nm; i++) acc = acc - ABij(j, i) * Zm(N - 2, i);
        V(r, N - 1, j) = acc;
    }
}

/* mu <- W^-1 mu  (solve_W_matrix_form, code_laxMPC_FISTA_C.c:577-651) */
static void solve_W(const fista_banded_data *d, double *mu) {
    const int n = d->n, N = d->N;
    for (int l = 0; l < N; l++)
        for (int j = 0; j < n; j++) {
            double acc = V(mu, l, j);
            if (l > 0)
                for (int i = 0; i < n; i++) acc = acc - ALPHA(l - 1, i, j) * V(mu, l - 1, i);
            for (int i = 0; i < j; i++) acc = acc - BETA(l, i, j) * V(mu, l, i);
            V(mu, l, j) = BETA(l, j, j) * acc;
        }
    for (int l = N - 1; l >= 0; l--)
        for (int j = n - 1; j >= 0; j--) {
            double acc = V(mu, l, j);
            if (l < N - 1)
                for (int i = n - 1; i >= 0; i--) acc = acc - ALPHA(l, j, i) * V(mu, l + 1, i);
            for (int i = n - 1; i > j; i--) acc = acc - BETA(l, j, i) * V(mu, l, i);
            V(mu, l, j) = BETA(l, j, j) * acc;
        }
}

/* One solve.  z_out: N*(n+m) [- n for equMPC] doubles, lam_out: N*n doubles (= y, :439-445); may be NULL. */
int oracle_fista_banded_solve(const fista_banded_data *d, const double *x0, const double *xr, const double *ur,
                              double *u_opt, int *k_out, int *e_flag, double *z_out, double *lam_out) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    if (n <= 0 || m <= 0 || N < 2) return -1;
    const size_t Nn = (size_t)N * n;
    double *z_0 = (double *)calloc((size_t)m, sizeof(double));
    double *z_mid = (double *)calloc((size_t)(N - 1) * nm, sizeof(double));
    double *z_N = (double *)calloc((size_t)n, sizeof(double));
    double *y = (double *)calloc(Nn, sizeof(double)), *lam = (double *)calloc(Nn, sizeof(double));
    double *lam1 = (double *)calloc(Nn, sizeof(double)), *dl = (double *)calloc(Nn, sizeof(double));
    double *b = (double *)calloc((size_t)n, sizeof(double)), *q = (double *)calloc((size_t)nm, sizeof(double));
    double *qT = (double *)calloc((size_t)n, sizeof(double));
    double t = 1.0, t1 = 1.0;

    for (int j = 0; j < n; j++) {
        b[j] = 0.0;
        for (int i = 0; i < n; i++) b[j] = b[j] - ABij(j, i) * x0[i];
    }
    for (int j = 0; j < n; j++) {
        q[j] = d->Q[j] * xr[j];
        qT[j] = d->terminal ? d->T[j] * xr[j] : 0.0;
    }
    for (int j = 0; j < m; j++) q[n + j] = d->R[j] * ur[j];

    /* initial step (:296-318) */
    z_of_lambda(d, lam, q, qT, z_0, z_mid, z_N);
    residual(d, z_0, z_mid, z_N, b, xr, dl);
    solve_W(d, dl);
    for (size_t i = 0; i < Nn; i++) lam[i] = lam[i] + dl[i];
    for (size_t i = 0; i < Nn; i++) y[i] = lam[i];

    int k = 0, done = 0, flag = -1;
    while (!done) {
        k += 1;
        memcpy(lam1, lam, sizeof(double) * Nn);
        t1 = t;
        z_of_lambda(d, y, q, qT, z_0, z_mid, z_N);
        residual(d, z_0, z_mid, z_N, b, xr, dl);
        int res_flag = 0;
        for (size_t i = 0; i < Nn; i++) {
            double r = dl[i];
            r = (r > 0.0) ? r : -r;
            if (r > d->tol) { res_flag = 1; break; }
        }
        if (!res_flag) { done = 1; flag = 1; }
        else if (k >= d->k_max) { done = 1; flag = -1; }
        if (!done) {
            solve_W(d, dl);
            for (size_t i = 0; i < Nn; i++) lam[i] = y[i] + dl[i];
            t = 0.5 * (1 + sqrt(1 + 4 * t1 * t1));
            for (size_t i = 0; i < Nn; i++) y[i] = lam[i] + (t1 - 1) * (lam[i] - lam1[i]) / t;
        }
    }
    for (int j = 0; j < m; j++) u_opt[j] = z_0[j];
    *k_out = k;
    *e_flag = flag;
    if (z_out) {
        size_t c = 0;
        for (int j = 0; j < m; j++) z_out[c++] = z_0[j];
        for (size_t i = 0; i < (size_t)(N - 1) * nm; i++) z_out[c++] = z_mid[i];
        if (d->terminal)
            for (int j = 0; j < n; j++) z_out[c++] = z_N[j];
    }
    if (lam_out) memcpy(lam_out, y, sizeof(double) * Nn);
    free(z_0); free(z_mid); free(z_N); free(y); free(lam); free(lam1); free(dl); free(b); free(q); free(qT);
    return 0;
}

int oracle_fista_banded_batch(const fista_banded_data *d, long B, const double *x0, const double *xr,
                              const double *ur, int ref_stride, double *u, int *k, int *e_flag, double *z,
                              double *lam) {
    const size_t dim = (size_t)d->N * (size_t)(d->n + d->m) - (d->terminal ? 0 : (size_t)d->n);
    const size_t Nn = (size_t)d->N * d->n;
    for (long i = 0; i < B; i++) {
        const double *xri = ref_stride ? xr + (size_t)i * d->n : xr;
        const double *uri = ref_stride ? ur + (size_t)i * d->m : ur;
        int rc = oracle_fista_banded_solve(d, x0 + (size_t)i * d->n, xri, uri, u + (size_t)i * d->m, k + i, e_flag + i,
                                           z ? z + (size_t)i * dim : NULL, lam ? lam + (size_t)i * Nn : NULL);
        if (rc) return rc;
    }
    return 0;
}
