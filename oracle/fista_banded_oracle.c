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

/* ------------------------------------------------------------------------------------------------
 * TIME_VARYING == 1: the update phase that computes the solver's ingredients from the model handed in with
 * every call (code_laxMPC_FISTA_C.c:117-262, code_equMPC_FISTA_C.c:113-248), then the same iteration.
 * Inputs as the 9-argument gateway: A [n][n] and B [n][m] COLUMN-major, Q [n], R [m] diagonals, LB / UB [n+m].
 * T (negated diagonal) and Ti = -1/diag(T) are the only controller constants (cons_laxMPC_FISTA_C.m:94-108).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    double *AB, *Alpha, *Beta, *Q, *R, *QRi; /* outputs, sized as in fista_banded_data */
} fista_tv_out;

void oracle_fista_tv_update(int n, int m, int N, int terminal, const double *A_in, const double *B_in, const double *Q_in,
                            const double *R_in, const double *Ti, fista_tv_out *o) {
    const int nm = n + m;
    double *Q_i = (double *)calloc((size_t)n, sizeof(double));
    double *R_i = (double *)calloc((size_t)m, sizeof(double));
    double *AQiAt = (double *)calloc((size_t)n * n, sizeof(double));
    double *BRiBt = (double *)calloc((size_t)n * n, sizeof(double));
#define TAB(i, j) (o->AB[(size_t)(i) * nm + (j)])
#define TALPHA(l, i, j) (o->Alpha[((size_t)(l) * n + (i)) * n + (j)])
#define TBETA(l, i, j) (o->Beta[((size_t)(l) * n + (i)) * n + (j)])
    memset(o->Alpha, 0, sizeof(double) * (size_t)(N - 1) * n * n);
    memset(o->Beta, 0, sizeof(double) * (size_t)N * n * n);
    /* :107-133 */
    for (int i = 0; i < n; i++) {
        o->Q[i] = Q_in[i];
        Q_i[i] = 1 / (o->Q[i]);
        for (int j = 0; j < n; j++) TAB(i, j) = A_in[i + j * n];
        for (int j = 0; j < m; j++) TAB(i, n + j) = B_in[i + j * n];
    }
    for (int j = 0; j < m; j++) {
        o->R[j] = R_in[j];
        R_i[j] = 1 / (o->R[j]);
    }
    for (int i = 0; i < nm; i++) o->QRi[i] = (i < n) ? -Q_i[i] : -R_i[i - n];
    /* :146-155 */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            for (int k = 0; k < n; k++) AQiAt[i * n + j] += A_in[i + k * n] * Q_i[k] * A_in[j + k * n];
            for (int k = 0; k < m; k++) BRiBt[i * n + j] += B_in[i + k * n] * R_i[k] * B_in[j + k * n];
        }
    /* Beta{0} :158-176 (the equMPC template also runs j below i: those entries come out zero and are never read) */
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            TBETA(0, i, j) = BRiBt[i * n + j];
            for (int l = 1; l <= i; l++) TBETA(0, i, j) -= TBETA(0, l - 1, i) * TBETA(0, l - 1, j);
            if (i == j) {
                TBETA(0, i, i) += Q_i[i];
                TBETA(0, i, i) = 1 / sqrt(TBETA(0, i, i));
            } else {
                TBETA(0, i, j) = TBETA(0, i, j) * TBETA(0, i, i);
            }
        }
    for (int h = 0; h < N - 1; h++) {
        if (h >= 1) { /* Beta{h} :194-219 */
            for (int i = 0; i < n; i++)
                for (int j = i; j < n; j++) {
                    TBETA(h, i, j) = AQiAt[i * n + j] + BRiBt[i * n + j];
                    for (int k = 0; k < n; k++) TBETA(h, i, j) -= TALPHA(h - 1, k, i) * TALPHA(h - 1, k, j);
                    for (int l = 1; l <= i; l++) TBETA(h, i, j) -= TBETA(h, l - 1, i) * TBETA(h, l - 1, j);
                    if (i == j) {
                        TBETA(h, i, i) += Q_i[i];
                        TBETA(h, i, i) = 1 / sqrt(TBETA(h, i, i));
                    } else {
                        TBETA(h, i, j) = TBETA(h, i, j) * TBETA(h, i, i);
                    }
                }
        }
        /* Alpha{h} :179-191, :223-236 */
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                TALPHA(h, i, j) = -Q_i[i] * TAB(j, i);
                for (int l = 1; l <= i; l++) TALPHA(h, i, j) -= TBETA(h, l - 1, i) * TALPHA(h, l - 1, j);
                TALPHA(h, i, j) = TALPHA(h, i, j) * TBETA(h, i, i);
            }
    }
    /* Beta{N-1} :241-262 (laxMPC: the diagonal terminal weight enters through Ti = -1/diag(T)) */
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            TBETA(N - 1, i, j) = AQiAt[i * n + j] + BRiBt[i * n + j];
            for (int k = 0; k < n; k++) TBETA(N - 1, i, j) -= TALPHA(N - 2, k, i) * TALPHA(N - 2, k, j);
            for (int l = 1; l <= i; l++) TBETA(N - 1, i, j) -= TBETA(N - 1, l - 1, i) * TBETA(N - 1, l - 1, j);
            if (i == j) {
                if (terminal) TBETA(N - 1, i, i) -= Ti[i];
                TBETA(N - 1, i, i) = 1 / sqrt(TBETA(N - 1, i, i));
            } else {
                TBETA(N - 1, i, j) = TBETA(N - 1, i, j) * TBETA(N - 1, i, i);
            }
        }
    /* :266-271 */
    for (int i = 0; i < n; i++) o->Q[i] = -o->Q[i];
    for (int i = 0; i < m; i++) o->R[i] = -o->R[i];
#undef TAB
#undef TALPHA
#undef TBETA
    free(Q_i); free(R_i); free(AQiAt); free(BRiBt);
}

/* Batch driver of the time-varying solver.  model is [B][n*n + n*m + n + m + 2(n+m)] = (A, B, Q, R, LB, UB) per instance
 * when model_stride != 0, else one shared model.  fac_out (optional): instance 0's Alpha then Beta. */
int oracle_fista_tv_batch(int n, int m, int N, int k_max, int terminal, double tol, const double *T, const double *Ti, long B,
                          const double *x0, const double *xr, const double *ur, int ref_stride, const double *model,
                          int model_stride, double *u, int *k, int *e_flag, double *z, double *lam, double *fac_out) {
    const int nm = n + m;
    const size_t dim = (size_t)N * nm - (terminal ? 0 : (size_t)n);
    const size_t msz = (size_t)n * n + (size_t)n * m + n + m + 2 * (size_t)nm;
    fista_tv_out o;
    o.AB = (double *)calloc((size_t)n * nm, sizeof(double));
    o.Alpha = (double *)calloc((size_t)(N - 1) * n * n, sizeof(double));
    o.Beta = (double *)calloc((size_t)N * n * n, sizeof(double));
    o.Q = (double *)calloc((size_t)n, sizeof(double));
    o.R = (double *)calloc((size_t)m, sizeof(double));
    o.QRi = (double *)calloc((size_t)nm, sizeof(double));
    int rc = 0;
    for (long i = 0; i < B && !rc; i++) {
        const double *mi = model_stride ? model + (size_t)i * msz : model;
        const double *A_in = mi, *B_in = A_in + (size_t)n * n, *Q_in = B_in + (size_t)n * m, *R_in = Q_in + n,
                     *LB = R_in + m, *UB = LB + nm;
        oracle_fista_tv_update(n, m, N, terminal, A_in, B_in, Q_in, R_in, Ti, &o);
        if (i == 0 && fac_out) {
            memcpy(fac_out, o.Alpha, sizeof(double) * (size_t)(N - 1) * n * n);
            memcpy(fac_out + (size_t)(N - 1) * n * n, o.Beta, sizeof(double) * (size_t)N * n * n);
        }
        fista_banded_data d;
        d.n = n; d.m = m; d.N = N; d.k_max = k_max; d.terminal = terminal; d.tol = tol;
        d.AB = o.AB; d.Alpha = o.Alpha; d.Beta = o.Beta; d.Q = o.Q; d.R = o.R; d.QRi = o.QRi; d.T = T; d.Ti = Ti;
        d.LB = LB; d.UB = UB;
        const double *xri = ref_stride ? xr + (size_t)i * n : xr;
        const double *uri = ref_stride ? ur + (size_t)i * m : ur;
        rc = oracle_fista_banded_solve(&d, x0 + (size_t)i * n, xri, uri, u + (size_t)i * m, k + i, e_flag + i,
                                       z ? z + (size_t)i * dim : NULL, lam ? lam + (size_t)i * (size_t)N * n : NULL);
    }
    free(o.AB); free(o.Alpha); free(o.Beta); free(o.Q); free(o.R); free(o.QRi);
    return rc;
}
