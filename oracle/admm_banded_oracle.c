/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement (plain C, FP64, one instance per call, runtime dimensions) of the ADMM solver
 * the reference generates for the laxMPC and equMPC formulations:
 *
 *   formulations/+laxMPC/code_laxMPC_ADMM_C.c:21-695   (terminal block present, `terminal = 1`)
 *   formulations/+equMPC/code_equMPC_ADMM_C.c:21-616   (no terminal block,      `terminal = 0`)
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.  The
 * floating-point operation order of every accumulation follows the reference loop nests (cited per
 * function) so that, fed the same constants, it reproduces the generated C bit for bit when built
 * without FMA contraction (the Makefile passes -ffp-contract=off).
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against the reference tests' hard-coded
 * optimum (tests/test_laxMPC_ADMM.m:35, tests/test_equMPC_ADMM.m:33; tolerance 1e-4 as
 * tests/spcies_tester.m:261) and against tests/golden/ vectors.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int n, m, N;          /* states, inputs, horizon                                      */
    int k_max;            /* iteration cap                                                 */
    int terminal;         /* 1 = laxMPC (x_N is a variable, cost T), 0 = equMPC (x_N = xr) */
    double tol, rho, rho_i;
    const double *AB;     /* [n][n+m] row-major, [A B]                                     */
    const double *Alpha;  /* [N-1][n][n]                                                   */
    const double *Beta;   /* [N][n][n], upper triangle, diagonal holds 1/diag              */
    const double *Hi;     /* [N-1][n+m] inverse diagonal of H+rho*I, middle stages         */
    const double *Hi_0;   /* [m]                                                           */
    const double *Hi_N;   /* [n][n] dense inverse of T+rho*I (terminal only)               */
    const double *Q, *R;  /* NEGATED diagonals of Q, R ([n], [m])                          */
    const double *T;      /* [n][n] NEGATED terminal cost (terminal only)                  */
    const double *LB, *UB;/* [n+m] state bounds first, then input bounds                   */
    /* ellipMPC ADMM (formulations/+ellipMPC/code_ellipMPC_ADMM_C.c, `ellip = 1`): terminal ellipsoid
     * (x_N - c)' P (x_N - c) <= r^2 imposed by a P-projection, stage-wise bounds                     */
    int ellip;
    const double *P, *P_half, *Pinv_half; /* [n][n]                                          */
    const double *c;                      /* [n]                                             */
    double r;
    const double *LBz, *UBz;              /* [N-1][n+m]                                      */
    const double *LBu0, *UBu0;            /* [m]                                             */
    /* lax/equ MPC switches (code_laxMPC_ADMM_C.c:323-348, 490-568): NULL = scalar rho / constant bounds       */
    const double *rho_0, *rho_v, *rho_N;       /* vector rho: [m], [N-1][n+m], [n]   (no SCALAR_RHO)          */
    const double *rho_i_0, *rho_i_v, *rho_i_N; /* their printed reciprocals                                    */
    const double *LB0, *UB0, *LBN, *UBN;       /* VAR_BOUNDS: [m], [n] and LB / UB become [N-1][n+m]          */
} admm_banded_data;
#define RHO_H(j) (d->rho_0 ? d->rho_0[j] : d->rho)
#define RHO_M(l, j) (d->rho_v ? d->rho_v[(size_t)(l) * nm + (j)] : d->rho)
#define RHO_T(j) (d->rho_N ? d->rho_N[j] : d->rho)
#define RHOI_H(j) (d->rho_i_0 ? d->rho_i_0[j] : d->rho_i)
#define RHOI_M(l, j) (d->rho_i_v ? d->rho_i_v[(size_t)(l) * nm + (j)] : d->rho_i)
#define RHOI_T(j) (d->rho_i_N ? d->rho_i_N[j] : d->rho_i)

/* Workspace layout mirrors the reference's split of every vector into a `_0` head (m inputs of
 * stage 0), N-1 middle rows of n+m and a `_N` tail (n terminal states).                     */
typedef struct {
    double *h;   /* [m]          */
    double *mid; /* [N-1][n+m]   */
    double *t;   /* [n]          */
} split_vec;

static void split_alloc(split_vec *s, int n, int m, int N) {
    s->h = (double *)calloc((size_t)m, sizeof(double));
    s->mid = (double *)calloc((size_t)(N - 1) * (size_t)(n + m), sizeof(double));
    s->t = (double *)calloc((size_t)n, sizeof(double));
}
static void split_free(split_vec *s) { free(s->h); free(s->mid); free(s->t); }

#define ABij(i, j) (d->AB[(size_t)(i) * nm + (j)])
#define ALPHA(l, i, j) (d->Alpha[((size_t)(l) * n + (i)) * n + (j)])
#define BETA(l, i, j) (d->Beta[((size_t)(l) * n + (i)) * n + (j)])
#define HI(l, j) (d->Hi[(size_t)(l) * nm + (j)])
#define MID(s, l, j) ((s).mid[(size_t)(l) * nm + (j)])
#define MU(l, j) (mu[(size_t)(l) * n + (j)])

/* q_hat = q + lambda - rho*v, written into z (code_laxMPC_ADMM_C.c:323-349). */
static void form_qhat(const admm_banded_data *d, const double *q, const double *qT,
                      const split_vec *lam, const split_vec *v, split_vec *z) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    for (int j = 0; j < m; j++) z->h[j] = q[n + j] + lam->h[j] - RHO_H(j) * v->h[j];
    for (int l = 0; l < N - 1; l++)
        for (int j = 0; j < nm; j++) MID(*z, l, j) = q[j] + MID(*lam, l, j) - RHO_M(l, j) * MID(*v, l, j);
    if (d->ellip) { /* code_ellipMPC_ADMM_C.c:146-156 */
        for (int j = 0; j < n; j++) {
            z->t[j] = qT[j];
            for (int i = 0; i < n; i++)
                z->t[j] = z->t[j] + d->P_half[(size_t)j * n + i] * lam->t[i] - d->P[(size_t)j * n + i] * RHO_T(i) * v->t[i];
        }
    } else if (d->terminal)
        for (int j = 0; j < n; j++) z->t[j] = qT[j] + lam->t[j] - RHO_T(j) * v->t[j];
}

/* Right-hand side  -G*Hhat^{-1}*q_hat - b  of the W system, stored in mu
 * (code_laxMPC_ADMM_C.c:355-381; equMPC: code_equMPC_ADMM_C.c:337-352 with the `- xr` line). */
static void form_rhs(const admm_banded_data *d, const double *b, const double *xr,
                     const split_vec *z, double *mu) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    for (int j = 0; j < n; j++) {
        double acc = HI(0, j) * MID(*z, 0, j) - b[j];
        for (int i = 0; i < m; i++) acc = acc - ABij(j, n + i) * d->Hi_0[i] * z->h[i];
        MU(0, j) = acc;
    }
    for (int l = 1; l < N - 1; l++)
        for (int j = 0; j < n; j++) {
            double acc = HI(l, j) * MID(*z, l, j);
            for (int i = 0; i < nm; i++) acc = acc - ABij(j, i) * HI(l - 1, i) * MID(*z, l - 1, i);
            MU(l, j) = acc;
        }
    for (int j = 0; j < n; j++) {
        double acc = 0.0;
        if (d->terminal)
            for (int i = 0; i < n; i++) acc = acc + d->Hi_N[(size_t)j * n + i] * z->t[i];
        for (int i = 0; i < nm; i++) acc = acc - ABij(j, i) * HI(N - 2, i) * MID(*z, N - 2, i);
        if (!d->terminal) acc = acc - xr[j];
        MU(N - 1, j) = acc;
    }
}

/* mu <- W^{-1} mu through the block-bidiagonal Cholesky factor (Alpha / Beta):
 * forward substitution code_laxMPC_ADMM_C.c:388-417, backward :422-451. */
static void solve_W(const admm_banded_data *d, double *mu) {
    const int n = d->n, N = d->N;
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
}

/* z = -Hhat^{-1} (q_hat + G' mu)   (code_laxMPC_ADMM_C.c:456-485). */
static void form_z(const admm_banded_data *d, const double *mu, split_vec *z, double *aux) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    for (int j = 0; j < m; j++) {
        double acc = z->h[j];
        for (int i = 0; i < n; i++) acc = acc + ABij(i, n + j) * MU(0, i);
        z->h[j] = -d->Hi_0[j] * acc;
    }
    for (int l = 0; l < N - 1; l++) {
        for (int j = 0; j < n; j++) MID(*z, l, j) = MID(*z, l, j) - MU(l, j);
        for (int j = 0; j < nm; j++) {
            double acc = MID(*z, l, j);
            for (int i = 0; i < n; i++) acc = acc + ABij(i, j) * MU(l + 1, i);
            MID(*z, l, j) = -HI(l, j) * acc;
        }
    }
    if (d->terminal) {
        for (int j = 0; j < n; j++) aux[j] = z->t[j] - MU(N - 1, j);
        for (int j = 0; j < n; j++) {
            double acc = 0.0;
            for (int i = 0; i < n; i++) acc = acc - d->Hi_N[(size_t)j * n + i] * aux[i];
            z->t[j] = acc;
        }
    }
}

static inline double clampd(double x, double lo, double hi) {
    x = (x > lo) ? x : lo; /* same comparison sense as the reference (:500-501) */
    x = (x > hi) ? hi : x;
    return x;
}

/* v = clamp(z + lambda/rho), lambda += rho (z - v)   (code_laxMPC_ADMM_C.c:490-568). */
/* ellipMPC: stage-wise boxes, P-projection of the terminal block, P_half in its dual update
 * (code_ellipMPC_ADMM_C.c:292-386). */
static void update_v_lambda_ellip(const admm_banded_data *d, const split_vec *z, split_vec *v, split_vec *lam, double *aux) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    for (int j = 0; j < m; j++) v->h[j] = clampd(z->h[j] + RHOI_H(j) * lam->h[j], d->LBu0[j], d->UBu0[j]);
    for (int l = 0; l < N - 1; l++)
        for (int j = 0; j < nm; j++)
            MID(*v, l, j) = clampd(MID(*z, l, j) + RHOI_M(l, j) * MID(*lam, l, j), d->LBz[(size_t)l * nm + j], d->UBz[(size_t)l * nm + j]);
    for (int j = 0; j < n; j++) {
        v->t[j] = z->t[j];
        for (int i = 0; i < n; i++) v->t[j] = v->t[j] + d->Pinv_half[(size_t)j * n + i] * RHOI_T(i) * lam->t[i];
    }
    for (int j = 0; j < n; j++) {
        aux[j] = 0.0;
        for (int i = 0; i < n; i++) aux[j] = aux[j] + d->P[(size_t)j * n + i] * (v->t[i] - d->c[i]);
    }
    double vPv = 0.0;
    for (int j = 0; j < n; j++) vPv = vPv + (v->t[j] - d->c[j]) * aux[j];
    if (vPv > d->r * d->r) {
        vPv = d->r / sqrt(vPv);
        for (int j = 0; j < n; j++) v->t[j] = vPv * (v->t[j] - d->c[j]) + d->c[j];
    }
    for (int j = 0; j < m; j++) lam->h[j] = lam->h[j] + RHO_H(j) * (z->h[j] - v->h[j]);
    for (int l = 0; l < N - 1; l++)
        for (int j = 0; j < nm; j++)
            MID(*lam, l, j) = MID(*lam, l, j) + RHO_M(l, j) * (MID(*z, l, j) - MID(*v, l, j));
    for (int j = 0; j < n; j++) aux[j] = RHO_T(j) * (z->t[j] - v->t[j]);
    for (int j = 0; j < n; j++)
        for (int i = 0; i < n; i++) lam->t[j] = lam->t[j] + d->P_half[(size_t)j * n + i] * aux[i];
}

static void update_v_lambda(const admm_banded_data *d, const split_vec *z, split_vec *v, split_vec *lam) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    const int vb = d->LB0 != NULL; /* VAR_BOUNDS */
    for (int j = 0; j < m; j++)
        v->h[j] = clampd(z->h[j] + RHOI_H(j) * lam->h[j], vb ? d->LB0[j] : d->LB[n + j], vb ? d->UB0[j] : d->UB[n + j]);
    for (int l = 0; l < N - 1; l++)
        for (int j = 0; j < nm; j++)
            MID(*v, l, j) = clampd(MID(*z, l, j) + RHOI_M(l, j) * MID(*lam, l, j), vb ? d->LB[(size_t)l * nm + j] : d->LB[j],
                                   vb ? d->UB[(size_t)l * nm + j] : d->UB[j]);
    if (d->terminal)
        for (int j = 0; j < n; j++)
            v->t[j] = clampd(z->t[j] + RHOI_T(j) * lam->t[j], vb ? d->LBN[j] : d->LB[j], vb ? d->UBN[j] : d->UB[j]);

    for (int j = 0; j < m; j++) lam->h[j] = lam->h[j] + RHO_H(j) * (z->h[j] - v->h[j]);
    for (int l = 0; l < N - 1; l++)
        for (int j = 0; j < nm; j++)
            MID(*lam, l, j) = MID(*lam, l, j) + RHO_M(l, j) * (MID(*z, l, j) - MID(*v, l, j));
    if (d->terminal)
        for (int j = 0; j < n; j++) lam->t[j] = lam->t[j] + RHO_T(j) * (z->t[j] - v->t[j]);
}

static inline int exceeds(double a, double b, double tol) {
    double r1 = a - b;
    r1 = (r1 > 0.0) ? r1 : -r1;
    return r1 > tol;
}

/* 1 if any |v1 - v| or |z - v| component is above tol (code_laxMPC_ADMM_C.c:572-620). */
static int residual_flag(const admm_banded_data *d, const split_vec *z, const split_vec *v, const split_vec *v1) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    for (int j = 0; j < m; j++)
        if (exceeds(v1->h[j], v->h[j], d->tol) || exceeds(z->h[j], v->h[j], d->tol)) return 1;
    if (d->terminal)
        for (int j = 0; j < n; j++)
            if (exceeds(v1->t[j], v->t[j], d->tol) || exceeds(z->t[j], v->t[j], d->tol)) return 1;
    for (int l = 0; l < N - 1; l++)
        for (int j = 0; j < nm; j++)
            if (exceeds(MID(*v1, l, j), MID(*v, l, j), d->tol) || exceeds(MID(*z, l, j), MID(*v, l, j), d->tol))
                return 1;
    return 0;
}

static void flatten(const admm_banded_data *d, const split_vec *s, double *out) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    size_t c = 0;
    for (int j = 0; j < m; j++) out[c++] = s->h[j];
    for (size_t i = 0; i < (size_t)(N - 1) * nm; i++) out[c++] = s->mid[i];
    if (d->terminal)
        for (int j = 0; j < n; j++) out[c++] = s->t[j];
}

/* One solve.  z_out / v_out / lam_out (each N*(n+m) [- n for equMPC] doubles) may be NULL. */
int oracle_admm_banded_solve(const admm_banded_data *d, const double *x0, const double *xr, const double *ur,
                             double *u_opt, int *k_out, int *e_flag, double *z_out, double *v_out,
                             double *lam_out) {
    const int n = d->n, m = d->m, nm = n + m, N = d->N;
    if (n <= 0 || m <= 0 || N < 2) return -1;
    split_vec z, v, v1, lam;
    split_alloc(&z, n, m, N);
    split_alloc(&v, n, m, N);
    split_alloc(&v1, n, m, N);
    split_alloc(&lam, n, m, N);
    double *mu = (double *)calloc((size_t)N * n, sizeof(double));
    double *b = (double *)calloc((size_t)n, sizeof(double));
    double *q = (double *)calloc((size_t)nm, sizeof(double));
    double *qT = (double *)calloc((size_t)n, sizeof(double));
    double *aux = (double *)calloc((size_t)n, sizeof(double));

    /* per-instance setup (code_laxMPC_ADMM_C.c:282-299) */
    for (int j = 0; j < n; j++) {
        b[j] = 0.0;
        for (int i = 0; i < n; i++) b[j] = b[j] - ABij(j, i) * x0[i];
    }
    for (int j = 0; j < n; j++) {
        q[j] = d->Q[j] * xr[j];
        qT[j] = 0.0;
        if (d->terminal)
            for (int i = 0; i < n; i++) qT[j] = qT[j] + d->T[(size_t)j * n + i] * xr[i];
    }
    for (int j = 0; j < m; j++) q[n + j] = d->R[j] * ur[j];

    int k = 0, done = 0, flag = -1;
    while (!done) {
        k += 1;
        memcpy(v1.h, v.h, sizeof(double) * (size_t)m);
        memcpy(v1.mid, v.mid, sizeof(double) * (size_t)(N - 1) * nm);
        memcpy(v1.t, v.t, sizeof(double) * (size_t)n);
        form_qhat(d, q, qT, &lam, &v, &z);
        form_rhs(d, b, xr, &z, mu);
        solve_W(d, mu);
        form_z(d, mu, &z, aux);
        if (d->ellip) update_v_lambda_ellip(d, &z, &v, &lam, aux);
        else update_v_lambda(d, &z, &v, &lam);
        if (!residual_flag(d, &z, &v, &v1)) {
            done = 1;
            flag = 1;
        } else if (k >= d->k_max) {
            done = 1;
            flag = -1;
        }
    }
    for (int j = 0; j < m; j++) u_opt[j] = v.h[j];
    *k_out = k;
    *e_flag = flag;
    if (z_out) flatten(d, &z, z_out);
    if (v_out) flatten(d, &v, v_out);
    if (lam_out) flatten(d, &lam, lam_out);

    split_free(&z); split_free(&v); split_free(&v1); split_free(&lam);
    free(mu); free(b); free(q); free(qT); free(aux);
    return 0;
}

/* Batch driver (instances are independent; used for fixtures and for the timed CPU baseline).
 * x0 is [B][n]; xr/ur are [B][n]/[B][m] when ref_stride != 0, else a single shared reference.
 * Outputs u [B][m], k [B], e_flag [B]; z/v/lam [B][dim] or NULL. */
int oracle_admm_banded_batch_mt(const admm_banded_data *d, long B, const double *x0, const double *xr, const double *ur,
                                int ref_stride, double *u, int *k, int *e_flag, double *z, double *v, double *lam,
                                int threads);

int oracle_admm_banded_batch(const admm_banded_data *d, long B, const double *x0, const double *xr,
                             const double *ur, int ref_stride, double *u, int *k, int *e_flag, double *z,
                             double *v, double *lam) {
    return oracle_admm_banded_batch_mt(d, B, x0, xr, ur, ref_stride, u, k, e_flag, z, v, lam, 1);
}

/* the same loop over instances spread over `threads` host threads (OpenMP; instances are independent, every instance
 * is still solved by the scalar code above): the multi-core CPU baseline of bench.py */
int oracle_admm_banded_batch_mt(const admm_banded_data *d, long B, const double *x0, const double *xr, const double *ur,
                                int ref_stride, double *u, int *k, int *e_flag, double *z, double *v, double *lam,
                                int threads) {
    const size_t dim = (size_t)d->N * (size_t)(d->n + d->m) - (d->terminal ? 0 : (size_t)d->n);
    int bad = 0;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 1 ? threads : 1) if (threads > 1)
    for (long i = 0; i < B; i++) {
        const double *xri = ref_stride ? xr + (size_t)i * d->n : xr;
        const double *uri = ref_stride ? ur + (size_t)i * d->m : ur;
        int rc = oracle_admm_banded_solve(d, x0 + (size_t)i * d->n, xri, uri, u + (size_t)i * d->m, k + i,
                                          e_flag + i, z ? z + (size_t)i * dim : NULL,
                                          v ? v + (size_t)i * dim : NULL, lam ? lam + (size_t)i * dim : NULL);
        if (rc) {
#pragma omp atomic write
            bad = rc;
        }
    }
    return bad;
}

/* ------------------------------------------------------------------------------------------------
 * TIME_VARYING == 1: the update phase that computes the solver's ingredients from the model handed in
 * with every call (code_laxMPC_ADMM_C.c:117-279, code_equMPC_ADMM_C.c:117-265), then the same iteration.
 * Inputs as the 9-argument mex gateway (struct_laxMPC_ADMM_C_Matlab.c:57-103): A [n][n] and B [n][m]
 * COLUMN-major, Q [n], R [m] diagonals, LB / UB [n+m].  T_rho_i = inv(T + rho I) and T (negated) are the
 * only controller constants (cons_laxMPC_ADMM_C.m:107-109).
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    double *AB, *Alpha, *Beta, *Hi, *Hi_0, *Q, *R; /* outputs, sized as in admm_banded_data */
} admm_tv_out;

void oracle_admm_tv_update(int n, int m, int N, int terminal, double rho, const double *A_in, const double *B_in,
                           const double *Q_in, const double *R_in, const double *T_rho_i, admm_tv_out *o) {
    const int nm = n + m;
    double *Q_rho_i = (double *)calloc((size_t)n, sizeof(double));
    double *R_rho_i = (double *)calloc((size_t)m, sizeof(double));
    double *AQiAt = (double *)calloc((size_t)n * n, sizeof(double));
    double *BRiBt = (double *)calloc((size_t)n * n, sizeof(double));
#define TAB(i, j) (o->AB[(size_t)(i) * nm + (j)])
#define TALPHA(l, i, j) (o->Alpha[((size_t)(l) * n + (i)) * n + (j)])
#define TBETA(l, i, j) (o->Beta[((size_t)(l) * n + (i)) * n + (j)])
    memset(o->Alpha, 0, sizeof(double) * (size_t)(N - 1) * n * n);
    memset(o->Beta, 0, sizeof(double) * (size_t)N * n * n);
    /* :117-145 */
    for (int i = 0; i < n; i++) {
        o->Q[i] = Q_in[i];
        Q_rho_i[i] = 1 / (o->Q[i] + rho);
        for (int j = 0; j < n; j++) TAB(i, j) = A_in[i + j * n];
        for (int j = 0; j < m; j++) TAB(i, n + j) = B_in[i + j * n];
    }
    for (int j = 0; j < m; j++) {
        o->R[j] = R_in[j];
        R_rho_i[j] = 1 / (o->R[j] + rho);
        o->Hi_0[j] = R_rho_i[j];
    }
    for (int i = 0; i < N - 1; i++)
        for (int j = 0; j < nm; j++) o->Hi[(size_t)i * nm + j] = (j < n) ? Q_rho_i[j] : R_rho_i[j - n];
    /* :151-161 */
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            for (int k = 0; k < n; k++) AQiAt[i * n + j] += A_in[i + k * n] * Q_rho_i[k] * A_in[j + k * n];
            for (int k = 0; k < m; k++) BRiBt[i * n + j] += B_in[i + k * n] * R_rho_i[k] * B_in[j + k * n];
        }
    /* Beta{0} :164-181, Alpha{0} :184-197 */
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            TBETA(0, i, j) = BRiBt[i * n + j];
            for (int l = 1; l <= i; l++) TBETA(0, i, j) -= TBETA(0, l - 1, i) * TBETA(0, l - 1, j);
            if (i == j) {
                TBETA(0, i, i) += Q_rho_i[i];
                TBETA(0, i, i) = 1 / sqrt(TBETA(0, i, i));
            } else {
                TBETA(0, i, j) = TBETA(0, i, j) * TBETA(0, i, i);
            }
        }
    for (int h = 0; h < N - 1; h++) {
        if (h >= 1) { /* Beta{h} :200-222 */
            for (int i = 0; i < n; i++)
                for (int j = i; j < n; j++) {
                    TBETA(h, i, j) = AQiAt[i * n + j] + BRiBt[i * n + j];
                    for (int k = 0; k < n; k++) TBETA(h, i, j) -= TALPHA(h - 1, k, i) * TALPHA(h - 1, k, j);
                    for (int l = 1; l <= i; l++) TBETA(h, i, j) -= TBETA(h, l - 1, i) * TBETA(h, l - 1, j);
                    if (i == j) {
                        TBETA(h, i, i) += Q_rho_i[i];
                        TBETA(h, i, i) = 1 / sqrt(TBETA(h, i, i));
                    } else {
                        TBETA(h, i, j) = TBETA(h, i, j) * TBETA(h, i, i);
                    }
                }
        }
        /* Alpha{h} :184-197, :225-238 */
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                TALPHA(h, i, j) = -Q_rho_i[i] * TAB(j, i);
                for (int l = 1; l <= i; l++) TALPHA(h, i, j) -= TBETA(h, l - 1, i) * TALPHA(h, l - 1, j);
                TALPHA(h, i, j) = TALPHA(h, i, j) * TBETA(h, i, i);
            }
    }
    /* Beta{N-1} :244-267 (laxMPC adds the dense T_rho_i; equMPC: code_equMPC_ADMM_C.c:234-255) */
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) {
            TBETA(N - 1, i, j) = AQiAt[i * n + j] + BRiBt[i * n + j];
            for (int k = 0; k < n; k++) TBETA(N - 1, i, j) -= TALPHA(N - 2, k, i) * TALPHA(N - 2, k, j);
            for (int l = 1; l <= i; l++) TBETA(N - 1, i, j) -= TBETA(N - 1, l - 1, i) * TBETA(N - 1, l - 1, j);
            if (terminal) TBETA(N - 1, i, j) += T_rho_i[(size_t)i * n + j];
            if (i == j) TBETA(N - 1, i, i) = 1 / sqrt(TBETA(N - 1, i, i));
            else TBETA(N - 1, i, j) = TBETA(N - 1, i, j) * TBETA(N - 1, i, i);
        }
    /* :271-276 */
    for (int i = 0; i < n; i++) o->Q[i] = -o->Q[i];
    for (int i = 0; i < m; i++) o->R[i] = -o->R[i];
#undef TAB
#undef TALPHA
#undef TBETA
    free(Q_rho_i); free(R_rho_i); free(AQiAt); free(BRiBt);
}

/* Batch driver of the time-varying solver.  model is [B][n*n + n*m + n + m + 2(n+m)] = (A, B, Q, R, LB, UB) per
 * instance when model_stride != 0, else one shared model.  fac_out (optional) receives instance 0's
 * Alpha ((N-1) n n) then Beta (N n n) for the tests. */
int oracle_admm_tv_batch(int n, int m, int N, int k_max, int terminal, double tol, double rho, const double *T,
                         const double *T_rho_i, long B, const double *x0, const double *xr, const double *ur,
                         int ref_stride, const double *model, int model_stride, double *u, int *k, int *e_flag,
                         double *z, double *v, double *lam, double *fac_out) {
    const int nm = n + m;
    const size_t dim = (size_t)N * nm - (terminal ? 0 : (size_t)n);
    const size_t msz = (size_t)n * n + (size_t)n * m + n + m + 2 * (size_t)nm;
    admm_tv_out o;
    o.AB = (double *)calloc((size_t)n * nm, sizeof(double));
    o.Alpha = (double *)calloc((size_t)(N - 1) * n * n, sizeof(double));
    o.Beta = (double *)calloc((size_t)N * n * n, sizeof(double));
    o.Hi = (double *)calloc((size_t)(N - 1) * nm, sizeof(double));
    o.Hi_0 = (double *)calloc((size_t)m, sizeof(double));
    o.Q = (double *)calloc((size_t)n, sizeof(double));
    o.R = (double *)calloc((size_t)m, sizeof(double));
    int rc = 0;
    for (long i = 0; i < B && !rc; i++) {
        const double *mi = model_stride ? model + (size_t)i * msz : model;
        const double *A_in = mi, *B_in = A_in + (size_t)n * n, *Q_in = B_in + (size_t)n * m, *R_in = Q_in + n,
                     *LB = R_in + m, *UB = LB + nm;
        oracle_admm_tv_update(n, m, N, terminal, rho, A_in, B_in, Q_in, R_in, T_rho_i, &o);
        if (i == 0 && fac_out) {
            memcpy(fac_out, o.Alpha, sizeof(double) * (size_t)(N - 1) * n * n);
            memcpy(fac_out + (size_t)(N - 1) * n * n, o.Beta, sizeof(double) * (size_t)N * n * n);
        }
        admm_banded_data d;
        d.n = n; d.m = m; d.N = N; d.k_max = k_max; d.terminal = terminal; d.tol = tol; d.rho = rho; d.rho_i = 1.0 / rho;
        d.AB = o.AB; d.Alpha = o.Alpha; d.Beta = o.Beta; d.Hi = o.Hi; d.Hi_0 = o.Hi_0; d.Hi_N = T_rho_i;
        d.Q = o.Q; d.R = o.R; d.T = T; d.LB = LB; d.UB = UB;
        d.ellip = 0;
        d.rho_0 = d.rho_v = d.rho_N = d.rho_i_0 = d.rho_i_v = d.rho_i_N = NULL;
        d.LB0 = d.UB0 = d.LBN = d.UBN = NULL;
        const double *xri = ref_stride ? xr + (size_t)i * n : xr;
        const double *uri = ref_stride ? ur + (size_t)i * m : ur;
        rc = oracle_admm_banded_solve(&d, x0 + (size_t)i * n, xri, uri, u + (size_t)i * m, k + i, e_flag + i,
                                      z ? z + (size_t)i * dim : NULL, v ? v + (size_t)i * dim : NULL,
                                      lam ? lam + (size_t)i * dim : NULL);
    }
    free(o.AB); free(o.Alpha); free(o.Beta); free(o.Hi); free(o.Hi_0); free(o.Q); free(o.R);
    return rc;
}
