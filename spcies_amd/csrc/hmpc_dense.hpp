// HMPC ADMM / SADMM WITHOUT the splitting - the reference's default HMPC solver (code_HMPC_ADMM_C.c:18-310) - for a
// batch.  The z-update of the reference is dense,  z = M2 b + M1 (q + C'(rho (s - d) + lambda))  (:123-157); here it is
// ONE matrix product per iteration for the whole batch,
//     Z [B x dim] = T [B x n_s] * (M1 C')' ,     T = rho (s - d) + lambda ,     z = Z + CI ,   CI = M2 b + M1 q
// (CI is constant over the iterations: one small product per instance at setup), a plain library GEMM (rocBLAS dgemm
// through the run-time binding of hmpc_gemm.hpp) with two small kernels around it:
//   * post  : one thread per (instance, chunk of rows of s): C z - d, symmetric half step, s = box / cone projection,
//             dual step, residual flags, T of the next iteration (:161-256);
//   * finish: one thread per instance: exit test (:258-268), k / e_flag, the record's z of the last iteration.
// State in structure-of-arrays form [row][B_pad]:  T | Z | CI | S | LAM | ZF (record) | QC (the 2n + m non-zero rows of q).
// Sums run in another order than the reference's loops (M1 C' is formed on the host, the product is blocked):
// parity with the oracle to 1e-10, iteration counts equal except where a residual sits within rounding of the tolerance.
// Variant STREAM (dense_stream_kernel below) is the reference-faithful form: one lane per instance, the reference's loops
// in their order without FMA contraction -> bit-identical; M1 is read through scalar loads, eight rows per pass.
#pragma once
#include "hmpc_gemm.hpp"

namespace spcies {
namespace hdense {

#pragma clang fp contract(fast)

struct Dev {  // offsets (doubles) into the constants allocation / (ints) into the index allocation, dimensions, scalars
    int M2xA, M1Q, QQ, Te, Se, LB, UB, LBy, UBy, Cval, dvec, M1, M2, A, Ctval;
    int Crow, Ccol, Ctrow, Ctcol;
    int n, m, N, dim, n_s, n_box, n_soc, k_max, use_soc, symmetric;
    int ks, ds;  // rows of T and of Z in the scratch = n_s and dim rounded up to a multiple of 32: the GEMM has no edge tiles
    double tol_p, tol_d, rho, rho_i, alpha;
};
constexpr int CHUNK = 24;  // rows of s per thread of the post kernel (a multiple of 3)

struct Host {  // what the blob carries (cons_HMPC_ADMM_C.m:88-131)
    int n = 0, m = 0, N = 0, dim = 0, n_s = 0, n_box = 0, n_soc = 0, n_eq = 0, k_max = 0, use_soc = 0, symmetric = 0;
    double tol_p = 0, tol_d = 0, rho = 0, rho_i = 0, alpha = 1;
    std::vector<double> A, QQ, Te, Se, LB, UB, LBy, UBy, d, C_val, Ct_val, M1, M2;
    std::vector<int> C_col, C_row, Ct_col, Ct_row;
};

struct Plan {
    bool ok = false;
    std::string why = "not built";
    Dev dev{};
    double *d_G1 = nullptr, *d_C = nullptr;  // G1 = M1 C' row-major [dim][n_s] (= its transpose column-major); constants
    int *d_I = nullptr;
    hgemm::RocBlas blas;
};
inline void plan_free(Plan &p) {
    if (p.d_G1) hipFree(p.d_G1);
    if (p.d_C) hipFree(p.d_C);
    if (p.d_I) hipFree(p.d_I);
    p.blas.close();
    p.d_G1 = p.d_C = nullptr;
    p.d_I = nullptr;
}

inline int plan_build(Plan &p, const Host &h) {
    const int n = h.n, m = h.m, nm = n + m, dim = h.dim, n_s = h.n_s;
    std::vector<double> flat;
    auto put = [&](const double *src, size_t cnt) {
        const int off = (int)flat.size();
        flat.insert(flat.end(), src, src + cnt);
        while (flat.size() % 8) flat.push_back(0.0);
        return off;
    };
    // CI = M2 b + M1 q = (-M2 A) x0 + M1[:, rows of q] qc   (b = -A x0, :83-88; q: :91-105)
    const int nq = 2 * n + m, q0 = (h.N - 1) * nm + m;
    std::vector<double> M2xA((size_t)dim * n, 0.0), M1Q((size_t)dim * nq, 0.0);
    for (int i = 0; i < dim; i++) {
        for (int c = 0; c < n; c++) {
            double acc = 0.0;
            for (int j = 0; j < n; j++) acc -= h.M2[(size_t)i * n + j] * h.A[(size_t)j * n + c];
            M2xA[(size_t)i * n + c] = acc;
        }
        for (int r = 0; r < nq; r++) {
            const int col = q0 + (r < n ? r : (r < 2 * n ? 2 * n + (r - n) : 3 * n + (r - 2 * n)));
            M1Q[(size_t)i * nq + r] = h.M1[(size_t)i * dim + col];
        }
    }
    // G1 = M1 C'  [dim][n_s], from the CSR form of C' the reference multiplies with (:130-136)
    std::vector<double> G1((size_t)dim * n_s, 0.0);
    for (int i = 0; i < dim; i++)
        for (int j = 0; j < dim; j++) {
            const double mij = h.M1[(size_t)i * dim + j];
            for (int q = h.Ct_row[j]; q < h.Ct_row[j + 1]; q++) G1[(size_t)i * n_s + h.Ct_col[q]] += mij * h.Ct_val[q];
        }
    for (double x : G1)
        if (!std::isfinite(x)) { p.why = "non-finite M1"; return 0; }
    Dev d{};
    int pad_to = 32;
    if (const char *ev = getenv("SPCIES_GEMM_PAD")) pad_to = std::max(1, atoi(ev));
    d.ks = (n_s + pad_to - 1) / pad_to * pad_to;
    d.ds = (dim + pad_to - 1) / pad_to * pad_to;
    std::vector<double> G1p((size_t)d.ds * d.ks, 0.0);
    for (int i = 0; i < dim; i++)
        for (int j = 0; j < n_s; j++) G1p[(size_t)i * d.ks + j] = G1[(size_t)i * n_s + j];
    d.M2xA = put(M2xA.data(), M2xA.size());
    d.M1Q = put(M1Q.data(), M1Q.size());
    d.QQ = put(h.QQ.data(), h.QQ.size());
    d.Te = put(h.Te.data(), h.Te.size());
    d.Se = put(h.Se.data(), h.Se.size());
    d.LB = put(h.LB.data(), h.LB.size());
    d.UB = put(h.UB.data(), h.UB.size());
    d.LBy = put(h.LBy.data(), h.LBy.size());
    d.UBy = put(h.UBy.data(), h.UBy.size());
    d.Cval = put(h.C_val.data(), h.C_val.size());
    std::vector<double> dv = h.d;
    if (dv.empty()) dv.assign(n_s, 0.0);
    d.dvec = put(dv.data(), dv.size());
    d.M1 = put(h.M1.data(), h.M1.size());  // STREAM variant: the reference's own arrays
    d.M2 = put(h.M2.data(), h.M2.size());
    d.A = put(h.A.data(), h.A.size());
    d.Ctval = put(h.Ct_val.data(), h.Ct_val.size());
    std::vector<int> idx(h.C_row);
    d.Crow = 0;
    d.Ccol = (int)idx.size();
    idx.insert(idx.end(), h.C_col.begin(), h.C_col.end());
    d.Ctrow = (int)idx.size();
    idx.insert(idx.end(), h.Ct_row.begin(), h.Ct_row.end());
    d.Ctcol = (int)idx.size();
    idx.insert(idx.end(), h.Ct_col.begin(), h.Ct_col.end());
    d.n = n; d.m = m; d.N = h.N; d.dim = dim; d.n_s = n_s; d.n_box = h.n_box; d.n_soc = h.n_soc; d.k_max = h.k_max;
    d.use_soc = h.use_soc; d.symmetric = h.symmetric;
    d.tol_p = h.tol_p; d.tol_d = h.tol_d; d.rho = h.rho; d.rho_i = h.rho_i; d.alpha = h.alpha;
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_G1, G1p.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_G1, G1p.data(), G1p.size() * sizeof(double), hipMemcpyHostToDevice));
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_C, flat.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_C, flat.data(), flat.size() * sizeof(double), hipMemcpyHostToDevice));
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_I, idx.size() * sizeof(int)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_I, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
    p.dev = d;
    p.ok = true;
    p.why.clear();
    return 0;
}

// scratch rows: T (n_s) | Z (dim) | CI (dim) | S (n_s) | LAM (n_s) | ZF (dim) | QC (2n + m);  ints: RES, ACT [Bp], NACT
struct Rows {
    long T, Z, CI, S, LAM, ZF, QC, total;
};
__host__ __device__ inline Rows rows_of(const Dev &d) {
    Rows r;
    r.T = 0; r.Z = r.T + d.ks; r.CI = r.Z + d.ds; r.S = r.CI + d.dim; r.LAM = r.S + d.n_s; r.ZF = r.LAM + d.n_s;
    r.QC = r.ZF + d.dim; r.total = r.QC + 2 * d.n + d.m;
    return r;
}
inline size_t scratch_bytes(const Dev &d, long B) {
    const long Bp = (B + 63) / 64 * 64;
    return (size_t)rows_of(d).total * Bp * sizeof(double) + 2 * (size_t)Bp * sizeof(int) + 64;
}

// setup (:83-105): zero state, q, CI = M2 b + M1 q, T of iteration 1 = -rho d
__global__ __launch_bounds__(64) void setup_kernel(Dev d, const double *__restrict__ C, const double *__restrict__ x0g,
                                                   const double *__restrict__ xrg, const double *__restrict__ urg, int ref_stride,
                                                   long B, long Bp, double *__restrict__ Sc, int *__restrict__ RES,
                                                   int *__restrict__ ACT) {
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= Bp) return;
    const int n = d.n, m = d.m, nq = 2 * n + m;
    const Rows R = rows_of(d);
#define AT(row0, i) Sc[((row0) + (long)(i)) * Bp + t]
    RES[t] = 0;
    ACT[t] = (t < B) ? 1 : 0;
    const long ti = (t < B) ? t : 0;
    const double *x0 = x0g + ti * n, *xr = ref_stride ? xrg + ti * n : xrg, *ur = ref_stride ? urg + ti * m : urg;
    const double *cQQ = C + d.QQ, *cTe = C + d.Te, *cSe = C + d.Se, *cMA = C + d.M2xA, *cMQ = C + d.M1Q, *cd = C + d.dvec;
    for (int j = 0; j < n; j++) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < n; i++) {
            a -= cTe[j * n + i] * xr[i] + cQQ[j * n + i] * x0[i];
            b -= cQQ[j * n + i] * x0[i];
        }
        AT(R.QC, j) = a;
        AT(R.QC, n + j) = b;
    }
    for (int j = 0; j < m; j++) {
        double a = 0.0;
        for (int i = 0; i < m; i++) a -= cSe[j * m + i] * ur[i];
        AT(R.QC, 2 * n + j) = a;
    }
    for (int j = 0; j < d.dim; j++) {
        double a = 0.0;
        for (int i = 0; i < n; i++) a += cMA[(long)j * n + i] * x0[i];
        for (int r = 0; r < nq; r++) a += cMQ[(long)j * nq + r] * AT(R.QC, r);
        AT(R.CI, j) = a;
        AT(R.ZF, j) = 0.0;
    }
    for (int i = 0; i < d.n_s; i++) {
        AT(R.S, i) = 0.0;
        AT(R.LAM, i) = 0.0;
        AT(R.T, i) = d.use_soc ? -d.rho * cd[i] : 0.0;
    }
    for (int i = d.n_s; i < d.ks; i++) AT(R.T, i) = 0.0;  // pad rows of the GEMM operand stay zero
#undef AT
}

// one thread per (instance, chunk of rows of s): everything between two products (:161-256)
__global__ __launch_bounds__(256) void post_kernel(Dev d, const double *__restrict__ C, const int *__restrict__ I, long Bp,
                                                   double *__restrict__ Sc, int *__restrict__ RES, const int *__restrict__ ACT,
                                                   int box_chunks) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= Bp || !ACT[t]) return;
    const Rows R = rows_of(d);
#define AT(row0, i) Sc[((row0) + (long)(i)) * Bp + t]
    const double *cLB = C + d.LB, *cUB = C + d.UB, *cLBy = C + d.LBy, *cUBy = C + d.UBy, *cCv = C + d.Cval, *cd = C + d.dvec;
    const int *Crow = I + d.Crow, *Ccol = I + d.Ccol;
    const double rho = d.rho, rho_i = d.rho_i, ar = d.alpha * d.rho, g = d.symmetric ? ar : rho;
    bool res = false;
    // C z - d for row i
    auto cz_row = [&](int i) {
        double a = d.use_soc ? -cd[i] : 0.0;
        for (int q = Crow[i]; q < Crow[i + 1]; q++) {
            const int c = Ccol[q];
            a += cCv[q] * (AT(R.Z, c) + AT(R.CI, c));
        }
        return a;
    };
    auto finish_row = [&](int i, double cz, double lam, double so, double s) {
        cz += s;
        lam += g * cz;
        res |= (fabs(cz) > d.tol_p) | (fabs(s - so) > d.tol_d);
        AT(R.S, i) = s;
        AT(R.LAM, i) = lam;
        AT(R.T, i) = d.use_soc ? rho * (s - cd[i]) + lam : rho * s + lam;
    };
    if ((int)blockIdx.y < box_chunks) {
        const int r0 = blockIdx.y * CHUNK, r1 = min(r0 + CHUNK, d.n_box);
        for (int i = r0; i < r1; i++) {
            const double cz = cz_row(i), so = AT(R.S, i);
            double lam = AT(R.LAM, i);
            if (d.symmetric) lam += ar * (cz + so);
            double s = -cz - rho_i * lam;
            s = fmin(fmax(s, cLB[i]), cUB[i]);
            finish_row(i, cz, lam, so, s);
        }
    } else {
        const int c0 = ((int)blockIdx.y - box_chunks) * (CHUNK / 3), c1 = min(c0 + CHUNK / 3, d.n_soc);
        for (int j = c0; j < c1; j++) {
            double cz[3], so[3], lam[3], s[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int i = d.n_box + 3 * j + r;
                cz[r] = cz_row(i);
                so[r] = AT(R.S, i);
                lam[r] = AT(R.LAM, i);
                if (d.symmetric) lam[r] += ar * (cz[r] + so[r]);
                s[r] = -cz[r] - rho_i * lam[r];
            }
            if (d.use_soc) {
                proj_soc3(s[0], s[1], s[2], 1.0, 0.0);
            } else {
                proj_soc3(s[0], s[1], s[2], 1.0, cLBy[j]);
                proj_soc3(s[0], s[1], s[2], -1.0, cUBy[j]);
            }
#pragma unroll
            for (int r = 0; r < 3; r++) finish_row(d.n_box + 3 * j + r, cz[r], lam[r], so[r], s[r]);
        }
    }
    if (res) atomicOr(&RES[t], 1);
#undef AT
}

// one thread per instance: exit test (:258-268), k, e_flag, the record's z of the last iteration
__global__ __launch_bounds__(64) void finish_kernel(Dev d, int k, long Bp, double *__restrict__ Sc, int *__restrict__ RES,
                                                    int *__restrict__ ACT, int *__restrict__ k_out, int *__restrict__ e_out,
                                                    int *__restrict__ n_active) {
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= Bp || !ACT[t]) return;
    const int r = RES[t];
    RES[t] = 0;
    if (r && k < d.k_max) return;
    const Rows R = rows_of(d);
    for (int j = 0; j < d.dim; j++) Sc[(R.ZF + j) * Bp + t] = Sc[(R.Z + j) * Bp + t] + Sc[(R.CI + j) * Bp + t];
    k_out[t] = k;
    e_out[t] = r ? -1 : 1;
    ACT[t] = 0;
    atomicSub(n_active, 1);
}

// host loop.  u, k, e, fields are device pointers; fields = z, s, lambda (NULL entries skipped)
inline int launch(Plan &p, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *scratch,
                  double *u, int *k, int *e, double *const *f, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "GEMM variant unavailable: %s", p.why.c_str());
    int rc = p.blas.open();
    if (rc) return rc;
    const Dev &d = p.dev;
    const long Bp = (B + 63) / 64 * 64;
    const Rows R = rows_of(d);
    double *Sc = scratch;
    int *RES = reinterpret_cast<int *>(Sc + (size_t)R.total * Bp), *ACT = RES + Bp, *NACT = ACT + Bp;
    const int nact0 = (int)B;
    SPCIES_HIP_CHECK(hipMemcpyAsync(NACT, &nact0, sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(setup_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, st, d, p.d_C, x0, xr, ur, ref_stride, B, Bp, Sc, RES, ACT);
    SPCIES_HIP_CHECK(hipGetLastError());
    if (p.blas.set_stream(p.blas.handle, st) != 0) return fail(SPCIES_HIP_EHIP, "rocblas_set_stream failed");
    const double one = 1.0, zero = 0.0;
    const int box_chunks = (d.n_box + CHUNK - 1) / CHUNK, cone_chunks = (d.n_soc + CHUNK / 3 - 1) / (CHUNK / 3);
    const dim3 pgrid((unsigned)((Bp + 255) / 256), (unsigned)(box_chunks + cone_chunks));
    const bool can_stop_early = d.tol_p > 0 || d.tol_d > 0;
    for (int it = 1; it <= d.k_max; it++) {
        // Z [Bp x dim] = T [Bp x n_s] * G1'  (column-major operands: T ld = Bp; the row-major G1 [dim][n_s] IS G1' column-major)
        if (p.blas.dgemm(p.blas.handle, hgemm::ROCBLAS_OP_N, hgemm::ROCBLAS_OP_N, (int)Bp, d.ds, d.ks, &one, Sc + R.T * Bp, (int)Bp,
                         p.d_G1, d.ks, &zero, Sc + R.Z * Bp, (int)Bp) != 0)
            return fail(SPCIES_HIP_EHIP, "rocblas_dgemm failed");
        hipLaunchKernelGGL(post_kernel, pgrid, dim3(256), 0, st, d, p.d_C, p.d_I, Bp, Sc, RES, ACT, box_chunks);
        hipLaunchKernelGGL(finish_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, st, d, it, Bp, Sc, RES, ACT, k, e, NACT);
        if (can_stop_early && (it % 16 == 0)) {
            int left = 0;
            SPCIES_HIP_CHECK(hipMemcpyAsync(&left, NACT, sizeof(int), hipMemcpyDeviceToHost, st));
            SPCIES_HIP_CHECK(hipStreamSynchronize(st));
            if (left <= 0) break;
        }
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    {
        dim3 tg((unsigned)(Bp / 64), (unsigned)((d.m + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, Sc + R.ZF * Bp, Bp, B, d.m, u);
    }
    const long src_row[3] = {R.ZF, R.S, R.LAM};
    const int src_rows[3] = {d.dim, d.n_s, d.n_s};
    for (int i = 0; i < 3; i++) {
        if (!f[i]) continue;
        dim3 tg((unsigned)(Bp / 64), (unsigned)((src_rows[i] + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, Sc + src_row[i] * Bp, Bp, B, src_rows[i], f[i]);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

#pragma clang fp contract(off)

// ---- variant STREAM: one lane per instance, reference operation order (code_HMPC_ADMM_C.c:82-268), bit-identical.
// scratch rows: Z (dim) | QH (dim) | S (n_s) | LAM (n_s) | CZ (n_s) | QV (dim) | BV (n)
inline size_t stream_scratch_bytes(const Dev &d, long B) {
    const long Bp = (B + 63) / 64 * 64;
    return (size_t)(3L * d.dim + 3L * d.n_s + d.n) * Bp * sizeof(double);
}
__global__ __launch_bounds__(64) void dense_stream_kernel(Dev d, const double *__restrict__ C, const int *__restrict__ I,
                                                          const double *__restrict__ x0g, const double *__restrict__ xrg,
                                                          const double *__restrict__ urg, int ref_stride, long B, long Bp,
                                                          double *__restrict__ Sc, double *__restrict__ u_out,
                                                          int *__restrict__ k_out, int *__restrict__ e_out) {
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= B) return;
    const int n = d.n, m = d.m, nm = n + m, N = d.N, dim = d.dim, n_s = d.n_s;
    double *Z = Sc + t, *QH = Z + (long)dim * Bp, *S = QH + (long)dim * Bp, *LAM = S + (long)n_s * Bp, *CZ = LAM + (long)n_s * Bp,
           *QV = CZ + (long)n_s * Bp, *BV = QV + (long)dim * Bp;
#define AT(P, i) (P)[(long)(i) * Bp]
    const double *x0 = x0g + t * n, *xr = ref_stride ? xrg + t * n : xrg, *ur = ref_stride ? urg + t * m : urg;
    const double *cA = C + d.A, *cQQ = C + d.QQ, *cTe = C + d.Te, *cSe = C + d.Se, *cLB = C + d.LB, *cUB = C + d.UB, *cLBy = C + d.LBy,
                 *cUBy = C + d.UBy, *cCv = C + d.Cval, *cCtv = C + d.Ctval, *cd = C + d.dvec, *cM1 = C + d.M1, *cM2 = C + d.M2;
    const int *Crow = I + d.Crow, *Ccol = I + d.Ccol, *Ctrow = I + d.Ctrow, *Ctcol = I + d.Ctcol;
    // setup (:82-105)
    for (int j = 0; j < n_s; j++) {
        AT(S, j) = 0.0;
        AT(LAM, j) = 0.0;
    }
    for (int j = 0; j < dim; j++) AT(QV, j) = 0.0;
    for (int j = 0; j < n; j++) {
        double b = 0.0;
        for (int i = 0; i < n; i++) b -= cA[j * n + i] * x0[i];
        AT(BV, j) = b;
    }
    const int q0 = (N - 1) * nm + m;
    for (int j = 0; j < n; j++) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < n; i++) {
            a -= cTe[j * n + i] * xr[i] + cQQ[j * n + i] * x0[i];
            b -= cQQ[j * n + i] * x0[i];
        }
        AT(QV, q0 + j) = a;
        AT(QV, q0 + 2 * n + j) = b;
    }
    for (int j = 0; j < m; j++) {
        double a = 0.0;
        for (int i = 0; i < m; i++) a -= cSe[j * m + i] * ur[i];
        AT(QV, q0 + 3 * n + j) = a;
    }
    const double rho = d.rho, rho_i = d.rho_i, ar = d.alpha * d.rho;
    int k = 0, flag = -1;
    while (true) {
        k += 1;
        // q_hat = q + C'(rho (s - d) + lambda)  (:123-137)
        for (int i = 0; i < n_s; i++) AT(CZ, i) = d.use_soc ? rho * (AT(S, i) - cd[i]) + AT(LAM, i) : rho * AT(S, i) + AT(LAM, i);
        for (int i = 0; i < dim; i++) {
            double a = AT(QV, i);
            for (int j = Ctrow[i]; j < Ctrow[i + 1]; j++) a += cCtv[j] * AT(CZ, Ctcol[j]);
            AT(QH, i) = a;
        }
        // z = M2 b + M1 q_hat  (:145-157): eight rows per pass over q_hat, every row summed in the reference's order
        for (int i0 = 0; i0 < dim; i0 += 8) {
            double acc[8];
#pragma unroll
            for (int r = 0; r < 8; r++) acc[r] = 0.0;
            for (int j = 0; j < n; j++) {
                const double bj = AT(BV, j);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int i = min(i0 + r, dim - 1);
                    acc[r] += cM2[(long)i * n + j] * bj;
                }
            }
            int j = 0;
            for (; j + 8 <= dim; j += 8) {  // eight entries of q_hat loaded before the first is used (one load in flight otherwise)
                double qv[8];
#pragma unroll
                for (int u = 0; u < 8; u++) qv[u] = AT(QH, j + u);
#pragma unroll
                for (int u = 0; u < 8; u++) {
#pragma unroll
                    for (int r = 0; r < 8; r++) {
                        const int i = min(i0 + r, dim - 1);
                        acc[r] += cM1[(long)i * dim + j + u] * qv[u];
                    }
                }
            }
            for (; j < dim; j++) {
                const double qj = AT(QH, j);
#pragma unroll
                for (int r = 0; r < 8; r++) {
                    const int i = min(i0 + r, dim - 1);
                    acc[r] += cM1[(long)i * dim + j] * qj;
                }
            }
#pragma unroll
            for (int r = 0; r < 8; r++)
                if (i0 + r < dim) AT(Z, i0 + r) = acc[r];
        }
        // C z - d (:161-170), symmetric half step (:174-180), s (:184-204), C z + s, lambda (:207-225), residuals (:229-245)
        bool res = false;
        auto cz_row = [&](int i) {
            double a = d.use_soc ? -cd[i] : 0.0;
            for (int j = Crow[i]; j < Crow[i + 1]; j++) a += cCv[j] * AT(Z, Ccol[j]);
            return a;
        };
        auto finish_row = [&](int i, double cz, double lam, double so, double s) {
            cz += s;
            lam += d.symmetric ? ar * cz : rho * cz;
            AT(S, i) = s;
            AT(LAM, i) = lam;
            res = res || (fabs(cz) > d.tol_p) || (fabs(s - so) > d.tol_d);
        };
        for (int i = 0; i < d.n_box; i++) {
            const double cz = cz_row(i), so = AT(S, i);
            double lam = AT(LAM, i);
            if (d.symmetric) lam += ar * (cz + so);
            double s = -cz - rho_i * lam;
            s = clamp_ref(s, cLB[i], cUB[i]);
            finish_row(i, cz, lam, so, s);
        }
        for (int j = 0; j < d.n_soc; j++) {
            double cz[3], so[3], lam[3], s[3];
            for (int r = 0; r < 3; r++) {
                const int i = d.n_box + 3 * j + r;
                cz[r] = cz_row(i);
                so[r] = AT(S, i);
                lam[r] = AT(LAM, i);
                if (d.symmetric) lam[r] += ar * (cz[r] + so[r]);
                s[r] = -cz[r] - rho_i * lam[r];
            }
            if (d.use_soc) {
                proj_soc3(s[0], s[1], s[2], 1.0, 0.0);
            } else {
                proj_soc3(s[0], s[1], s[2], 1.0, cLBy[j]);
                proj_soc3(s[0], s[1], s[2], -1.0, cUBy[j]);
            }
            for (int r = 0; r < 3; r++) finish_row(d.n_box + 3 * j + r, cz[r], lam[r], so[r], s[r]);
        }
        if (!res) {
            flag = 1;
            break;
        }
        if (k >= d.k_max) {
            flag = -1;
            break;
        }
    }
    for (int j = 0; j < m; j++) u_out[t * m + j] = AT(Z, j);
#undef AT
    k_out[t] = k;
    e_out[t] = flag;
}

inline int launch_stream(Plan &p, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *scratch,
                         double *u, int *k, int *e, double *const *f, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "HMPC solver unavailable: %s", p.why.c_str());
    const Dev &d = p.dev;
    const long Bp = (B + 63) / 64 * 64;
    hipLaunchKernelGGL(dense_stream_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, st, d, p.d_C, p.d_I, x0, xr, ur, ref_stride, B, Bp,
                       scratch, u, k, e);
    SPCIES_HIP_CHECK(hipGetLastError());
    const long src_row[3] = {0, 2L * d.dim, 2L * d.dim + d.n_s};  // z, s, lambda = Z | S | LAM
    const int src_rows[3] = {d.dim, d.n_s, d.n_s};
    for (int i = 0; i < 3; i++) {
        if (!f[i]) continue;
        dim3 tg((unsigned)(Bp / 64), (unsigned)((src_rows[i] + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, scratch + src_row[i] * Bp, Bp, B, src_rows[i], f[i]);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace hdense
}  // namespace spcies
