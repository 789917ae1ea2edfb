// Variant GEMM of the HMPC ADMM / SADMM split solver: the reference's NON_SPARSE path (its default option
// `sparse = false`, def_options_HMPC_ADMM.m:31) - primal_hat = M2 bh - M1 q_hat with the dense M1
// (code_HMPC_ADMM_split_C.c:174-190) - executed for the whole batch as ONE matrix product per iteration,
//     PH [B x np] = - QH [B x np] * M1' [np x np]           (np = dim + n_s; 282 at C5),
// a plain library GEMM (rocBLAS dgemm, loaded on first use), with two small kernels around it:
//   * post  : one thread per (instance, chunk of 24 rows): symmetric half step, z = box(z_hat + lambda / sigma),
//             s = cone(s_hat + mu / rho), duals, residual flags, and q_hat of the NEXT iteration;
//   * finish: one thread per instance: exit test, k / e_flag, the record's z_hat / s_hat of the last iteration.
// State in structure-of-arrays form [row][B_pad] (instances contiguous = the column-major operand layout of
// the product): PR = (z, s), DU = (lambda, mu), QH, PH, CI = M2 bh (per instance, constant over the iterations),
// ZH (record), QC (the 2n + m non-zero rows of q).  The L D L' factor of the sparse path holds 43 762 non-zeros
// at C5 (30 % dense); its two sweeps cost about what the dense product does, without the matrix pipe.
#pragma once
#include <dlfcn.h>

#include "hmpc_stream.hpp"

namespace spcies {
namespace hgemm {

#pragma clang fp contract(fast)

// ---- rocBLAS, bound at run time (no link-time dependency for users of the other solvers)
struct RocBlas {
    void *lib = nullptr, *handle = nullptr;
    int (*create)(void **) = nullptr;
    int (*destroy)(void *) = nullptr;
    int (*set_stream)(void *, hipStream_t) = nullptr;
    int (*dgemm)(void *, int, int, int, int, int, const double *, const double *, int, const double *, int, const double *,
                 double *, int) = nullptr;
    int open() {
        if (handle) return 0;
        for (const char *name : {"librocblas.so", "librocblas.so.5", "/opt/rocm/lib/librocblas.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) return fail(SPCIES_HIP_ENOSUP, "GEMM variant: cannot load librocblas.so (%s)", dlerror());
        create = (int (*)(void **))dlsym(lib, "rocblas_create_handle");
        destroy = (int (*)(void *))dlsym(lib, "rocblas_destroy_handle");
        set_stream = (int (*)(void *, hipStream_t))dlsym(lib, "rocblas_set_stream");
        dgemm = (decltype(dgemm))dlsym(lib, "rocblas_dgemm");
        if (!create || !destroy || !set_stream || !dgemm) return fail(SPCIES_HIP_ENOSUP, "GEMM variant: rocBLAS symbols missing");
        if (create(&handle) != 0) return fail(SPCIES_HIP_EHIP, "rocblas_create_handle failed");
        return 0;
    }
    void close() {
        if (handle && destroy) destroy(handle);
        handle = nullptr;
    }
};
constexpr int ROCBLAS_OP_N = 111;  // rocblas_operation_none

struct Dev {  // offsets (doubles) into the GEMM constants allocation, dimensions, scalars
    int M2xA, c_const, A, QQ, Te, Se, LB, UB, LBy, UBy;
    int n, m, N, dim, n_s, np, npad, n_soc, k_max, use_soc, symmetric;  // npad: rows per state vector = np rounded up (GEMM n, k)
    double tol_p, tol_d, rho, rho_i, sigma, sigma_i, alpha;
};
constexpr int CHUNK = 24;  // rows per thread of the post kernel (a multiple of 3: cone triples never straddle chunks)

struct Plan {
    bool ok = false;
    std::string why = "not built";
    Dev dev{};
    double *d_M1 = nullptr, *d_C = nullptr;  // M1 row-major [np][np] (= M1' column-major); the other constants
    RocBlas blas;
};
inline void plan_free(Plan &p) {
    if (p.d_M1) hipFree(p.d_M1);
    if (p.d_C) hipFree(p.d_C);
    p.blas.close();
    p.d_M1 = p.d_C = nullptr;
}

// M1 [np][np], M2 [np][n_eq + n_s], bh in natural order; the dense small matrices and bounds as in HmpcDev
inline int plan_build(Plan &p, const HmpcDev &h, const std::vector<double> &M1, const std::vector<double> &M2,
                      const std::vector<double> &bh_nat, const double *A, const double *QQ, const double *Te, const double *Se,
                      const double *LB, int n_lb, const double *UB, const double *LBy, const double *UBy) {
    const int n = h.n, m = h.m, nm = n + m, np = h.dim + h.n_s, nc = h.n_eq + h.n_s;
    if ((int)M1.size() != np * np || (int)M2.size() != np * nc || (int)bh_nat.size() != nc) { p.why = "M1 / M2 / bh missing"; return 0; }
    std::vector<double> flat;
    auto put = [&](const double *src, size_t cnt) {
        const int off = (int)flat.size();
        flat.insert(flat.end(), src, src + cnt);
        while (flat.size() % 8) flat.push_back(0.0);
        return off;
    };
    // c_inst = M2 bh_inst = c_const + (-M2[:, :n] A) x0   (bh_inst = bh with its first n rows replaced by -A x0, :97-104)
    std::vector<double> M2xA((size_t)np * n, 0.0), c_const(np, 0.0);
    for (int i = 0; i < np; i++) {
        for (int j = n; j < nc; j++) c_const[i] += M2[(size_t)i * nc + j] * bh_nat[j];
        for (int c = 0; c < n; c++) {
            double acc = 0.0;
            for (int j = 0; j < n; j++) acc -= M2[(size_t)i * nc + j] * A[j * n + c];
            M2xA[(size_t)i * n + c] = acc;
        }
    }
    Dev d{};
    d.M2xA = put(M2xA.data(), M2xA.size());
    d.c_const = put(c_const.data(), c_const.size());
    d.A = put(A, (size_t)n * n);
    d.QQ = put(QQ, (size_t)n * n);
    d.Te = put(Te, (size_t)n * n);
    d.Se = put(Se, (size_t)m * m);
    d.LB = put(LB, (size_t)n_lb);
    d.UB = put(UB, (size_t)n_lb);
    d.LBy = put(LBy, (size_t)nm);
    d.UBy = put(UBy, (size_t)nm);
    int pad_to = 32;
    if (const char *ev = getenv("SPCIES_GEMM_PAD")) pad_to = std::max(1, atoi(ev));
    d.npad = (np + pad_to - 1) / pad_to * pad_to;
    d.n = n; d.m = m; d.N = h.N; d.dim = h.dim; d.n_s = h.n_s; d.np = np; d.n_soc = h.n_soc; d.k_max = h.k_max;
    d.use_soc = h.use_soc; d.symmetric = h.symmetric;
    d.tol_p = h.tol_p; d.tol_d = h.tol_d; d.rho = h.rho; d.rho_i = h.rho_i; d.sigma = h.sigma; d.sigma_i = h.sigma_i; d.alpha = h.alpha;
    for (double x : M1)
        if (!std::isfinite(x)) { p.why = "non-finite M1"; return 0; }
    {  // M1 zero-padded to [npad][npad]: the product then has no edge tiles (n = k = 288 instead of 282 at C5)
        std::vector<double> M1p((size_t)d.npad * d.npad, 0.0);
        for (int i = 0; i < np; i++)
            for (int j = 0; j < np; j++) M1p[(size_t)i * d.npad + j] = M1[(size_t)i * np + j];
        SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_M1, M1p.size() * sizeof(double)));
        SPCIES_HIP_CHECK(hipMemcpy(p.d_M1, M1p.data(), M1p.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_C, flat.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_C, flat.data(), flat.size() * sizeof(double), hipMemcpyHostToDevice));
    p.dev = d;
    p.ok = true;
    p.why.clear();
    return 0;
}

// scratch rows: PR | DU | QH | PH | CI | ZH (np each) | QC (2n + m);  ints: RES [Bp], ACT [Bp]
inline size_t scratch_bytes(const Dev &d, long B) {
    const long Bp = (B + 63) / 64 * 64;
    return (size_t)(6 * d.npad + 2 * d.n + d.m) * Bp * sizeof(double) + 2 * (size_t)Bp * sizeof(int) + 64;
}

// row j of q (only 2n + m rows of q are non-zero, :110-129): index into QC, or -1
__device__ __forceinline__ int qc_row(const Dev &d, int j) {
    const int e = j - ((d.N - 1) * (d.n + d.m) + d.m);
    if (e >= 0 && e < d.n) return e;
    if (e >= 2 * d.n && e < 3 * d.n) return d.n + (e - 2 * d.n);
    if (e >= 3 * d.n && e < 3 * d.n + d.m) return 2 * d.n + (e - 3 * d.n);
    return -1;
}

// setup (:97-129): zero state, q, c_inst = M2 bh, q_hat of iteration 1 = -q
__global__ __launch_bounds__(64) void setup_kernel(Dev d, const double *__restrict__ C, const double *__restrict__ x0g,
                                                   const double *__restrict__ xrg, const double *__restrict__ urg, int ref_stride,
                                                   long B, long Bp, double *__restrict__ S, int *__restrict__ RES,
                                                   int *__restrict__ ACT) {
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= Bp) return;
    const int n = d.n, m = d.m, np = d.np, ns = d.npad;
    double *PR = S + t, *DU = PR + (long)ns * Bp, *QH = DU + (long)ns * Bp, *CI = QH + 2L * ns * Bp, *QC = CI + 2L * ns * Bp;
#define AT(P, i) (P)[(long)(i) * Bp]
    RES[t] = 0;
    ACT[t] = (t < B) ? 1 : 0;
    const long ti = (t < B) ? t : 0;
    const double *x0 = x0g + ti * n, *xr = ref_stride ? xrg + ti * n : xrg, *ur = ref_stride ? urg + ti * m : urg;
    const double *cQQ = C + d.QQ, *cTe = C + d.Te, *cSe = C + d.Se, *cM = C + d.M2xA, *cc = C + d.c_const;
    for (int j = 0; j < n; j++) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < n; i++) {
            a -= cTe[j * n + i] * xr[i] + cQQ[j * n + i] * x0[i];
            b -= cQQ[j * n + i] * x0[i];
        }
        AT(QC, j) = a;
        AT(QC, n + j) = b;
    }
    for (int j = 0; j < m; j++) {
        double a = 0.0;
        for (int i = 0; i < m; i++) a -= cSe[j * m + i] * ur[i];
        AT(QC, 2 * n + j) = a;
    }
    for (int j = 0; j < np; j++) {
        double a = cc[j];
        for (int i = 0; i < n; i++) a += cM[j * n + i] * x0[i];
        AT(CI, j) = a;
        AT(PR, j) = 0.0;
        AT(DU, j) = 0.0;
        const int qr = (j < d.dim) ? qc_row(d, j) : -1;
        AT(QH, j) = (qr >= 0) ? -AT(QC, qr) : 0.0;
    }
    for (int j = np; j < ns; j++) AT(QH, j) = 0.0;  // pad rows of the GEMM operand stay zero
#undef AT
}

// one thread per (instance, chunk of CHUNK rows): everything between two products (:215-333)
__global__ __launch_bounds__(256) void post_kernel(Dev d, const double *__restrict__ C, long Bp, double *__restrict__ S,
                                                   int *__restrict__ RES, const int *__restrict__ ACT) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= Bp || !ACT[t]) return;
    const int np = d.np, dim = d.dim, nm = d.n + d.m;
    const int r0 = blockIdx.y * CHUNK, r1 = min(r0 + CHUNK, np);
    const long ns = d.npad;
    double *PR = S + t, *DU = PR + ns * Bp, *QH = DU + ns * Bp, *PH = QH + ns * Bp, *CI = PH + ns * Bp, *QC = CI + 2L * ns * Bp;
#define AT(P, i) (P)[(long)(i) * Bp]
    const double *cLB = C + d.LB, *cUB = C + d.UB, *cLBy = C + d.LBy, *cUBy = C + d.UBy;
    const double rho = d.rho, rho_i = d.rho_i, sigma = d.sigma, sigma_i = d.sigma_i;
    const double as = d.alpha * d.sigma, ar = d.alpha * d.rho;
    const double gz = d.symmetric ? as : sigma, gs = d.symmetric ? ar : rho;
    bool res = false;
    if (r0 < dim) {  // z rows (dim is a multiple of CHUNK or the chunk is cut at dim below)
        const int re = min(r1, dim);
        for (int j = r0; j < re; j++) {
            const double zh = AT(PH, j) + AT(CI, j), zo = AT(PR, j);
            double lam = AT(DU, j);
            if (d.symmetric) lam += as * (zh - zo);
            double z = zh + sigma_i * lam;
            if (j < dim - 3 * nm) z = fmin(fmax(z, cLB[j]), cUB[j]);
            lam = lam + gz * (zh - z);
            res |= (fabs(zo - z) > d.tol_d) | (fabs(z - zh) > d.tol_p);
            AT(PR, j) = z;
            AT(DU, j) = lam;
            const int qr = qc_row(d, j);
            AT(QH, j) = sigma * z - ((qr >= 0) ? AT(QC, qr) : 0.0) - lam;
        }
    }
    if (r1 > dim) {  // s rows, in triples
        const int j0 = (max(r0, dim) - dim) / 3, j1 = (r1 - dim) / 3;
        for (int j = j0; j < j1; j++) {
            double sh[3], so[3], mu[3], s[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int row = dim + 3 * j + r;
                sh[r] = AT(PH, row) + AT(CI, row);
                so[r] = AT(PR, row);
                mu[r] = AT(DU, row);
                if (d.symmetric) mu[r] += ar * (sh[r] - so[r]);
                s[r] = sh[r] + rho_i * mu[r];
            }
            if (d.use_soc) {
                proj_soc3(s[0], s[1], s[2], 1.0, 0.0);
            } else {
                proj_soc3(s[0], s[1], s[2], 1.0, cLBy[j]);
                proj_soc3(s[0], s[1], s[2], -1.0, cUBy[j]);
            }
#pragma unroll
            for (int r = 0; r < 3; r++) {
                const int row = dim + 3 * j + r;
                const double m2 = mu[r] + gs * (sh[r] - s[r]);
                res |= (fabs(so[r] - s[r]) > d.tol_d) | (fabs(s[r] - sh[r]) > d.tol_p);
                AT(PR, row) = s[r];
                AT(DU, row) = m2;
                AT(QH, row) = rho * s[r] - m2;
            }
        }
    }
    if (res) atomicOr(&RES[t], 1);
#undef AT
}

// one thread per instance: exit test (:335-352), k, e_flag, and the record's z_hat / s_hat of the last iteration
__global__ __launch_bounds__(64) void finish_kernel(Dev d, int k, long Bp, double *__restrict__ S, int *__restrict__ RES,
                                                    int *__restrict__ ACT, int *__restrict__ k_out, int *__restrict__ e_out,
                                                    int *__restrict__ n_active) {
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= Bp || !ACT[t]) return;
    const int r = RES[t];
    RES[t] = 0;
    if (r && k < d.k_max) return;
    const int np = d.np;
    const long ns = d.npad;
    const double *PH = S + t + 3L * ns * Bp, *CI = PH + ns * Bp;
    double *ZH = S + t + 5L * ns * Bp;
    for (int j = 0; j < np; j++) ZH[(long)j * Bp] = PH[(long)j * Bp] + CI[(long)j * Bp];
    k_out[t] = k;
    e_out[t] = r ? -1 : 1;
    ACT[t] = 0;
    atomicSub(n_active, 1);
}

// host loop.  u, k, e, fields are device pointers; fields = z, s, z_hat, s_hat, lambda, mu (NULL entries skipped)
inline int launch(Plan &p, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *scratch,
                  double *u, int *k, int *e, double *const *f, hipStream_t st) {
    if (!p.ok) return fail(SPCIES_HIP_ENOSUP, "GEMM variant unavailable: %s", p.why.c_str());
    int rc = p.blas.open();
    if (rc) return rc;
    const Dev &d = p.dev;
    const long Bp = (B + 63) / 64 * 64;
    const int np = d.np, ns = d.npad;
    double *S = scratch;
    int *RES = reinterpret_cast<int *>(S + (size_t)(6 * ns + 2 * d.n + d.m) * Bp), *ACT = RES + Bp, *NACT = ACT + Bp;
    double *QH = S + 2L * ns * Bp, *PH = S + 3L * ns * Bp;
    const int nact0 = (int)B;
    SPCIES_HIP_CHECK(hipMemcpyAsync(NACT, &nact0, sizeof(int), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(setup_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, st, d, p.d_C, x0, xr, ur, ref_stride, B, Bp, S, RES, ACT);
    SPCIES_HIP_CHECK(hipGetLastError());
    if (p.blas.set_stream(p.blas.handle, st) != 0) return fail(SPCIES_HIP_EHIP, "rocblas_set_stream failed");
    const double alpha = -1.0, beta = 0.0;
    const dim3 pgrid((unsigned)((Bp + 255) / 256), (unsigned)((np + CHUNK - 1) / CHUNK));
    const bool can_stop_early = d.tol_p > 0 || d.tol_d > 0;
    for (int it = 1; it <= d.k_max; it++) {
        // PH [Bp x np] = -QH [Bp x np] * M1'  (column-major operands: QH ld = Bp; the row-major M1 IS M1' column-major)
        if (p.blas.dgemm(p.blas.handle, ROCBLAS_OP_N, ROCBLAS_OP_N, (int)Bp, ns, ns, &alpha, QH, (int)Bp, p.d_M1, ns, &beta, PH,
                         (int)Bp) != 0)
            return fail(SPCIES_HIP_EHIP, "rocblas_dgemm failed");
        hipLaunchKernelGGL(post_kernel, pgrid, dim3(256), 0, st, d, p.d_C, Bp, S, RES, ACT);
        hipLaunchKernelGGL(finish_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, st, d, it, Bp, S, RES, ACT, k, e, NACT);
        if (can_stop_early && (it % 16 == 0)) {
            int left = 0;
            SPCIES_HIP_CHECK(hipMemcpyAsync(&left, NACT, sizeof(int), hipMemcpyDeviceToHost, st));
            SPCIES_HIP_CHECK(hipStreamSynchronize(st));
            if (left <= 0) break;
        }
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    // u = first m entries of z; record fields from PR (z, s), ZH (z_hat, s_hat), DU (lambda, mu)
    {
        dim3 tg((unsigned)(Bp / 64), (unsigned)((d.m + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, S, Bp, B, d.m, u);
    }
    const double *base[3] = {S, S + 5L * ns * Bp, S + (long)ns * Bp};
    for (int i = 0; i < 6; i++) {
        if (!f[i]) continue;
        const int rows = (i % 2 == 0) ? d.dim : d.n_s;
        const double *src = base[i / 2] + ((i % 2 == 0) ? 0 : (long)d.dim * Bp);
        dim3 tg((unsigned)(Bp / 64), (unsigned)((rows + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, src, Bp, B, rows, f[i]);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace hgemm
}  // namespace spcies
