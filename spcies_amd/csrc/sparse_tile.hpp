// Variant TILE of the sparse-KKT solvers (ellipMPC ADMM soc, HMPC ADMM / SADMM split): LPI lanes per
// instance, T = 64 / LPI instances per wavefront, the vectors that the sparse kernels access at random
// (the LDL right-hand side, the SpMV inputs and outputs) live in LDS, everything that is accessed row by
// row streams through HBM in a tile-major layout scratch[tile][row][T] (LPI consecutive rows x T
// instances = 512 contiguous bytes per wavefront access).
//
// Why: with one lane per instance (soc_stream.hpp / hmpc_stream.hpp) the L D L' solve is a chain of
// dependent read-modify-writes THROUGH GLOBAL MEMORY (the right-hand side of 64 instances is 100-280 KB:
// no cache level holds it next to the lane), and at B = 65536 there is one wavefront per SIMD to hide
// that latency with.  Here the chain runs on LDS, the nonzeros of a column are spread over the LPI
// lanes of an instance, and 4-16 wavefronts share a CU.
//
// The sparsity pattern is the controller's, so the host turns each sparse operation into a STEP STREAM
// of 16-byte records, one record per lane group and step, consumed strictly in order with block prefetch
// (no index chasing on the device):
//   * forward  L y = b  : column-oriented scatter, as the reference (code_ellipMPC_ADMM_soc_C.c:166-177);
//                         step = up to LPI nonzeros of one column: RH[row] -= val * RH[col];
//   * backward L' x = y : ALSO a scatter, over the rows of L taken last to first - the reference's gather
//                         form (:182-188) would need a cross-lane reduction per row.  Same sums, other
//                         order: results agree to rounding (tests: 1e-10), not bit for bit - STREAM does that;
//   * SpMV (CSR, :152-160 / :193-205): lane group g owns row r0 + g of a group of LPI rows, a step is
//     one nonzero of each of them: acc += val * IN[col]; the last step of a group stores acc.
#pragma once
#include <cstdlib>

#include "hmpc_stream.hpp"
#include "soc_stream.hpp"

namespace spcies {
namespace tile {

#pragma clang fp contract(fast)

struct Rec {  // one lane group's share of a step
    int a;     // scatter: target row            | SpMV: input row (in the unified LDS row space)
    int b;     // scatter: source row (uniform)  | SpMV: 1 on the last step of a row group (uniform)
    double v;  // value (0 in padding)
};
static_assert(sizeof(Rec) == 16, "Rec");
#ifndef SPCIES_TILE_BS
#define SPCIES_TILE_BS 8
#endif
constexpr int BS = SPCIES_TILE_BS;  // steps per prefetch block; streams are padded to whole blocks
constexpr int WAVES = 4;        // wavefronts (tiles) per workgroup: they read the same streams and share them through L1
constexpr int SYNC_BLOCKS = 2;  // a workgroup barrier every SYNC_BLOCKS blocks keeps them within a few KB of each other

struct Stream {
    int off = 0;    // first record (in Rec units) inside the device stream allocation
    int steps = 0;  // multiple of BS
};

// ---- host: stream builders -------------------------------------------------------------------------
inline void pad_block(std::vector<Rec> &out, int lpi, int &steps) {
    while (steps % BS) {
        for (int g = 0; g < lpi; g++) out.push_back(Rec{0, 0, 0.0});
        steps++;
    }
}
// forward and backward scatter streams of the strictly lower-triangular L (CSC: col_ptr, row_idx, val)
inline void build_ldl_streams(int nrow, const int *col_ptr, const int *row_idx, const double *val, int lpi,
                              std::vector<Rec> &out, Stream &fwd, Stream &bwd) {
    fwd.off = (int)out.size();
    int steps = 0;
    for (int i = 0; i < nrow; i++)
        for (int j0 = col_ptr[i]; j0 < col_ptr[i + 1]; j0 += lpi) {
            for (int g = 0; g < lpi; g++) {
                const int j = j0 + g;
                out.push_back(j < col_ptr[i + 1] ? Rec{row_idx[j], i, val[j]} : Rec{i, i, 0.0});
            }
            steps++;
        }
    pad_block(out, lpi, steps);
    fwd.steps = steps;
    // rows of L (CSR copy)
    const int nnz = col_ptr[nrow];
    std::vector<int> ptr(nrow + 1, 0), idx(nnz);
    std::vector<double> v(nnz);
    for (int j = 0; j < nnz; j++) ptr[row_idx[j] + 1]++;
    for (int i = 0; i < nrow; i++) ptr[i + 1] += ptr[i];
    std::vector<int> fill(ptr.begin(), ptr.end() - 1);
    for (int c = 0; c < nrow; c++)
        for (int j = col_ptr[c]; j < col_ptr[c + 1]; j++) {
            const int at = fill[row_idx[j]]++;
            idx[at] = c;
            v[at] = val[j];
        }
    bwd.off = (int)out.size();
    steps = 0;
    for (int i = nrow - 1; i >= 0; i--)
        for (int j0 = ptr[i]; j0 < ptr[i + 1]; j0 += lpi) {
            for (int g = 0; g < lpi; g++) {
                const int j = j0 + g;
                out.push_back(j < ptr[i + 1] ? Rec{idx[j], i, v[j]} : Rec{i, i, 0.0});
            }
            steps++;
        }
    pad_block(out, lpi, steps);
    bwd.steps = steps;
}
// row-parallel SpMV stream of [M1 M2] (CSR each; M2 optional) with the inputs of M1 at LDS rows in1 + col
// and those of M2 at in2 + col
inline void build_spmv_stream(int nrows, const int *p1, const int *c1, const double *v1, int in1, const int *p2,
                              const int *c2, const double *v2, int in2, int lpi, std::vector<Rec> &out, Stream &st) {
    st.off = (int)out.size();
    int steps = 0;
    for (int r0 = 0; r0 < nrows; r0 += lpi) {
        int len = 1;
        for (int g = 0; g < lpi && r0 + g < nrows; g++) {
            const int r = r0 + g;
            len = std::max(len, (p1[r + 1] - p1[r]) + (p2 ? p2[r + 1] - p2[r] : 0));
        }
        for (int s = 0; s < len; s++) {
            for (int g = 0; g < lpi; g++) {
                const int r = r0 + g;
                Rec rec{0, s == len - 1 ? 1 : 0, 0.0};
                if (r < nrows) {
                    const int l1 = p1[r + 1] - p1[r];
                    if (s < l1) {
                        rec.a = in1 + c1[p1[r] + s];
                        rec.v = v1[p1[r] + s];
                    } else if (p2 && s - l1 < p2[r + 1] - p2[r]) {
                        rec.a = in2 + c2[p2[r] + s - l1];
                        rec.v = v2[p2[r] + s - l1];
                    }
                }
                out.push_back(rec);
            }
            steps++;
        }
    }
    pad_block(out, lpi, steps);
    st.steps = steps;
}

// lanes per instance: what maximises (wavefronts per CU by LDS, capped) x (instances per wavefront) / steps
inline int pick_lpi(long lds_rows, const int *col_ptr, int nrow) {
    int best = 0;
    double best_score = -1.0;
    if (const char *ev = getenv("SPCIES_TILE_LPI")) {  // experiments: force the lane split
        const int lpi = atoi(ev);
        if ((lpi == 4 || lpi == 8 || lpi == 16 || lpi == 32 || lpi == 64) && lds_rows * (64 / lpi) * 8 * WAVES <= 160 * 1024 - 2048) return lpi;
    }
    for (int lpi = 4; lpi <= 64; lpi *= 2) {
        const long bytes = lds_rows * (64 / lpi) * 8;
        if (bytes * WAVES > 160 * 1024 - 2048) continue;
        const int waves = (int)std::min<long>(16, WAVES * ((160 * 1024) / (bytes * WAVES + 256)));  // 16: what ~128 VGPRs allow
        long steps = 0;
        for (int i = 0; i < nrow; i++) steps += (col_ptr[i + 1] - col_ptr[i] + lpi - 1) / lpi;
        const double score = (double)waves * (64 / lpi) / (double)std::max<long>(steps, 1);
        if (score > best_score) { best_score = score; best = lpi; }
    }
    return best;
}

// ---- device primitives ---------------------------------------------------------------------------
struct Blk {  // one prefetch block of a lane's records
    int4 r[BS];
    __device__ __forceinline__ void load(const int4 *__restrict__ recs, int blk, int lpi, int g) {
#pragma unroll
        for (int s = 0; s < BS; s++) r[s] = recs[(blk * BS + s) * lpi + g];
    }
};
__device__ __forceinline__ double rec_val(const int4 &r) {
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)r.w << 32) | (unsigned)r.z);
}

// V[a] -= v * V[b] for every step, in order; V is [rows][T] in LDS
template <int LPI>
__device__ __forceinline__ void scatter_stream(double *V, const int4 *__restrict__ recs, int steps, int g, int cc) {
    constexpr int T = 64 / LPI;
    const int nblk = steps / BS;
    // (fetching two blocks ahead was measured slower: the extra registers cost a wavefront per SIMD)
    Blk cur, n1;
    if (nblk > 0) cur.load(recs, 0, LPI, g);
    for (int b = 0; b < nblk; b++) {
        if (b + 1 < nblk) n1.load(recs, b + 1, LPI, g);
        if ((b & (SYNC_BLOCKS - 1)) == 0) __syncthreads();  // keep the workgroup's wavefronts on the same stream lines (L1)
#pragma unroll
        for (int s = 0; s < BS; s++) {
            const double x = V[cur.r[s].y * T + cc];
            V[cur.r[s].x * T + cc] -= rec_val(cur.r[s]) * x;
        }
        cur = n1;
    }
}

// OUT[row] = sum over the row's steps of v * IN[a]; rows r0 + g of consecutive groups of LPI rows
template <int LPI>
__device__ __forceinline__ void spmv_stream(const double *IN, double *OUT, int nrows, const int4 *__restrict__ recs, int steps,
                                            int g, int cc) {
    constexpr int T = 64 / LPI;
    const int nblk = steps / BS;
    Blk cur, n1;
    if (nblk > 0) cur.load(recs, 0, LPI, g);
    double acc = 0.0;
    int row = g;
    for (int b = 0; b < nblk; b++) {
        if (b + 1 < nblk) n1.load(recs, b + 1, LPI, g);
        if ((b & (SYNC_BLOCKS - 1)) == 0) __syncthreads();
        double x[BS];
#pragma unroll
        for (int s = 0; s < BS; s++) x[s] = IN[cur.r[s].x * T + cc];
#pragma unroll
        for (int s = 0; s < BS; s++) {
            acc += rec_val(cur.r[s]) * x[s];
            if (__builtin_amdgcn_readfirstlane(cur.r[s].y)) {  // (the flag is the same in every lane)
                if (row < nrows) OUT[row * T + cc] = acc;
                acc = 0.0;
                row += LPI;
            }
        }
        cur = n1;
    }
}

// OR of a per-lane flag over the LPI lanes (g = 0..LPI-1) of instance c = lane % T
template <int LPI>
__device__ __forceinline__ bool or_over_group(bool f, int c) {
    constexpr int T = 64 / LPI;
    unsigned long long bal = __ballot(f);
#pragma unroll
    for (int sh = 32; sh >= T; sh >>= 1) bal |= bal >> sh;
    return (bal >> c) & 1ull;
}

// scratch[tile][row0 + r][T] -> out[instance][r], r < rows
__global__ __launch_bounds__(256) void tile_rows_to_aos_kernel(const double *__restrict__ S, long rows_per_tile, int T,
                                                               int row0, int rows, long B, double *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= B * rows) return;
    const long inst = i / rows;
    const int r = (int)(i % rows);
    out[i] = S[((inst / T) * rows_per_tile + row0 + r) * T + (inst % T)];
}

struct TileDev {
    Stream fwd, bwd, rhs, prim;  // LDL scatter streams; soc only: rhs = G q_hat, prim = H q_hat + HG mu
    int lpi = 0;
    size_t lds_bytes = 0;
};

// ---------------------------------------------------------------------------------------------------------
// ellipMPC ADMM soc (code_ellipMPC_ADMM_soc_C.c:84-296).  Every global row is owned by lane group row % LPI.
// scratch rows per tile: PR (np) | DU (np) | PH (np) | BH (nr) | QV (dim);  LDS rows: RH (nr) | QH (np) | PL (np)
// ---------------------------------------------------------------------------------------------------------
template <int LPI>
__global__ __launch_bounds__(64 * WAVES) void soc_tile_kernel(SocDev c, TileDev td, const double *__restrict__ C,
                                                      const int4 *__restrict__ recs, const double *__restrict__ x0g,
                                                      const double *__restrict__ xrg, const double *__restrict__ urg,
                                                      int ref_stride, const double *__restrict__ rg, int r_stride, long B,
                                                      double *__restrict__ S, int *__restrict__ k_out, int *__restrict__ e_out) {
    constexpr int T = 64 / LPI, U = 4;
    extern __shared__ __attribute__((aligned(16))) double lds_wg[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane / T, cc = lane % T;
    double *lds = lds_wg + (size_t)wave * td.lds_bytes / sizeof(double);  // each wavefront's own vectors
    const long tile = (long)blockIdx.x * WAVES + wave;  // (tiles past the batch run on instance 0's inputs, store nothing)
    const long t = tile * T + cc;
    const bool valid = t < B;
    const int n = c.n, m = c.m, nm = n + m, N = c.N, dim = c.dim, n_s = c.n_s, n_eq = c.n_eq;
    const int np = dim + n_s, nr = n_eq + n_s;
    double *RH = lds, *QH = RH + nr * T, *PL = QH + np * T;
    const long rows_per_tile = 3L * np + nr + dim;
    double *PR = S + tile * rows_per_tile * T + cc, *DU = PR + (long)np * T, *PH = DU + (long)np * T, *BH = PH + (long)np * T,
           *QV = BH + (long)nr * T;
#define AT(P, i) (P)[(i) * T]
    const long ti = valid ? t : 0;  // out-of-range lanes compute on instance 0's inputs and never store results
    const double *x0 = x0g + ti * n;
    const double *xr = ref_stride ? xrg + ti * n : xrg;
    const double *ur = ref_stride ? urg + ti * m : urg;
    const double r_ellip = rg[r_stride ? ti : 0];
    const double *cA = C + c.A, *cQ = C + c.Q, *cR = C + c.R, *cT = C + c.T, *cLB = C + c.LB, *cUB = C + c.UB,
                 *cPhiP = C + c.PhiP, *Dinv = C + c.Dinv;
    // ---- setup (:84-131), row-parallel
    for (int j = g; j < np; j += LPI) {
        AT(PR, j) = 0.0;
        AT(DU, j) = 0.0;
    }
    for (int j = g; j < nr; j += LPI) {
        double v = 0.0;
        if (j < n) {
            for (int i = 0; i < n; i++) v -= cA[j * n + i] * x0[i];
        } else if (j == n_eq - 1) {
            v = r_ellip;
        } else if (j > n_eq && j <= n_eq + n) {
            const int jj = j - n_eq - 1;
            for (int i = 0; i < n; i++) v -= cPhiP[jj * n + i] * xr[i];
        }
        AT(BH, j) = v;
    }
    for (int j = g; j < dim; j += LPI) {
        double v = 0.0;
        if (j < m) {
            for (int i = 0; i < m; i++) v += cR[j * m + i] * ur[i];
        } else if (j < m + (N - 1) * nm) {
            const int e = (j - m) % nm;
            if (e < n) {
                for (int i = 0; i < n; i++) v += cQ[e * n + i] * xr[i];
            } else {
                for (int i = 0; i < m; i++) v += cR[(e - n) * m + i] * ur[i];
            }
        } else if (j < m + (N - 1) * nm + n) {
            const int e = j - m - (N - 1) * nm;
            for (int i = 0; i < n; i++) v += cT[e * n + i] * xr[i];
        }
        AT(QV, j) = v;
    }
    const double rho = c.rho, rho_i = c.rho_i, sigma = c.sigma, sigma_i = c.sigma_i;

    // The setup above writes per-instance rows (q, bh) that OTHER lane groups of the instance read in the first iteration: without
    // this fence such a read can overtake the store and pick up whatever the scratch allocation held (a NaN there ends the solve
    // at k = 2 with flag 1: clamp(NaN) is a bound, NaN > tol is false) - seen once in a full test run, never in isolation
    __syncthreads();
    int k = 0;
    bool active = valid;
    while (true) {
        k += 1;
        // q_hat = [q + lambda - sigma z; mu - rho s]  (:144-149)
        // (element-wise loops: U row groups per pass, all global loads of a pass issued before they are used)
        for (int j0 = g; j0 < np; j0 += U * LPI) {
            double q[U], du[U], pr[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * LPI;
                q[u] = (j < dim) ? AT(QV, j) : 0.0;
                du[u] = (j < np) ? AT(DU, j) : 0.0;
                pr[u] = (j < np) ? AT(PR, j) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * LPI;
                if (j < np) QH[j * T + cc] = (j < dim) ? q[u] + du[u] - sigma * pr[u] : du[u] - rho * pr[u];
            }
        }
        // rhs = (-Gh Hh^-1) q_hat - bh  (:152-160)
        spmv_stream<LPI>(lds, RH, nr, recs + td.rhs.off, td.rhs.steps, g, cc);
        for (int i0 = g; i0 < nr; i0 += U * LPI) {
            double bh[U];
#pragma unroll
            for (int u = 0; u < U; u++) bh[u] = (i0 + u * LPI < nr) ? AT(BH, i0 + u * LPI) : 0.0;
#pragma unroll
            for (int u = 0; u < U; u++)
                if (i0 + u * LPI < nr) RH[(i0 + u * LPI) * T + cc] -= bh[u];
        }
        // W mu = rhs through L D L' (:166-188)
        scatter_stream<LPI>(RH, recs + td.fwd.off, td.fwd.steps, g, cc);
        for (int i = g; i < nr; i += LPI) RH[i * T + cc] *= Dinv[i];
        scatter_stream<LPI>(RH, recs + td.bwd.off, td.bwd.steps, g, cc);
        // primal_hat = (-Hh^-1) q_hat + (-Hh^-1 Gh') mu  (:193-205)
        spmv_stream<LPI>(lds, PL, np, recs + td.prim.off, td.prim.steps, g, cc);
        double s_norm = 0.0;
        bool res = false;
        // z: box (:209-217), lambda (:246-248), residuals (:256-267)
        for (int i0 = g; i0 < dim; i0 += U * LPI) {
            double lam[U], zo[U], lb[U], ub[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + u * LPI;
                lam[u] = (i < dim) ? AT(DU, i) : 0.0;
                zo[u] = (i < dim) ? AT(PR, i) : 0.0;
                lb[u] = (i < dim - n - 1) ? cLB[i] : 0.0;
                ub[u] = (i < dim - n - 1) ? cUB[i] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int i = i0 + u * LPI;
                if (i < dim) {
                    const double zh = PL[i * T + cc];
                    double z = zh + sigma_i * lam[u];
                    if (i < dim - n - 1) z = fmin(fmax(z, lb[u]), ub[u]);
                    if (active) {
                        AT(PH, i) = zh;
                        AT(PR, i) = z;
                        AT(DU, i) = lam[u] + sigma * (zh - z);
                    }
                    res |= (fabs(zo[u] - z) > c.tol_d) | (fabs(z - zh) > c.tol_p);
                }
            }
        }
        // un-projected s = s_hat + mu / rho into QH's tail (free now), then the SOC projection (:220-242), mu (:251-253)
        for (int i = dim + (g + LPI - dim % LPI) % LPI; i < np; i += LPI) QH[i * T + cc] = PL[i * T + cc] + rho_i * AT(DU, i);
        const double v0 = QH[dim * T + cc];
        for (int j = 1; j < n_s; j++) {
            const double v = QH[(dim + j) * T + cc];
            s_norm += v * v;
        }
        s_norm = sqrt(s_norm);
        for (int i = dim + (g + LPI - dim % LPI) % LPI; i < np; i += LPI) {
            double v = QH[i * T + cc];
            if (s_norm <= v0) {
            } else if (s_norm <= -v0) {
                v = 0.0;
            } else {
                const double step = (v0 + s_norm) / (2 * s_norm);
                v = (i == dim) ? step * s_norm : step * v;
            }
            const double so = AT(PR, i), sh = PL[i * T + cc], mu = AT(DU, i);
            if (active) {
                AT(PH, i) = sh;
                AT(PR, i) = v;
                AT(DU, i) = mu + rho * (sh - v);
            }
            res |= (fabs(so - v) > c.tol_d) | (fabs(v - sh) > c.tol_p);
        }
        const bool res_inst = or_over_group<LPI>(res, cc);
        const bool done_now = active && (!res_inst || k >= c.k_max);
        if (done_now) {
            if (g == 0) {
                k_out[t] = k;
                e_out[t] = res_inst ? -1 : 1;
            }
            active = false;
        }
        if (!__syncthreads_or(active ? 1 : 0)) break;
    }
#undef AT
}

// ---------------------------------------------------------------------------------------------------------
// HMPC ADMM / SADMM split (code_HMPC_ADMM_split_C.c:102-333).  Rows < dim and the rows of bh are owned by
// lane group row % LPI, the s rows by the lane group of their triple (dim + 3 j + r: j % LPI).
// scratch rows per tile: PR (np) | DU (np) | PH (np) | BH (nc) | QV (dim);  LDS rows: RH (nrow_M)
// ---------------------------------------------------------------------------------------------------------
template <int LPI>
__global__ __launch_bounds__(64 * WAVES) void hmpc_tile_kernel(HmpcDev c, TileDev td, const double *__restrict__ C,
                                                       const int *__restrict__ I, const int4 *__restrict__ recs,
                                                       const double *__restrict__ x0g, const double *__restrict__ xrg,
                                                       const double *__restrict__ urg, int ref_stride, long B,
                                                       double *__restrict__ S, int *__restrict__ k_out, int *__restrict__ e_out) {
    constexpr int T = 64 / LPI, U = 4;
    extern __shared__ __attribute__((aligned(16))) double lds_wg[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g = lane / T, cc = lane % T;
    double *lds = lds_wg + (size_t)wave * td.lds_bytes / sizeof(double);  // each wavefront's own vectors
    const long tile = (long)blockIdx.x * WAVES + wave;  // (tiles past the batch run on instance 0's inputs, store nothing)
    const long t = tile * T + cc;
    const bool valid = t < B;
    const int n = c.n, m = c.m, nm = n + m, N = c.N, dim = c.dim, n_s = c.n_s, n_eq = c.n_eq, nrow = c.nrow_M;
    const int np = dim + n_s, nc = n_eq + n_s;
    double *RH = lds;
    const long rows_per_tile = 3L * np + nc + dim;
    double *PR = S + tile * rows_per_tile * T + cc, *DU = PR + (long)np * T, *PH = DU + (long)np * T, *BH = PH + (long)np * T,
           *QV = BH + (long)nc * T;
#define AT(P, i) (P)[(i) * T]
    const long ti = valid ? t : 0;
    const double *x0 = x0g + ti * n;
    const double *xr = ref_stride ? xrg + ti * n : xrg;
    const double *ur = ref_stride ? urg + ti * m : urg;
    const double *cA = C + c.A, *cQQ = C + c.QQ, *cTe = C + c.Te, *cSe = C + c.Se, *cLB = C + c.LB, *cUB = C + c.UB,
                 *cLBy = C + c.LBy, *cUBy = C + c.UBy, *Dinv = C + c.Dinv, *cbh = C + c.bh;
    const int *ix0 = I + c.idx_x0;
    const int triples = c.triples(), cone0 = c.cone0();  // coupled constraints: N n_y box slacks of the outputs sit before the cones
    // ---- setup (:102-129), by row ownership
    for (int j = g; j < dim; j += LPI) {
        AT(PR, j) = 0.0;
        AT(DU, j) = 0.0;
        double v = 0.0;
        const int e = j - (N - 1) * nm - m;  // position inside the terminal / artificial-reference block of q
        if (e >= 0 && e < n) {
            for (int i = 0; i < n; i++) v -= cTe[e * n + i] * xr[i] + cQQ[e * n + i] * x0[i];
        } else if (e >= 2 * n && e < 3 * n) {
            for (int i = 0; i < n; i++) v -= cQQ[(e - 2 * n) * n + i] * x0[i];
        } else if (e >= 3 * n && e < 3 * n + m) {
            for (int i = 0; i < m; i++) v -= cSe[(e - 3 * n) * m + i] * ur[i];
        }
        AT(QV, j) = v;
    }
    for (int j = dim + g; j < cone0; j += LPI) {  // rows of the box slacks: owner = row % LPI, as in the projection below
        AT(PR, j) = 0.0;
        AT(DU, j) = 0.0;
    }
    for (int j = g; j < triples; j += LPI)
        for (int r = 0; r < 3; r++) {
            AT(PR, cone0 + 3 * j + r) = 0.0;
            AT(DU, cone0 + 3 * j + r) = 0.0;
        }
    for (int j = g; j < nc; j += LPI) {
        double v = cbh[j];
        for (int jj = 0; jj < n; jj++)
            if (ix0[jj] == j) {
                v = 0.0;
                for (int i = 0; i < n; i++) v -= cA[jj * n + i] * x0[i];
            }
        AT(BH, j) = v;
    }
    const double rho = c.rho, rho_i = c.rho_i, sigma = c.sigma, sigma_i = c.sigma_i;
    const double as = c.alpha * c.sigma, ar = c.alpha * c.rho;
    const double gz = c.symmetric ? as : sigma, gs = c.symmetric ? ar : rho;

    // The setup above writes per-instance rows (q, bh) that OTHER lane groups of the instance read in the first iteration: without
    // this fence such a read can overtake the store and pick up whatever the scratch allocation held (a NaN there ends the solve
    // at k = 2 with flag 1: clamp(NaN) is a bound, NaN > tol is false) - seen once in a full test run, never in isolation
    __syncthreads();
    int k = 0;
    bool active = valid;
    while (true) {
        k += 1;
        // rhs = [sigma z - q - lambda; rho s - mu; bh]  (:156-165)
        for (int j0 = g; j0 < dim; j0 += U * LPI) {
            double q[U], du[U], pr[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * LPI;
                q[u] = (j < dim) ? AT(QV, j) : 0.0;
                du[u] = (j < dim) ? AT(DU, j) : 0.0;
                pr[u] = (j < dim) ? AT(PR, j) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (j0 + u * LPI < dim) RH[(j0 + u * LPI) * T + cc] = sigma * pr[u] - q[u] - du[u];
        }
        for (int j = dim + g; j < cone0; j += LPI) RH[j * T + cc] = rho * AT(PR, j) - AT(DU, j);
        for (int j = g; j < triples; j += LPI)
            for (int r = 0; r < 3; r++) RH[(cone0 + 3 * j + r) * T + cc] = rho * AT(PR, cone0 + 3 * j + r) - AT(DU, cone0 + 3 * j + r);
        for (int j = g; j < nc; j += LPI) RH[(np + j) * T + cc] = AT(BH, j);
        // KKT solve through L D L' (:193-209)
        scatter_stream<LPI>(RH, recs + td.fwd.off, td.fwd.steps, g, cc);
        for (int i = g; i < nrow; i += LPI) RH[i * T + cc] *= Dinv[i];
        scatter_stream<LPI>(RH, recs + td.bwd.off, td.bwd.steps, g, cc);
        bool res = false;
        // z (:215-238, 288-312, 318-333)
        for (int j0 = g; j0 < dim; j0 += U * LPI) {
            double lm[U], zo[U], lb[U], ub[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * LPI;
                lm[u] = (j < dim) ? AT(DU, j) : 0.0;
                zo[u] = (j < dim) ? AT(PR, j) : 0.0;
                lb[u] = (!c.coupled && j < dim - 3 * nm) ? cLB[j] : 0.0;
                ub[u] = (!c.coupled && j < dim - 3 * nm) ? cUB[j] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int j = j0 + u * LPI;
                if (j < dim) {
                    const double zh = RH[j * T + cc];
                    double lam = lm[u];
                    if (c.symmetric) lam += as * (zh - zo[u]);
                    double z = zh + sigma_i * lam;
                    if (!c.coupled && j < dim - 3 * nm) z = fmin(fmax(z, lb[u]), ub[u]);
                    if (active) {
                        AT(PH, j) = zh;
                        AT(PR, j) = z;
                        AT(DU, j) = lam + gz * (zh - z);
                    }
                    res |= (fabs(zo[u] - z) > c.tol_d) | (fabs(z - zh) > c.tol_p);
                }
            }
        }
        // coupled constraints: box on the output slacks (:262-269)
        for (int j = dim + g; j < cone0; j += LPI) {
            const double sh = RH[j * T + cc], so = AT(PR, j);
            double mu = AT(DU, j);
            if (c.symmetric) mu += ar * (sh - so);
            double sv = sh + rho_i * mu;
            sv = fmin(fmax(sv, cLBy[(j - dim) % c.n_y]), cUBy[(j - dim) % c.n_y]);
            if (active) {
                AT(PH, j) = sh;
                AT(PR, j) = sv;
                AT(DU, j) = mu + gs * (sh - sv);
            }
            res |= (fabs(so - sv) > c.tol_d) | (fabs(sv - sh) > c.tol_p);
        }
        // s in triples (:241-259)
        for (int j = g; j < triples; j += LPI) {
            double sh[3], so[3], mu[3], s[3];
#pragma unroll
            for (int r = 0; r < 3; r++) {
                sh[r] = RH[(cone0 + 3 * j + r) * T + cc];
                so[r] = AT(PR, cone0 + 3 * j + r);
                mu[r] = AT(DU, cone0 + 3 * j + r);
                if (c.symmetric) mu[r] += ar * (sh[r] - so[r]);
                s[r] = sh[r] + rho_i * mu[r];
            }
            if (c.use_soc) {
                proj_soc3(s[0], s[1], s[2], 1.0, 0.0);
            } else {
                proj_soc3(s[0], s[1], s[2], 1.0, cLBy[j]);
                proj_soc3(s[0], s[1], s[2], -1.0, cUBy[j]);
            }
#pragma unroll
            for (int r = 0; r < 3; r++) {
                if (active) {
                    AT(PH, cone0 + 3 * j + r) = sh[r];
                    AT(PR, cone0 + 3 * j + r) = s[r];
                    AT(DU, cone0 + 3 * j + r) = mu[r] + gs * (sh[r] - s[r]);
                }
                res |= (fabs(so[r] - s[r]) > c.tol_d) | (fabs(s[r] - sh[r]) > c.tol_p);
            }
        }
        const bool res_inst = or_over_group<LPI>(res, cc);
        const bool done_now = active && (!res_inst || k >= c.k_max);
        if (done_now) {
            if (g == 0) {
                k_out[t] = k;
                e_out[t] = res_inst ? -1 : 1;
            }
            active = false;
        }
        if (!__syncthreads_or(active ? 1 : 0)) break;
    }
#undef AT
}

}  // namespace tile
}  // namespace spcies
