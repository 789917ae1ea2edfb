// Variant MFMA4G ("general"): the banded-Cholesky solvers on v_mfma_f64_4x4x4_4b_f64 for shapes whose
// per-instance state does NOT fit the register file (long horizons, n > 16) - the regime of
// BASELINE.json configs[2] (equMPC-FISTA, N = 30) and configs[3] (MPCT-EADMM, n = 20, N = 20).
//
// Same lane layout as admm_mfma4.hpp (16 instances per wavefront, lane = 16 g + c holds row 4 s + g of
// slab s for instance c; the D layout of one product is the B layout of the next), but
//   * the stage loop is ROLLED (N is a run-time value, one kernel per (ceil(n/4), ceil((n+m)/4)));
//   * the per-instance state streams through HBM in "slab vectors" of 64 doubles = 512 contiguous
//     bytes per wavefront access: state[tile][vector][lane];
//   * the per-stage 4x4 blocks (Beta^-1, Beta^-1 Alpha products) are staged global -> LDS by the whole
//     workgroup, one chunk per stage and sweep, double-buffered, one __syncthreads per stage; blocks
//     that do not depend on the stage (AB, AB') stay in LDS for the whole kernel.
// The kernel is bound by HBM (state bytes per iteration), not by the matrix pipe: see DESIGN.md.
#pragma once
#include <algorithm>
#include <cmath>

#include "admm_mfma4.hpp"

namespace spcies {
namespace g4 {

// ------------------------------------------------------------------------------------------------
// host: small dense helpers (any size) and the block packer
// ------------------------------------------------------------------------------------------------
struct DM {
    int r = 0, c = 0;
    std::vector<double> a;
    DM() {}
    DM(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
    double &operator()(int i, int j) { return a[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return a[(size_t)i * c + j]; }
};
inline DM mul(const DM &A, const DM &B) {
    DM C(A.r, B.c);
    for (int i = 0; i < A.r; i++)
        for (int k = 0; k < A.c; k++) {
            const double v = A(i, k);
            if (v == 0.0) continue;
            for (int j = 0; j < B.c; j++) C(i, j) += v * B(k, j);
        }
    return C;
}
inline DM tr(const DM &A) {
    DM T(A.c, A.r);
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < A.c; j++) T(j, i) = A(i, j);
    return T;
}
inline DM neg(DM A) {
    for (auto &x : A.a) x = -x;
    return A;
}
inline DM scale_cols(DM A, const std::vector<double> &d) {
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < A.c; j++) A(i, j) *= d[j];
    return A;
}
inline DM scale_rows(DM A, const std::vector<double> &d) {
    for (int i = 0; i < A.r; i++)
        for (int j = 0; j < A.c; j++) A(i, j) *= d[i];
    return A;
}
// inverse of the upper-triangular Beta block as the reference stores it (reciprocal diagonal)
inline DM beta_inverse(const double *beta, int n) {
    DM U(n, n), X(n, n);
    for (int i = 0; i < n; i++)
        for (int j = i; j < n; j++) U(i, j) = (i == j) ? 1.0 / beta[i * n + j] : beta[i * n + j];
    for (int j = 0; j < n; j++) {
        X(j, j) = 1.0 / U(j, j);
        for (int i = j - 1; i >= 0; i--) {
            double s = 0.0;
            for (int k = i + 1; k <= j; k++) s += U(i, k) * X(k, j);
            X(i, j) = -s / U(i, i);
        }
    }
    return X;
}

enum { DENSE = 0, LOWER = 1, UPPER = 2 };
__host__ __device__ constexpr bool blk_nz(int I, int J, int pat) { return pat == DENSE || (pat == LOWER ? J <= I : J >= I); }
__host__ __device__ constexpr int blk_count(int KI, int KJ, int pat) {
    int c = 0;
    for (int J = 0; J < KJ; J++)
        for (int I = 0; I < KI; I++) c += blk_nz(I, J, pat) ? 1 : 0;
    return c;
}

// Appends the 4x4 blocks of M in issue order (J outer, I inner) to a block area that starts at `base`
// (doubles).  Blocks are stored in element-interleaved pairs (one ds_read_b128 per lane fetches its
// element of two consecutive blocks): block t, element e = 4k + i at base + 32 (t/2) + 2 e + t%2.
struct BlockWriter {
    std::vector<double> &tab;
    size_t base;
    int cursor = 0;
    bool structure_ok = true;
    BlockWriter(std::vector<double> &t, size_t b) : tab(t), base(b) {}
    void emit(const DM &M, int KI, int KJ, int pat) {
        auto at = [&](int i, int j) { return (i < M.r && j < M.c) ? M(i, j) : 0.0; };
        for (int J = 0; J < KJ; J++)
            for (int I = 0; I < KI; I++) {
                if (!blk_nz(I, J, pat)) {
                    for (int i = 0; i < 4; i++)
                        for (int k = 0; k < 4; k++)
                            if (at(4 * I + i, 4 * J + k) != 0.0) structure_ok = false;
                    continue;
                }
                double *t = tab.data() + base + (size_t)(cursor / 2) * 32 + (cursor % 2);
                for (int k = 0; k < 4; k++)
                    for (int i = 0; i < 4; i++) t[2 * (k * 4 + i)] = at(4 * I + i, 4 * J + k);
                cursor++;
            }
    }
};

struct Plan {
    bool ok = false;
    std::string why = "not built";
    int KX = 0, KS = 0;
    bool general = false;  // EADMM: general Q, R (IS_DIAG == 0)
    double *d_table = nullptr;
    size_t table_bytes = 0;
    int num_cu = 256;
};
inline void plan_free(Plan &p) {
    if (p.d_table) hipFree(p.d_table);
    p.d_table = nullptr;
}
inline int plan_upload(Plan &p, const std::vector<double> &tab) {
    for (double x : tab)
        if (!std::isfinite(x)) { p.why = "non-finite folded constant (singular Beta block?)"; return 0; }
    p.table_bytes = tab.size() * sizeof(double);
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_table, p.table_bytes + 64));
    SPCIES_HIP_CHECK(hipMemcpy(p.d_table, tab.data(), p.table_bytes, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    int dev = 0;
    SPCIES_HIP_CHECK(hipGetDevice(&dev));
    SPCIES_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    p.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    p.ok = true;
    p.why.clear();
    return 0;
}

struct Args {
    int n, m, N, k_max;
    double tol;
    long B;
    int ref_stride;
};

// number of 16-instance tiles the state arrays are sized for (whole workgroups of 4 tiles)
inline long padded_tiles(long B) { return ((B + 15) / 16 + 3) / 4 * 4; }

// Persistent grid: how many workgroups per CU to launch (<= max_wgs, what the kernel was compiled for).
// More co-resident workgroups hide more HBM latency but each runs slower; what counts is the number of
// rounds over the groups times the per-round time.  Relative round times measured on C3 (FISTA, N = 30):
// 1 : 1.15 : 1.40 for 1 : 2 : 3 workgroups per CU.
inline int pick_wgs(long groups, int num_cu, int max_wgs) {
    static const double t[] = {1.0, 1.15, 1.40, 1.70};
    int best = 1;
    double best_cost = 1e300;
    for (int w = 1; w <= max_wgs && w <= 4; w++) {
        const long rounds = (groups + (long)num_cu * w - 1) / ((long)num_cu * w);
        const double cost = rounds * t[w - 1];
        if (cost < best_cost - 1e-12) { best_cost = cost; best = w; }
    }
    return best;
}

// ------------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------------
#define SPCIES_G4_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (acc), 0, 0, 0)

// acc[I] += M[I][J] x[J] over the non-zero blocks of the pattern; blocks are read from `blk` (LDS) at
// the running block index `tix` (a compile-time constant at every use once the body is unrolled)
template <int KI, int KJ, int PAT>
__device__ __forceinline__ void prod(double (&acc)[KI], const double (&x)[KJ], const double *blk, int ao, int &tix,
                                     double2 &cur) {
    bool fresh = true;  // a product may start on the second block of a pair: fetch the pair then too
#pragma unroll
    for (int J = 0; J < KJ; J++)
#pragma unroll
        for (int I = 0; I < KI; I++)
            if (blk_nz(I, J, PAT)) {
                if (tix % 2 == 0 || fresh) cur = *reinterpret_cast<const double2 *>(blk + (tix / 2) * 32 + 2 * ao);
                fresh = false;
                SPCIES_G4_MFMA(acc[I], (tix % 2 == 0) ? cur.x : cur.y, x[J]);
                tix++;
            }
#ifdef SPCIES_G4_PROD_BARRIER
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// cooperative global -> LDS staging of one chunk of CHD doubles by 256 threads
template <int CHD>
struct Stager {
    static constexpr int CH16 = CHD / 2, NST = (CH16 + 255) / 256;
    double2 r[NST];
    __device__ __forceinline__ void issue(const double *src) {
        const double2 *s = reinterpret_cast<const double2 *>(src);
#pragma unroll
        for (int i = 0; i < NST; i++) {
            const int idx = threadIdx.x + 256 * i;
            r[i] = s[idx < CH16 ? idx : CH16 - 1];  // unconditional: a conditionally written r[] is kept in scratch memory
        }
    }
    __device__ __forceinline__ void commit(double *dst) {
        double2 *d = reinterpret_cast<double2 *>(dst);
#pragma unroll
        for (int i = 0; i < NST; i++) {
            const int idx = threadIdx.x + 256 * i;
            if (idx < CH16) d[idx] = r[i];
        }
    }
};

// One tile's slice of a state array, addressed through a buffer resource: SGPR descriptor + SGPR vector
// offset + one VGPR lane offset.  (With flat/global addressing hipcc materialises a 64-bit VGPR pointer per
// access of the rolled stage loops, hoists them all out of the iteration loop and spills them.)
struct SlabBuf {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t r;
    __device__ __forceinline__ SlabBuf(double *tile_base, long vectors)
        : r(__builtin_amdgcn_make_buffer_rsrc(tile_base, 0, (int)(vectors * 512), 0x00020000)) {}
    // vector index `vec` is wave-uniform; voff = 8 * lane
    __device__ __forceinline__ double ld(int vec, int voff) const {
        return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, vec * 512, 0));
    }
    __device__ __forceinline__ void st(int vec, int voff, double x) const {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, x), r, voff, vec * 512, 0);
    }
};

// OR of a per-lane flag over the 4 lanes (g = 0..3) that hold the same instance c = lane % 16
__device__ __forceinline__ bool or_over_rows(bool f, int c) {
    unsigned long long bal = __ballot(f);
    bal |= bal >> 32;
    bal |= bal >> 16;
    return (bal >> c) & 1ull;
}

// state[tile][vector][lane] -> out[instance][row] for vectors that are slabs of a (stages x rows) array:
// vector v = l * K + s holds rows 4 s + g of stage l; rows >= rows_per_stage are padding
__global__ __launch_bounds__(256) void tile_state_to_aos_kernel(const double *__restrict__ S, long B, int stages, int K,
                                                                int rows_per_stage, double *__restrict__ out) {
    const long total = B * (long)stages * rows_per_stage;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const long inst = i / ((long)stages * rows_per_stage);
    const int e = (int)(i % ((long)stages * rows_per_stage));
    const int l = e / rows_per_stage, row = e % rows_per_stage;
    const long tile = inst / 16;
    const int c = (int)(inst % 16), s = row / 4, g = row % 4;
    out[i] = S[((tile * stages + l) * K + s) * 64 + 16 * g + c];
}

}  // namespace g4
}  // namespace spcies
