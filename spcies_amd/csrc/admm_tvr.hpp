// Variant MFMA4R of the time-varying lax/equ ADMM solvers (TIME_VARYING == 1; formulations/+laxMPC/code_laxMPC_ADMM_C.c:117-279 update
// phase, :308-633 iteration): ONE WAVEFRONT PER INSTANCE, THE INSTANCE'S FACTORS IN ITS REGISTERS FOR THE WHOLE SOLVE.
//
// With `time_varying` every instance brings its own model (A, B, Q, R, LB, UB: the 9-input gateway, struct_laxMPC_ADMM_C_Matlab.c:57-103),
// so the sixteen instances of an MFMA4 wavefront no longer share their matrices.  The STREAM form (one lane per instance, bit-exact)
// re-reads the instance's factors from HBM in every iteration: 1.40 TB per 65 536 x 200 launch at the BASELINE configs[1] shape, 6.2 TB/s,
// 0.78 of the HBM peak (profiles/r04_C2tv_stream_*) - at the memory wall, 0.28 M solves/s.  Fewer bytes is the only lever, and the
// fewest is none: an instance's factors are 39 KB, a wavefront's registers hold 128 KB.
//
//   * the matrices live in registers in the A-OPERAND layout of v_mfma_f64_4x4x4 (four independent 4x4x4 blocks b; lane 16 k + 4 b + i
//     holds A_b[i][k]): register J of a matrix M holds M[4 b + i][4 J + k] - sixteen rows by four columns - so  y = M x  is ONE MFMA per
//     four columns of M.  Per stage: Bi_l = Beta_l^-1 and its transpose, -Alpha_l and its transpose (the forward sweep multiplies with
//     the transposes, the backward sweep with the matrices; the matrix pipe has no transposed operand form), 3 registers each at n = 12;
//     per instance: -AB diag(Hi) and -diag(Hi) AB' (4 + 3): 374 of the wavefront's 512 registers at n = 12, m = 2, N = 15;
//   * a vector lives in the D layout's column 0, replicated over the other three (lane 16 i + 4 b + j holds x[4 b + i] for every j): the
//     result of a product IS a vector, and the B operand of k-slab J is that register with lane 4 J of every 16-lane row broadcast
//     over the row - one v_mov_b32_dpp row_newbcast per dword, no LDS, no cross-row traffic;
//   * the iteration is MFMA4's (admm_mfma4.hpp): one state vector w = z + lambda / rho per stage (v = clamp(w), lambda = rho (w - v)),
//     explicit Beta^-1 instead of the triangular recurrences - 19 MFMAs per stage and sweep pair, no dependent scalar chain, no memory
//     access at all between the prologue and the exit; four wavefronts = four instances per CU, no barrier, no LDS.
// What it costs: the update phase also writes Bi (admm_tv_bi_kernel: one lane per instance, after the reference's factorisation), the
// prologue gathers the instance's 374 register images from the structure-of-arrays rows the update phase wrote; sums are re-associated
// (explicit inverse, w-form): 1e-10 against the oracle like every MFMA variant, `k` equal but for exit tests decided within rounding.
// STREAM stays the bit-exact variant.  Shapes: n + m <= 16, N compile-time (register arrays): the build-time list below.
#pragma once
#include "admm_tvw.hpp"

namespace spcies {
namespace tvr {

// No automatic contraction: the WANT_SOL and the plain instantiation must return the same bits; fused multiply-adds are written out.
#pragma clang fp contract(off)

struct Args {
    int k_max, ref_stride;
    double rho, tol;
    long B, Bp;
};

// Bi_l = inverse of the upper-triangular Cholesky block whose rows the update phase left in Beta (diagonal stored as its reciprocal,
// code_laxMPC_ADMM_C.c:160-279): one lane per instance, column by column (back substitution on the unit vectors).
template <int n, int m>
__global__ __launch_bounds__(64) void admm_tv_bi_kernel(int N, long B, long Bp, double *__restrict__ TVS) {
    const long t = (long)blockIdx.x * 64 + threadIdx.x;
    if (t >= B) return;
    const TvLayout tl = tv_layout(n, m, N);
    double *S = TVS + t;
    for (int l = 0; l < N; l++) {
        const int b0 = tl.Beta + l * n * n, o0 = tl.Bi + l * n * n;
#pragma unroll
        for (int j = 0; j < n; j++) {
            double x[n];
#pragma unroll
            for (int i = n - 1; i >= 0; i--) {
                if (i > j) {
                    x[i] = 0.0;
                } else if (i == j) {
                    x[i] = S[(long)(b0 + i * n + i) * Bp];  // 1 / U[j][j]: Beta holds the reciprocal
                } else {
                    double acc = 0.0;
#pragma unroll
                    for (int k = i + 1; k <= j; k++) acc += S[(long)(b0 + i * n + k) * Bp] * x[k];
                    x[i] = -S[(long)(b0 + i * n + i) * Bp] * acc;
                }
            }
#pragma unroll
            for (int i = 0; i < n; i++) S[(long)(o0 + i * n + j) * Bp] = x[i];
        }
    }
}

#define SPCIES_TVR_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (acc), 0, 0, 0)

// every lane of a 16-lane row <- lane I of that row (v_mov_b32_dpp row_newbcast:I, every lane written: no previous value to keep)
template <int I>
__device__ __forceinline__ double bcast(double v) {
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x150 + I, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x150 + I, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcast_i(double v, int i) {  // i = 0, 4, 8, 12: a constant at every call once the loops are unrolled
    return i == 0 ? bcast<0>(v) : (i == 4 ? bcast<4>(v) : (i == 8 ? bcast<8>(v) : bcast<12>(v)));
}

template <int n, int m, int N, bool TERMINAL, bool WANT_SOL>
__global__ __launch_bounds__(256, 1) void admm_tvr_kernel(Args p, const double *__restrict__ TRIg, const double *__restrict__ Tg,
                                                          const double *__restrict__ TVS, const double *__restrict__ x0g,
                                                          const double *__restrict__ xrg, const double *__restrict__ urg,
                                                          double *__restrict__ u_out, int *__restrict__ k_out, int *__restrict__ e_out,
                                                          double *__restrict__ z_out, double *__restrict__ v_out, double *__restrict__ lam_out) {
    constexpr int nm = n + m, KX = (n + 3) / 4, KS = (nm + 3) / 4;
    static_assert(nm <= 16 && N >= 2, "one 16-row register per stage vector");
    // Alpha blocks kept in the LDS: as many as a quarter of it holds (40 KB per wavefront = 80 register images)
    constexpr int NL = (N - 1) < 80 / (2 * KX) ? (N - 1) : 80 / (2 * KX);
    __shared__ double s_al[4 * (NL > 0 ? NL : 1) * 2 * KX * 64];
    const TvLayout tl = tv_layout(n, m, N);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane >> 4, lb = (lane >> 2) & 3, lj = lane & 3;
    const int vr = 4 * lb + li;  // row of a vector this lane holds (D layout, every column j)
    const int ar = 4 * lb + lj;  // row of a matrix element this lane holds as A operand; its column is 4 J + li
    const double rho = p.rho, tol = p.tol;
    const int dim = TERMINAL ? N * nm : N * nm - n;
    // the shared terminal constants (controller constants, not per instance): Hi_N = T_rho_i = (T + rho I)^-1 as A operand, T (negated) rows
    double TRIA[KX];
#pragma unroll
    for (int J = 0; J < KX; J++) {
        const int c = 4 * J + li;
        TRIA[J] = (TERMINAL && ar < n && c < n) ? TRIg[ar * n + c] : 0.0;
    }

    for (long inst = (long)blockIdx.x * 4 + wave; inst < p.B; inst += (long)gridDim.x * 4) {
        // (buffer form: SGPR descriptor + one 32-bit lane offset per gather - with flat addressing every gather carries a 64-bit VGPR pointer
        // and the scheduler, which issues all of a group's loads before their first use, runs out of registers; the host keeps one launch's
        // rows below the 4 GB a buffer resource addresses)
        const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(TVS), 0, -1, 0x00020000);
        const unsigned bp8 = (unsigned)(p.Bp * 8), io8 = (unsigned)(inst * 8);
        typedef unsigned tvr_u2 __attribute__((ext_vector_type(2)));
        auto TV = [&](int row) -> double {
            return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(trs, (unsigned)row * bp8 + io8, 0, 0));
        };
        // ---- the instance's matrices, A-operand layout (zero outside the matrix).  Bi, Bi', -AB Hi, -Hi AB' in registers; -Alpha and
        // -Alpha' of the first NL blocks in this wavefront's 40 KB of the LDS (nothing else uses it; a register image is 512 B, read back
        // conflict-free with one ds_read_b64 per lane - its address is static, so the read is issued well ahead of the MFMA), the rest
        // in registers: at n = 12, N = 15 the matrices alone would take 374 of the 512 registers and the allocator spilled 1.7 KB per lane
        double BiA[N][KX], BiTA[N][KX], AlR[(N - 1 - NL > 0 ? N - 1 - NL : 1) * 2][KX], ABHA[KS], ZTA[KX];
        double *ldsw = s_al + (size_t)wave * (NL * 2 * KX * 64) + lane;
        // (reads go through a 32-bit LDS address laundered per stage: a laundered generic pointer would turn them into flat loads, and an
        // unlaundered one lets the compiler forward the prologue's stores - every image back in a register for the whole solve)
        unsigned ldsa = (unsigned)(size_t)(__attribute__((address_space(3))) double *)ldsw;
        typedef __attribute__((address_space(3))) const double *tvr_lds_p;
        auto AL = [&](int kind, int l, int J) -> double {  // kind 0: -Alpha_l, 1: -Alpha_l'
            return l < NL ? *(tvr_lds_p)(size_t)(ldsa + (unsigned)(((l * 2 + kind) * KX + J) * 512)) : AlR[((l < NL ? NL : l) - NL) * 2 + kind][J];
        };
#pragma unroll
        for (int l = 0; l < N; l++)
#pragma unroll
            for (int J = 0; J < KX; J++) {
                const int c = 4 * J + li;
                const bool in = ar < n && c < n;
                BiA[l][J] = in ? TV(tl.Bi + (l * n + ar) * n + c) : 0.0;
                BiTA[l][J] = in ? TV(tl.Bi + (l * n + c) * n + ar) : 0.0;
                if (l < N - 1) {
                    const double a0 = in ? -TV(tl.Alpha + (l * n + ar) * n + c) : 0.0, a1 = in ? -TV(tl.Alpha + (l * n + c) * n + ar) : 0.0;
                    if (l < NL) {
                        ldsw[((l * 2 + 0) * KX + J) * 64] = a0;
                        ldsw[((l * 2 + 1) * KX + J) * 64] = a1;
                    } else {
                        AlR[((l < NL ? NL : l) - NL) * 2 + 0][J] = a0;
                        AlR[((l < NL ? NL : l) - NL) * 2 + 1][J] = a1;
                    }
                }
                if (J == KX - 1) {  // one block's gathers in flight at a time
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
        for (int J = 0; J < KS; J++) {  // -AB diag(Hi): rows < n, columns < n + m
            const int c = 4 * J + li;
            ABHA[J] = (ar < n && c < nm) ? -TV(tl.AB + ar * nm + c) * TV(tl.Hi + c) : 0.0;
        }
#pragma unroll
        for (int J = 0; J < KX; J++) {  // -diag(Hi) AB': rows < n + m, columns < n
            const int c = 4 * J + li;
            ZTA[J] = (ar < nm && c < n) ? -TV(tl.Hi + ar) * TV(tl.AB + c * nm + ar) : 0.0;
        }
        // The matrices are only ever A operands of MFMAs, which read the accumulation half of the register file directly: pinned there
        // (the allocator otherwise keeps them in the 256 architectural registers next to the state and moves them back and forth)
#pragma unroll
        for (int l = 0; l < N; l++)
#pragma unroll
            for (int J = 0; J < KX; J++) {
                asm volatile("" : "+a"(BiA[l][J]));
                asm volatile("" : "+a"(BiTA[l][J]));
            }
#pragma unroll
        for (int q = 0; q < (N - 1 - NL > 0 ? N - 1 - NL : 0) * 2; q++)
#pragma unroll
            for (int J = 0; J < KX; J++) asm volatile("" : "+a"(AlR[q][J]));
        // ---- the instance's vectors (row vr in every lane that holds it)
        const bool isx = vr < n, isu = vr >= n && vr < nm;
        const double *xrp = p.ref_stride ? xrg + inst * n : xrg, *urp = p.ref_stride ? urg + inst * m : urg;
        const double hdv = (vr < nm) ? TV(tl.Hi + (vr < nm ? vr : 0)) : 0.0;  // Hi of the middle stages; its u part is Hi_0 (both 1 / (R + rho))
        const double lbv = (vr < nm) ? TV(tl.LB + (vr < nm ? vr : 0)) : 0.0, ubv = (vr < nm) ? TV(tl.UB + (vr < nm ? vr : 0)) : 0.0;
        const double xrv = isx ? xrp[isx ? vr : 0] : 0.0;
        double qv = 0.0;  // [Q o xr; R o ur] with the negated weights the update phase stored (code_laxMPC_ADMM_C.c:282-299)
        if (isx) qv = TV(tl.Q + (isx ? vr : 0)) * xrv;
        if (isu) qv = TV(tl.R + (isu ? vr - n : 0)) * urp[isu ? vr - n : 0];
        double qTv = 0.0, c0v = 0.0;  // T xr (T negated), A x0
        if (isx) {
            for (int c = 0; c < n; c++) {
                if (TERMINAL) qTv += Tg[vr * n + c] * xrp[c];
                c0v += TV(tl.AB + vr * nm + c) * x0g[inst * n + c];
            }
        }
        const double hdx = isx ? hdv : 0.0, nhd = -hdv;
        // stage kinds: 0 (u rows), middle, N (x rows): rows a stage does not have keep w = v = 0 (bounds 0, q 0)
        const double lb0 = isu ? lbv : 0.0, ub0 = isu ? ubv : 0.0, q0 = isu ? qv : 0.0;
        const double lbN = isx ? lbv : 0.0, ubN = isx ? ubv : 0.0;

        double w[N + 1], mu[N];
#pragma unroll
        for (int t = 0; t <= N; t++) w[t] = 0.0;
        auto LBt = [&](int t) { return t == 0 ? lb0 : (t == N ? lbN : lbv); };
        auto UBt = [&](int t) { return t == 0 ? ub0 : (t == N ? ubN : ubv); };
        auto Qt = [&](int t) { return t == 0 ? q0 : (t == N ? qTv : qv); };
        auto clampv = [](double x, double lo, double hi) { return fmin(fmax(x, lo), hi); };
        // acc += M x: one MFMA per four columns; the B operand is the vector's register with lane 4 J of every row broadcast
        // (the B operands of a product are formed in ONE run of DPP moves in front of its MFMAs: a vector instruction alone between two
        // MFMAs costs 12 clocks, in a run 4 - profiles/r03_microbench_issue.txt)
        auto bops = [&](double (&b)[4], const int KJ, const double x) __attribute__((always_inline)) {
#pragma unroll
            for (int J = 0; J < 4; J++) b[J] = J < KJ ? bcast_i(x, 4 * J) : 0.0;
        };
        auto mvb = [&](double &acc, const double *A, const int KJ, const double (&b)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int J = 0; J < KJ; J++) SPCIES_TVR_MFMA(acc, A[J], b[J]);
        };
        auto mv = [&](double &acc, const double *A, const int KJ, const double x) __attribute__((always_inline)) {
            double b[4];
            bops(b, KJ, x);
            __builtin_amdgcn_sched_barrier(0);  // (the moves stay one run: the scheduler would sink each pair next to its MFMA)
            mvb(acc, A, KJ, b);
        };
        auto al_load = [&](double (&a)[4], const int kind, const int l) __attribute__((always_inline)) {  // this stage's LDS images, requested early
#pragma unroll
            for (int J = 0; J < 4; J++) a[J] = J < KX ? AL(kind, l, J) : 0.0;
        };
        int kk = 0;
        while (true) {
            kk += 1;
            asm volatile("" : "+v"(ldsa));  // (the LDS images never change: without this every read is hoisted out of the iteration loop - into registers)
            const double fz = (kk == 1) ? 0.0 : 1.0, rf = rho * fz;  // cold start: v = lambda = 0 in iteration 1
            auto qhat = [&](int t, double &cw) -> double {
                cw = clampv(w[t], LBt(t), UBt(t));
                return __builtin_fma(rf, __builtin_fma(-2.0, cw, w[t]), Qt(t));
            };
            double cw;
            // ============ forward sweep: right-hand side and forward substitution, block by block ============
            double qh = qhat(0, cw);
#pragma unroll
            for (int l = 0; l < N; l++) {
                asm volatile("" : "+v"(ldsa));  // (per stage: this stage's LDS images are read here, not at the top of the iteration)
                double acc, qn = 0.0;
                if (l + 1 < N) {
                    qn = qhat(l + 1, cw);
                    acc = hdx * qn;  // Hi o q_hat_x of the next stage
                } else if (TERMINAL) {
                    qn = qhat(N, cw);
                    acc = 0.0;
                    mv(acc, TRIA, KX, qn);  // Hi_N q_hat_N (dense)
                } else {
                    acc = -xrv;  // equMPC: x_N = xr in the last block row (code_equMPC_ADMM_C.c:337-352)
                }
                if (l == 0) acc += c0v;      // - b = A x0
                double bq[4], by[4], alt[4];
                if (l >= 1) al_load(alt, 1, l >= 1 ? l - 1 : 0);
                bops(bq, KS, qh);
                if (l >= 1) bops(by, KX, mu[l >= 1 ? l - 1 : 0]);
                __builtin_amdgcn_sched_barrier(0);
                mvb(acc, ABHA, KS, bq);      // - AB (Hi o q_hat_l)
                if (l >= 1) mvb(acc, alt, KX, by);  // - Alpha_{l-1}' y_{l-1}
                double y = 0.0;
                mv(y, BiTA[l], KX, acc);     // Bi_l' ( . )
                mu[l] = y;
                qh = qn;
                __builtin_amdgcn_sched_barrier(0);  // (stage by stage: the scheduler otherwise runs every stage's chain-independent work first and keeps it all alive)
            }
            // ============ backward sweep, z, w, residuals ============
            bool res = false;
            auto finish = [&](int t, double z, double cwt) __attribute__((always_inline)) {  // w_t <- z + lambda / rho, the residual tests (:575-620)
                const double wn = __builtin_fma(fz, w[t] - cwt, z);
                const double vn = clampv(wn, LBt(t), UBt(t));
                res |= (fabs(__builtin_fma(fz, cwt, -vn)) > tol) | (fabs(z - vn) > tol);
                w[t] = wn;
                if constexpr (WANT_SOL) {
                    const int off = (t == 0) ? -n : (m + (t - 1) * nm);
                    const bool in = (t == 0) ? isu : (t == N ? isx : vr < nm);
                    if (in && lj == 0) z_out[inst * (long)dim + off + vr] = z;  // (the last write is the iteration the instance stops at)
                }
            };
#pragma unroll
            for (int l = N - 1; l >= 0; l--) {
                asm volatile("" : "+v"(ldsa));
                double acc = mu[l];
                double bm[4], al[4];  // B operands of mu_{l+1}: for Alpha_l here and for AB' of stage l + 1 below
                if (l < N - 1) {
                    al_load(al, 0, l < N - 1 ? l : 0);
                    bops(bm, KX, mu[l < N - 1 ? l + 1 : 0]);
                    __builtin_amdgcn_sched_barrier(0);
                    mvb(acc, al, KX, bm);  // y_l - Alpha_l mu_{l+1}
                }
                double mn = 0.0;
                mv(mn, BiA[l], KX, acc);
                const int t = l + 1;
                if (t == N) {
                    if constexpr (TERMINAL) {
                        const double qq = qhat(N, cw);
                        double z = 0.0;
                        mv(z, TRIA, KX, mn - qq);  // z_N = -Hi_N (q_hat_N - mu_{N-1})
                        finish(N, z, cw);
                    }
                } else {
                    const double qq = qhat(t, cw);
                    double z = nhd * (qq - mn);  // (mu has x rows only: its u rows are zero)
                    mvb(z, ZTA, KX, bm);          // - Hi o (AB' mu_t): t = l + 1 <= N - 1, the operands formed above
                    finish(t, z, cw);
                }
                mu[l] = mn;
                __builtin_amdgcn_sched_barrier(0);
            }
            {
                const double qq = qhat(0, cw);
                double z = nhd * qq;
                mv(z, ZTA, KX, mu[0]);
                z = isu ? z : 0.0;  // stage 0 has its input rows only
                finish(0, z, cw);
            }
            // ============ exit (one instance per wavefront: :572-631) ============
            const bool cont = __ballot(res) != 0ull;
            if (!cont || kk >= p.k_max) {
                const double v0 = clampv(w[0], lb0, ub0);
                if (isu && lj == 0) u_out[inst * m + (vr - n)] = v0;
                if (lane == 0) {
                    k_out[inst] = kk;
                    e_out[inst] = cont ? -1 : 1;
                }
                if constexpr (WANT_SOL) {
#pragma unroll
                    for (int t = 0; t <= N; t++) {
                        if (t == N && !TERMINAL) continue;
                        const int off = (t == 0) ? -n : (m + (t - 1) * nm);
                        const bool in = (t == 0) ? isu : (t == N ? isx : vr < nm);
                        const double vt = clampv(w[t], LBt(t), UBt(t));
                        if (in && lj == 0) {
                            v_out[inst * (long)dim + off + vr] = vt;
                            lam_out[inst * (long)dim + off + vr] = rho * (w[t] - vt);
                        }
                    }
                }
                break;
            }
        }
    }
}
#undef SPCIES_TVR_MFMA

// shapes instantiated at build time: (n, m, N)
#define SPCIES_TVR_SHAPES(X) X(12, 2, 15) X(6, 2, 10)
inline bool shape_built(int n, int m, int N) {
#define X(nn, mm, NN) \
    if (n == nn && m == mm && N == NN) return true;
    SPCIES_TVR_SHAPES(X)
#undef X
    return false;
}
// (admm_tvr.hip) update-phase follow-up + solve of one chunk; pointers are device memory, TVS as admm_tv_update_kernel left it
int launch(int n, int m, int N, bool terminal, bool want_sol, const Args &a, const double *TRI, const double *T, double *TVS, const double *x0,
           const double *xr, const double *ur, double *u, int *k, int *e, double *z, double *v, double *lam, int num_cu, hipStream_t st);

}  // namespace tvr
}  // namespace spcies
