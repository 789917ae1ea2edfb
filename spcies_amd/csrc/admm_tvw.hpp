// Variant TILE of the time-varying lax/equ ADMM solvers (TIME_VARYING == 1): ONE WAVEFRONT PER INSTANCE.
//
// With `time_varying` every instance brings its own model, so sixteen instances no longer share their matrices and the
// matrix-pipe variants do not apply; the STREAM form (one lane per instance, admm_stream.hpp) re-reads the instance's 39 KB of
// factors from HBM in every iteration: 83 KB per instance and iteration, 0.28 M solves/s at the BASELINE configs[1] shape.  This
// kernel is the north star's sketch taken literally: the instance's constants - AB, the block-bidiagonal Cholesky factor
// Alpha / Beta of W = G H^-1 G'; 34.8 KB at n = 12, m = 2, N = 15 - are staged in LDS once per solve
// (from the rows admm_tv_update_kernel wrote), four wavefronts = four instances per CU, and the iteration of
// code_laxMPC_ADMM_C.c:308-633 runs on the lanes:
//   * lane 16 g + j holds row j of the stage rows s = 4 k + g (s = 0: u_0 | 1 .. N-1: (x_s, u_s) | N: x_N): z, v, lambda are
//     ceil((N + 1) / 4) registers each, every elementwise step is one instruction per register;
//   * the mat-vecs (right-hand side, z from the dual) run four stages side by side, one per 16-lane row: lane j accumulates its row
//     over the operand's entries, which it reads from LDS (a broadcast read) or gets by a DPP row broadcast;
//   * the triangular recurrences run on every 16-lane row redundantly in "column" form: mu_i leaves lane i by v_mov_dpp
//     row_newbcast:i, every lane j subtracts Beta(i, j) mu_i - the wavefront shuffle of the north star; no barrier anywhere in
//     the iteration (a wavefront never waits for another one).
// Every sum is taken in the reference's order with separate multiplications and additions (no contraction), one lane per row:
// the results are BIT-IDENTICAL to the oracle and to the STREAM variant.  Exit per instance = per wavefront.
// MEASURED (MI355X, configs[1] shape, one model per instance, 200 iterations): 6.4 ms per instance whatever the batch - 64 k cycles
// per iteration, the sum of the latencies of one long dependent chain (2 N n substitution steps of DPP move, two multiplications
// and a subtraction each, at ~12 cycles per dependent FP64 instruction) - and four instances per CU (the LDS holds no more):
// 0.157 M solves/s against 0.279 M for STREAM, which keeps 256 instances per CU in flight and is bound by HBM bandwidth instead.
// The variant is therefore on request only (set_variant TILE); it stays in the tree as the measured form of the north star's sketch.
#pragma once
#include "admm_stream.hpp"

namespace spcies {
namespace tvw {

#pragma clang fp contract(off)

struct Args {
    int N, k_max, ref_stride;
    double rho, rho_i, tol;
    long B, Bp;
};

template <int I>
__device__ __forceinline__ double row_bcast(double v) {  // every lane of a 16-lane row <- lane I of that row
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + I, 0xF, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + I, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double row_bcast_i(double v, int i) {  // i is a constant at every call once the loops are unrolled
    switch (i) {
        case 0: return row_bcast<0>(v);
        case 1: return row_bcast<1>(v);
        case 2: return row_bcast<2>(v);
        case 3: return row_bcast<3>(v);
        case 4: return row_bcast<4>(v);
        case 5: return row_bcast<5>(v);
        case 6: return row_bcast<6>(v);
        case 7: return row_bcast<7>(v);
        case 8: return row_bcast<8>(v);
        case 9: return row_bcast<9>(v);
        case 10: return row_bcast<10>(v);
        case 11: return row_bcast<11>(v);
        case 12: return row_bcast<12>(v);
        case 13: return row_bcast<13>(v);
        case 14: return row_bcast<14>(v);
        default: return row_bcast<15>(v);
    }
}

__host__ __device__ inline int lds_doubles_per_wave(int n, int m, int N) {
    const TvLayout tl = tv_layout(n, m, N);
    return (tl.Hi + 15) / 16 * 16 + 3 * (N + 1) * 16;  // constants | q_hat by stage row | mu by block (+ a zero row) | 1 / diag(H) by stage row
}

// NRK = ceil((N + 1) / 4) registers per vector; NW wavefronts (instances) per workgroup
template <int n, int m, bool TERMINAL, int NRK>
__global__ __launch_bounds__(256) void admm_tvw_kernel(Args a, const double *__restrict__ HiN_g, const double *__restrict__ T_g,
                                                       const double *__restrict__ TVS, const double *__restrict__ x0g,
                                                       const double *__restrict__ xrg, const double *__restrict__ urg,
                                                       double *__restrict__ u_out, int *__restrict__ k_out, int *__restrict__ e_out,
                                                       double *__restrict__ z_out, double *__restrict__ v_out, double *__restrict__ lam_out) {
    constexpr int nm = n + m;
    static_assert(nm <= 16, "one stage row per 16-lane row");
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int N = a.N;
    const TvLayout tl = tv_layout(n, m, N);
    const int per_wave = lds_doubles_per_wave(n, m, N);
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    const int g = lane >> 4, j = lane & 15;
    const int jx = j < n ? j : n - 1;  // a valid state row for the lanes past n (their results are not used)
    double *F = lds + (long)wave * per_wave;                      // the instance's AB, Alpha, Beta: rows 0 .. tl.Hi of its scratch
    double *ZB = F + (tl.Hi + 15) / 16 * 16;                   // q_hat by stage row: [N + 1][16]
    double *MUB = ZB + (N + 1) * 16;                              // mu by block: [N][16] and a zero row
    double *HE = MUB + (N + 1) * 16;                              // 1 / diag(Hhat) by stage row: [N + 1][16] (row 0: Hi_0 on the u rows; zero elsewhere)
    const double rho = a.rho, rho_i = a.rho_i, tol = a.tol;
    const int dim = TERMINAL ? N * nm : N * nm - n;

    for (long inst = (long)blockIdx.x * nw + wave; inst < a.B; inst += (long)gridDim.x * nw) {
        // ---- the instance's constants: rows of the structure-of-arrays scratch -> LDS
        for (int r = lane; r < tl.Hi; r += 64) F[r] = TVS[(long)r * a.Bp + inst];
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#define ABij(i_, j_) F[tl.AB + (i_) * nm + (j_)]
#define ALPHA(l_, i_, j_) F[tl.Alpha + ((l_) * n + (i_)) * n + (j_)]
#define BETA(l_, i_, j_) F[tl.Beta + ((l_) * n + (i_)) * n + (j_)]
#define TVROW(r_) TVS[(long)(r_) * a.Bp + inst]  // the other constants: read once, at set-up, straight from the scratch
#define HI(l_, j_) TVROW(tl.Hi + (l_) * nm + (j_))
#define HI0(j_) TVROW(tl.Hi_0 + (j_))
        for (int s = g; s <= N; s += 4)
            HE[s * 16 + j] = s == 0 ? ((j >= n && j < nm) ? HI0(j - n) : 0.0) : ((s < N && j < nm) ? HI(s - 1, j) : 0.0);
        if (g == 0) MUB[N * 16 + j] = 0.0;
        __builtin_amdgcn_wave_barrier();
        // ---- per-lane constants and the per-instance setup (code_laxMPC_ADMM_C.c:282-299)
        const double *xrp = a.ref_stride ? xrg + inst * n : xrg, *urp = a.ref_stride ? urg + inst * m : urg;
        double abr[nm], abc[n], hin[n];  // row j of AB (j < n), column j of AB, row j of Hi_N (terminal)
#pragma unroll
        for (int i = 0; i < nm; i++) abr[i] = ABij(jx, i);
#pragma unroll
        for (int i = 0; i < n; i++) {
            abc[i] = ABij(i, j < nm ? j : 0);
            hin[i] = TERMINAL ? HiN_g[jx * n + i] : 0.0;
        }
        const double lb = j < nm ? TVROW(tl.LB + j) : 0.0, ub = j < nm ? TVROW(tl.UB + j) : 0.0;
        double bj = 0.0, qT = 0.0;
#pragma unroll
        for (int i = 0; i < n; i++) bj = bj - abr[i] * x0g[inst * n + i];
        const double xrj = xrp[jx];
        if (TERMINAL) {
#pragma unroll
            for (int i = 0; i < n; i++) qT = qT + T_g[jx * n + i] * xrp[i];
        }
        const double qj = j < n ? TVROW(tl.Q + j) * xrj : (j < nm ? TVROW(tl.R + (j - n)) * urp[j - n] : 0.0);
        // validity of this lane's entry of stage row s
        auto valid = [&](int s) -> bool { return s == 0 ? (j >= n && j < nm) : (s < N ? j < nm : (s == N && TERMINAL && j < n)); };

        double z[NRK], v[NRK], lam[NRK], v1[NRK];
#pragma unroll
        for (int k = 0; k < NRK; k++) { z[k] = 0.0; v[k] = 0.0; lam[k] = 0.0; }
        int kk = 0, flag = -1;
        while (true) {
            kk += 1;
            // ---- q_hat into z (:323-349), v1 = v
#pragma unroll
            for (int k = 0; k < NRK; k++) {
                const int s = 4 * k + g;
                v1[k] = v[k];
                const double q = (s == N) ? qT : qj;
                z[k] = valid(s) ? (q + lam[k]) - rho * v[k] : 0.0;
                if (s <= N) ZB[s * 16 + j] = z[k];
            }
            __builtin_amdgcn_wave_barrier();
            // ---- right-hand side -G H^-1 q_hat - b into mu (:355-381; equMPC: code_equMPC_ADMM_C.c:337-352).  One code path for every
            // block - the four 16-lane rows work on four different blocks at once, a branch per kind of block would run every kind one
            // after the other: what a block does not have enters as a zero (x - 0, x + 0 y and 0 + x are exact, so the reference's sums
            // keep their values bit for bit)
#pragma unroll
            for (int k = 0; k < NRK; k++) {
                const int l = 4 * k + g;
                const bool in = l < N, last = l == N - 1;
                const int lc = in ? l : 0;
                const double p = ZB[(lc + 1) * 16 + jx], hx = HE[(lc + 1) * 16 + jx];
                double acc = last ? 0.0 : hx * p - (lc == 0 ? bj : 0.0);
                if (TERMINAL) {
#pragma unroll
                    for (int i = 0; i < n; i++) acc = acc + (last ? hin[i] : 0.0) * ZB[N * 16 + i];
                }
#pragma unroll
                for (int i = 0; i < nm; i++) acc = acc - abr[i] * HE[lc * 16 + i] * ZB[lc * 16 + i];
                if (!TERMINAL) acc = acc - (last ? xrj : 0.0);
                if (in) MUB[l * 16 + j] = acc;
            }
            __builtin_amdgcn_wave_barrier();
            // ---- mu <- W^-1 mu (:388-451): forward and backward substitution, every 16-lane row redundantly.  The coefficients of a
            // block are read from LDS in one go (pinned: the compiler otherwise sinks every read to its use and waits for it there, twelve
            // LDS round trips per block); mu_i leaves lane i as acc_i and is scaled by Beta(i, i) on every lane (the same operation on
            // the same operands: the same bits); the update runs on every lane - Beta's other triangle is stored as zeros, so the lanes
            // it does not concern subtract 0 x mu_i from values nobody reads again - which keeps the chain free of branches.
#define SPCIES_TVW_PIN12(a_) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a_[0]), "+v"(a_[1]), "+v"(a_[2]), "+v"(a_[3]), "+v"(a_[4]), "+v"(a_[5]), "+v"(a_[6 % n]), "+v"(a_[7 % n]), "+v"(a_[8 % n]), "+v"(a_[9 % n]), "+v"(a_[10 % n]), "+v"(a_[11 % n]))
            static_assert(n >= 6 && n <= 12, "pin list");
            {
                double mp = 0.0;
                for (int l = 0; l < N; l++) {
                    double acc = MUB[l * 16 + j];
                    double av[n], bv[n], bd[n];
                    const double *al = &ALPHA(l > 0 ? l - 1 : 0, 0, jx), *be = &BETA(l, 0, jx), *dg = &BETA(l, 0, 0);
#pragma unroll
                    for (int i = 0; i < n; i++) {
                        av[i] = al[i * n];        // Alpha(l - 1, i, j)
                        bv[i] = be[i * n];        // Beta(l, i, j): zero for i > j
                        bd[i] = dg[i * (n + 1)];  // Beta(l, i, i)
                    }
                    SPCIES_TVW_PIN12(av);
                    SPCIES_TVW_PIN12(bv);
                    SPCIES_TVW_PIN12(bd);
                    if (l > 0) {
#pragma unroll
                        for (int i = 0; i < n; i++) acc = acc - av[i] * row_bcast_i(mp, i);
                    }
                    double mur = 0.0;
#pragma unroll
                    for (int i = 0; i < n; i++) {
                        const double mui = bd[i] * row_bcast_i(acc, i);  // mu_i = Beta(i, i) acc_i, on every lane
                        if (j == i) mur = mui;
                        acc = acc - bv[i] * mui;
                    }
                    MUB[l * 16 + j] = mur;
                    mp = mur;
                }
                double mn = 0.0;
                for (int l = N - 1; l >= 0; l--) {
                    double acc = MUB[l * 16 + j];
                    double av[n], bv[n], bd[n];
                    const double *al = &ALPHA(l < N - 1 ? l : 0, jx, 0), *be = &BETA(l, jx, 0), *dg = &BETA(l, 0, 0);
#pragma unroll
                    for (int i = 0; i < n; i++) {
                        av[i] = al[i];            // Alpha(l, j, i)
                        bv[i] = be[i];            // Beta(l, j, i): zero for i < j
                        bd[i] = dg[i * (n + 1)];  // Beta(l, i, i)
                    }
                    SPCIES_TVW_PIN12(av);
                    SPCIES_TVW_PIN12(bv);
                    SPCIES_TVW_PIN12(bd);
                    if (l < N - 1) {
#pragma unroll
                        for (int i = n - 1; i >= 0; i--) acc = acc - av[i] * row_bcast_i(mn, i);
                    }
                    double mur = 0.0;
#pragma unroll
                    for (int i = n - 1; i >= 0; i--) {
                        const double mui = bd[i] * row_bcast_i(acc, i);
                        if (j == i) mur = mui;
                        acc = acc - bv[i] * mui;
                    }
                    MUB[l * 16 + j] = mur;
                    mn = mur;
                }
            }
#undef SPCIES_TVW_PIN12
            __builtin_amdgcn_wave_barrier();
            // ---- z = -Hhat^-1 (q_hat + G' mu) (:456-485), one code path for every stage row as above
#pragma unroll
            for (int k = 0; k < NRK; k++) {
                const int s = 4 * k + g;
                const int sc = s <= N ? s : 0;
                const double msub = (sc >= 1 && j < n) ? MUB[(sc - 1) * 16 + jx] : 0.0;
                double acc = z[k] - msub;  // (stage row 0 and the u rows: z - 0)
                const double aux = acc;    // terminal row: z_N - mu_{N-1}
#pragma unroll
                for (int i = 0; i < n; i++) acc = acc + abc[i] * MUB[sc * 16 + i];  // mu_s (row N of the buffer is zero)
                double zn = -HE[sc * 16 + (j < nm ? j : 0)] * acc;
                if (TERMINAL) {
                    double at = 0.0;
#pragma unroll
                    for (int i = 0; i < n; i++) at = at - hin[i] * row_bcast_i(aux, i);
                    if (sc == N) zn = at;
                }
                z[k] = valid(s) ? zn : 0.0;
            }
            // ---- v = clamp(z + lambda / rho), lambda += rho (z - v) (:490-568); residuals (:572-620)
            bool res = false;
#pragma unroll
            for (int k = 0; k < NRK; k++) {
                const int s = 4 * k + g;
                if (valid(s)) {
                    double x = z[k] + rho_i * lam[k];
                    x = (x > lb) ? x : lb;
                    x = (x > ub) ? ub : x;
                    v[k] = x;
                    lam[k] = lam[k] + rho * (z[k] - x);
                    double r1 = v1[k] - x, r2 = z[k] - x;
                    r1 = (r1 > 0.0) ? r1 : -r1;
                    r2 = (r2 > 0.0) ? r2 : -r2;
                    res |= (r1 > tol) | (r2 > tol);
                }
            }
            const bool any = __ballot(res) != 0ull;
            if (!any) { flag = 1; break; }
            if (kk >= a.k_max) { flag = -1; break; }
        }
        // ---- results (:636-686)
        if (g == 0 && j >= n && j < nm) u_out[inst * m + (j - n)] = v[0];
        if (lane == 0) {
            k_out[inst] = kk;
            e_out[inst] = flag;
        }
        if (z_out || v_out || lam_out) {
#pragma unroll
            for (int k = 0; k < NRK; k++) {
                const int s = 4 * k + g;
                if (valid(s)) {
                    const long o = inst * (long)dim + (s == 0 ? j - n : m + (s - 1) * nm + j);
                    if (z_out) z_out[o] = z[k];
                    if (v_out) v_out[o] = v[k];
                    if (lam_out) lam_out[o] = lam[k];
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
#undef ABij
#undef ALPHA
#undef BETA
#undef HI
#undef HI0
#undef TVROW
    }
}

}  // namespace tvw
}  // namespace spcies
