// MFMA4 in "unit-box" coordinates: the same kernel as admm_mfma4.hpp with a third fewer FP64 vector instructions.
//
// On gfx950 an FP64 vector instruction does not co-issue with the FP64 matrix instruction: the ~750 elementwise
// instructions of an iteration cost as much pipe time as ~370 of its 786 MFMAs.  A third of them form q_hat - twice per stage,
// once in each sweep:  c = min(max(w, lb), ub);  q_hat = q + rho (w - 2 c)  - four instructions per register.  With every
// row scaled to its box,  w' = (w - lb) / D,  D = ub - lb,  the clamp is the hardware's [0, 1] output modifier on the instruction
// that forms its argument, and with the state stored shifted by the row's constant,  w^ = w' + kappa,
//     kappa = (q - rho lb) / (rho D),      c' = clamp01(w^ - kappa),      s = w^ - 2 c',      q_hat = rho D s,
// q_hat costs TWO instructions (v_add_f64 ... clamp, v_fma_f64).  The factors rho D and 1 / D fold into the block table on the
// host (the blocks that multiply q_hat get their columns scaled, the blocks that produce z their rows), the shifts - lb / D into
// the constants that seed the accumulators:
//     z'_t = (rho nhd) o s_t - (nhd / D) o mu_{t-1} - lb / D + (D^-1 Z) mu_t          w^+ = z' + (w^ - c')
// Everything else - products, their order, the ring, the exit rule - is admm_mfma4.hpp's.  The cold start (v = lambda = 0:
// q_hat = q, w^+ = z' + kappa) is not a special value of this state, so the first iteration is a second instantiation of the
// iteration body (peeled), not a multiplier inside it.
// Built when every real row has finite bounds with ub > lb (mfma4_plan_build decides; otherwise the plain kernel runs).
// Results: the iterates of admm_mfma4.hpp up to rounding (1e-10 bar of the MFMA variants; `k` equal but for exit tests decided
// within rounding).
#pragma once
#include "admm_mfma4.hpp"

namespace spcies {

// [rtc-begin]
// row constants of the unit-box kernel, 16 doubles each, behind the block stream (they replace Mfma4Layout's RC rows)
struct Mfma4uRC {
    enum {
        QR = 0,    // [Q; R] (negated weights), as RC_QR
        A1_MID,    // rho * (-Hd)            z' += A1 o s
        A2_MID,    // (-Hd) / D              z' -= A2 o mu_{t-1}   (x rows)
        A3_MID,    // lb / D
        IRD_MID,   // 1 / (rho D)            kappa = q IRD - A3
        D_MID,     // D                      (record)
        ID_MID,    // 1 / D                  (residual tolerances in the box's coordinates)
        LB_MID,
        A1_0, A3_0, IRD_0, D_0, ID_0, LB_0,     // stage 0 (u rows of the last slab; zero / neutral elsewhere)
        RD_N,      // rho D_N                wv = RD_N o s_N - mu_{N-1}
        A3_N, IRD_N, D_N, ID_N, LB_N,
        COUNT
    };
};
template <bool B>
struct Mfma4uTag { static constexpr bool value = B; };
// [rtc-end]

// The row scalings (declared in admm_mfma4.hpp, whose packer uses them) and the row constants of the table.
inline Mfma4uScaling mfma4u_scaling(const AdmmHost &a) {
    Mfma4uScaling s;
    const int n = a.n, m = a.m, nm = n + m;
    s.D_mid.assign(16, 1.0); s.D_0.assign(16, 1.0); s.D_N.assign(16, 1.0);
    s.lb_mid.assign(16, 0.0); s.lb_0.assign(16, 0.0); s.lb_N.assign(16, 0.0);
    auto take = [&](int j, std::vector<double> &D, std::vector<double> &lb) {
        const double lo = a.LB[j], hi = a.UB[j];
        if (!(std::isfinite(lo) && std::isfinite(hi)) || std::fabs(lo) > 1e8 || std::fabs(hi) > 1e8) { s.why = "a bound is infinite"; return false; }
        if (!(hi - lo > 1e-9 * (1.0 + std::fabs(lo) + std::fabs(hi)))) { s.why = "a box has no width"; return false; }
        D[j] = hi - lo;
        lb[j] = lo;
        return true;
    };
    for (int j = 0; j < nm; j++)
        if (!take(j, s.D_mid, s.lb_mid)) return s;
    for (int j = n; j < nm; j++)
        if (!take(j, s.D_0, s.lb_0)) return s;
    if (a.terminal)
        for (int j = 0; j < n; j++)
            if (!take(j, s.D_N, s.lb_N)) return s;
    if (!(a.rho > 0)) { s.why = "rho"; return s; }
    s.ok = true;
    return s;
}

inline int mfma4u_rc_count() { return Mfma4uRC::COUNT; }

inline void mfma4u_row_constants(const AdmmHost &a, const Mfma4uScaling &sc, const std::vector<double> &hd_mid, const std::vector<double> &hd_0,
                                 double *rc0) {
    const int n = a.n, m = a.m, nm = n + m;
    auto rc = [&](int i) { return rc0 + i * 16; };
    for (int j = 0; j < n; j++) rc(Mfma4uRC::QR)[j] = a.Q[j];
    for (int j = 0; j < m; j++) rc(Mfma4uRC::QR)[n + j] = a.R[j];
    for (int j = 0; j < 16; j++) {
        rc(Mfma4uRC::A1_MID)[j] = a.rho * -hd_mid[j];
        rc(Mfma4uRC::A2_MID)[j] = j < n ? -hd_mid[j] / sc.D_mid[j] : 0.0;
        rc(Mfma4uRC::A3_MID)[j] = sc.lb_mid[j] / sc.D_mid[j];
        rc(Mfma4uRC::IRD_MID)[j] = j < nm ? 1.0 / (a.rho * sc.D_mid[j]) : 0.0;
        rc(Mfma4uRC::D_MID)[j] = sc.D_mid[j];
        rc(Mfma4uRC::ID_MID)[j] = 1.0 / sc.D_mid[j];
        rc(Mfma4uRC::LB_MID)[j] = sc.lb_mid[j];
        const bool u0 = j >= n && j < nm;
        rc(Mfma4uRC::A1_0)[j] = a.rho * -hd_0[j];
        rc(Mfma4uRC::A3_0)[j] = sc.lb_0[j] / sc.D_0[j];
        rc(Mfma4uRC::IRD_0)[j] = u0 ? 1.0 / (a.rho * sc.D_0[j]) : 0.0;
        rc(Mfma4uRC::D_0)[j] = sc.D_0[j];
        rc(Mfma4uRC::ID_0)[j] = 1.0 / sc.D_0[j];
        rc(Mfma4uRC::LB_0)[j] = sc.lb_0[j];
        const bool xN = a.terminal && j < n;
        rc(Mfma4uRC::RD_N)[j] = a.rho * sc.D_N[j];
        rc(Mfma4uRC::A3_N)[j] = sc.lb_N[j] / sc.D_N[j];
        rc(Mfma4uRC::IRD_N)[j] = xN ? 1.0 / (a.rho * sc.D_N[j]) : 0.0;
        rc(Mfma4uRC::D_N)[j] = sc.D_N[j];
        rc(Mfma4uRC::ID_N)[j] = 1.0 / sc.D_N[j];
        rc(Mfma4uRC::LB_N)[j] = sc.lb_N[j];
    }
}

// ---------------------------------------------------------------------------------------------
// Device
// ---------------------------------------------------------------------------------------------
// [rtc-begin]
template <int N, int KX, int KS, bool TERMINAL, bool WANT_SOL>
__global__ __launch_bounds__(256, 1) void admm_mfma4u_kernel(MfmaArgs p, const double *__restrict__ table_g,
                                                             const double *__restrict__ x0g,
                                                             const double *__restrict__ xrg,
                                                             const double *__restrict__ urg, double *__restrict__ u_out,
                                                             int *__restrict__ k_out, int *__restrict__ e_out,
                                                             double *__restrict__ z_out, double *__restrict__ v_out,
                                                             double *__restrict__ lam_out, double *__restrict__ dump) {
    constexpr Mfma4Layout LL{N, KX, KS, TERMINAL};
    constexpr int RC0 = LL.n_tiles() * 16, TOTAL = RC0 + Mfma4uRC::COUNT * 16;
#ifdef SPCIES_RTC_STATIC_LDS
    __shared__ __attribute__((aligned(16))) double lds[TOTAL];
#else
    extern __shared__ __attribute__((aligned(16))) double lds[];
#endif
    const int n = p.n, m = p.m, nm = n + m;
    {
        const double2 *src = reinterpret_cast<const double2 *>(table_g);
        double2 *dst = reinterpret_cast<double2 *>(lds);
        for (int i = threadIdx.x; i < TOTAL / 2; i += 256) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const long n_tiles = (p.B + 15) / 16;
    const double rho = p.rho, tol = p.tol;
    const int dim = TERMINAL ? N * nm : N * nm - n;

    int ao = g * 4 + (lane & 3), go = g;
    auto BLK = [&](int t) -> double { return lds[(t / 2) * 32 + 2 * ao + (t % 2)]; };
    // The block stream is longer than the 64 KB an LDS instruction's immediate offset reaches: without two bases kept in registers the compiler
    // forms `base + 0x...` with a vector add in front of every few reads - an isolated vector instruction between MFMAs costs 12 clocks
    // (profiles/r03_microbench_issue.txt).  Both bases are laundered once per iteration (like ao) so that they stay registers.
    unsigned pb0 = (unsigned)(size_t)(lds + 2 * ao), pb1 = pb0 + 65536u;
    auto PAIR = [&](int p) -> double2 {
        const unsigned off = (unsigned)p * 256u;
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        typedef __attribute__((address_space(3))) const dbl2 *lds_p;
        const dbl2 v = off < 65536u ? *(lds_p)(size_t)(pb0 + off) : *(lds_p)(size_t)(pb1 + (off - 65536u));
        double2 r;
        r.x = v[0];
        r.y = v[1];
        return r;
    };
    auto RC = [&](int i) -> d4 {
        const double *r = lds + RC0 + i * 16;
        return d4{r[go], r[4 + go], r[8 + go], r[12 + go]};
    };
#define MFMA4(acc, a, b) acc = __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (acc), 0, 0, 0)
// (the seed of an accumulator chain is moved to the accumulator file inside the run of vector instructions that formed it)
#ifdef SPCIES_MFMA4U_SEED_ACC
#define SPCIES_SEED_TO_ACC(z) asm volatile("" : "+a"(z[0]), "+a"(z[1]), "+a"(z[2]), "+a"(z[3]))
#else
#define SPCIES_SEED_TO_ACC(z)
#endif
#ifdef SPCIES_MFMA4U_NOSPLIT
#define SPCIES_SEG_SPLIT
#else
#define SPCIES_SEG_SPLIT __builtin_amdgcn_sched_barrier(0)
#endif
    auto clamp01 = [](const d4 &x) -> d4 {
        d4 r;
#pragma unroll
        for (int i = 0; i < 4; i++) r[i] = fmin(fmax(x[i], 0.0), 1.0);  // (folds into the clamp modifier of the instruction forming x)
        return r;
    };

    for (long tile = (long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long)gridDim.x * 4) {
        const long inst = tile * 16 + c;
        const bool valid = inst < p.B;
        d4 x0v = {0, 0, 0, 0}, xrv = {0, 0, 0, 0}, xuv = {0, 0, 0, 0};
        {
            const double *xrp = p.ref_stride ? xrg + inst * n : xrg;
            const double *urp = p.ref_stride ? urg + inst * m : urg;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = 4 * r + g;
                if (valid && row < n) {
                    x0v[r] = x0g[inst * n + row];
                    xrv[r] = xrp[row];
                    xuv[r] = xrv[r];
                } else if (valid && row < nm) {
                    xuv[r] = urp[row - n];
                }
            }
        }
        const d4 qraw = RC(Mfma4uRC::QR) * xuv;  // [Q o xr; R o ur]  (negated weights)
        d4 c0 = {0, 0, 0, 0}, qT = {0, 0, 0, 0}, cN = {0, 0, 0, 0};
        {
            int t = LL.setup_base();
#pragma unroll
            for (int J = 0; J < KX; J++)
#pragma unroll
                for (int I = 0; I < KX; I++) { MFMA4(c0[I], BLK(t), x0v[J]); t++; }
#pragma unroll
            for (int J = 0; J < KX; J++)
#pragma unroll
                for (int I = 0; I < KX; I++) {
                    if constexpr (TERMINAL) MFMA4(qT[I], BLK(t), xrv[J]);
                    else MFMA4(cN[I], BLK(t), xrv[J]);
                    t++;
                }
        }
        // kappa = q / (rho D) - lb / D per kind of stage; the middle stages' row constants stay in registers
        d4 a1m = RC(Mfma4uRC::A1_MID), a2m = RC(Mfma4uRC::A2_MID), a3m = RC(Mfma4uRC::A3_MID);
        // merged backward seed (round 4; -DSPCIES_MFMA4U_PLAIN_SEED keeps the round-3 form for A/B timing): with d = w^ - c' the next state is  w^+ = z' + d = (a1 + 1) o d - a1 o c' - a3 - a2 o mu + Z mu:
        // the products accumulate ON TOP of d and deliver w^+ itself - five vector instructions per register and stage where
        // "s, seed, seed, then d, then z' + d" took six, and no value of the stage but the accumulator stays alive across its products
        d4 a1p = a1m + 1.0;
        d4 kap_m = qraw * RC(Mfma4uRC::IRD_MID) - a3m;
        // (stage 0 has its real rows - the inputs - inside the last slab, stage N in the first KX slabs: the other components are zero,
        // and the compiler is told so: registers)
        double kap_0s = (qraw * RC(Mfma4uRC::IRD_0) - RC(Mfma4uRC::A3_0))[KS - 1];
        d4 kap_N = qT * RC(Mfma4uRC::IRD_N) - RC(Mfma4uRC::A3_N);
#pragma unroll
        for (int r = KX; r < 4; r++) { kap_N[r] = 0.0; a2m[r] = 0.0; }
        auto KAP = [&](int t) -> d4 {
            if (t == 0) {
                d4 k0 = {0, 0, 0, 0};
                k0[KS - 1] = kap_0s;
                return k0;
            }
            return t == N ? kap_N : kap_m;
        };
#ifndef SPCIES_MFMA4U_RC_REGS
#define SPCIES_MFMA4U_RC_REGS 1
#endif
#if SPCIES_MFMA4U_RC_REGS
        // (round 4, default: the row constants of stages 0 and N in registers too - an LDS read in front of the instruction that needs it is
        // waited for: 5.86 -> 5.77 ms at configs[1]; -DSPCIES_MFMA4U_RC_REGS=0 reads them from LDS)
        d4 a3N = RC(Mfma4uRC::A3_N), rdN = RC(Mfma4uRC::RD_N);
        double a30s = RC(Mfma4uRC::A3_0)[KS - 1], a10s = RC(Mfma4uRC::A1_0)[KS - 1];
#pragma unroll
        for (int r = KX; r < 4; r++) { a3N[r] = 0.0; rdN[r] = 0.0; }
        auto one = [&](double x) -> d4 { d4 k0 = {0, 0, 0, 0}; k0[KS - 1] = x; return k0; };
        auto A3 = [&](int t) -> d4 { return t == 0 ? one(a30s) : (t == N ? a3N : a3m); };
#else
        auto A3 = [&](int t) -> d4 { return t == 0 ? RC(Mfma4uRC::A3_0) : (t == N ? RC(Mfma4uRC::A3_N) : a3m); };
#endif
        auto DD = [&](int t) -> d4 { return RC(t == 0 ? Mfma4uRC::D_0 : (t == N ? Mfma4uRC::D_N : Mfma4uRC::D_MID)); };
        auto LBo = [&](int t) -> d4 { return RC(t == 0 ? Mfma4uRC::LB_0 : (t == N ? Mfma4uRC::LB_N : Mfma4uRC::LB_MID)); };

        d4 w[N + 1], mu[N];  // w: the shifted state w^ = (w - lb) / D + kappa
#pragma unroll
        for (int t = 0; t <= N; t++) w[t] = d4{0, 0, 0, 0};
        bool active = valid;
        int kk = 0;

#ifndef SPCIES_MFMA4_PF
#define SPCIES_MFMA4_PF 8
#endif
        constexpr int NP = LL.stream_pairs(), PF = SPCIES_MFMA4_PF;
        static_assert(NP > PF, "ring");
#ifndef SPCIES_MFMA4U_PF
#define SPCIES_MFMA4U_PF 8
#endif
        constexpr int PFU = SPCIES_MFMA4U_PF;
        static_assert(PFU == 4 || PFU == 6 || PFU == 8, "ring depth");
        double2 r0 = PAIR(0), r1 = PAIR(1), r2 = PAIR(2), r3 = PAIR(3), r4 = PAIR(4 % PFU), r5 = PAIR(5 % PFU), r6 = PAIR(6 % PFU), r7 = PAIR(7 % PFU), cur = r0;

        // one iteration; FIRST: the cold start (q_hat = q, w^+ = z' + kappa, v_old = 0).  Returns false when the wavefront is done.
        auto iteration = [&](auto first_tag) -> bool {
            constexpr bool FIRST = decltype(first_tag)::value;
            kk += 1;
            asm volatile("" : "+v"(ao), "+v"(go), "+v"(pb0), "+v"(pb1));
            long il = inst;
            asm volatile("" : "+v"(il));
            double *zp = WANT_SOL ? z_out + il * dim + g : nullptr;
            LAUNDER4(a1m); LAUNDER4(a3m);
#ifndef SPCIES_MFMA4U_PLAIN_SEED
            LAUNDER4(a1p);
#endif
            int tix = 0;
            auto prod = [&](d4 &acc, const d4 &x, const Prod4 P) {
#pragma unroll
                for (int J = 0; J < 4; J++)
#pragma unroll
                    for (int I = 0; I < 4; I++)
                        if (P.nz(I, J)) {
                            if (tix % 2 == 0) {
                                const double2 nw = PAIR((tix / 2 + PFU) % NP);
                                if constexpr (PFU == 8) { cur = r0, r0 = r1, r1 = r2, r2 = r3, r3 = r4, r4 = r5, r5 = r6, r6 = r7, r7 = nw; }
                                else if constexpr (PFU == 6) { cur = r0, r0 = r1, r1 = r2, r2 = r3, r3 = r4, r4 = r5, r5 = nw; }
                                else { cur = r0, r0 = r1, r1 = r2, r2 = r3, r3 = nw; }
                            }
                            MFMA4(acc[I], (tix % 2 == 0) ? cur.x : cur.y, x[J]);
                            tix++;
                        }
            };
            // s_t = q_hat_t / (rho D): two instructions per register (FIRST: q / (rho D) = kappa + lb / D)
            auto qhat = [&](int t, d4 &cw) -> d4 {
                if constexpr (FIRST) {
                    cw = -A3(t);  // v_old = 0 in the box's coordinates
                    return KAP(t) + A3(t);
                } else {
                    cw = clamp01(w[t] - KAP(t));
                    return w[t] - 2.0 * cw;
                }
            };
            d4 cw;
            // ============ forward sweep ============
            d4 qh = qhat(0, cw);
            d4 qn = qhat(1, cw);
#pragma unroll
            for (int l = 0; l < N; l++) {
                d4 acc = (l == 0) ? c0 : d4{0, 0, 0, 0};
                if constexpr (!TERMINAL) {
                    if (l == N - 1) acc = cN;
                }
                // (s of stage l + 2 in ONE run of vector instructions in front of the trip's products, not spread between them)
                d4 qnn = qn;
                if (LL.stage_exists(l + 2)) qnn = qhat(l + 2, cw);
                SPCIES_SEG_SPLIT;
                prod(acc, qh, LL.F2(l));
                if (LL.hasF1(l)) prod(acc, qn, LL.F1(l));
                if (l >= 1) prod(acc, mu[l - 1], LL.F3());
                mu[l] = acc;
                qh = qn;
                qn = qnn;
                SPCIES_SEG_BARRIER;
            }
            // ============ backward sweep ============
            // (q_hat is formed again from the same state: without laundering its constants LLVM keeps the forward sweep's clamp and
            // difference of every stage alive to the end of the backward sweep - 25 spilled values per iteration)
            asm volatile("" : "+v"(go));
            LAUNDER4(a1m); LAUNDER4(a3m); LAUNDER4(kap_m);
#ifndef SPCIES_MFMA4U_PLAIN_SEED
            LAUNDER4(a1p);
#endif
            asm volatile("" : "+v"(kap_0s), "+v"(kap_N[0]), "+v"(kap_N[1]), "+v"(kap_N[2]));
            bool res = false, all_hit = false;
#ifndef SPCIES_MFMA4U_PLAIN_SEED
            constexpr bool MERGED = !FIRST;  // (the cold start keeps the plain form: it runs once)
#else
            constexpr bool MERGED = false;
#endif
            // z'_t in two parts: the elementwise seed (vector instructions only) and the products on top of it.  The seed of a stage is
            // formed BEFORE the products of the loop trip it belongs to, behind the previous stage's update: one run of vector instructions
            // and one run of MFMAs per trip (every switch from the matrix instruction to a vector instruction costs 8 clocks,
            // profiles/r03_microbench_issue.txt)
            auto stage_seed = [&](int t, d4 &cwt, d4 &x) -> d4 {
                d4 z;
                if constexpr (MERGED) {
                    // returns the seed of w^+ (not of z'): d + a1 o (d - c') - a3 [- a2 o mu]
                    const d4 cw0 = clamp01(w[t] - KAP(t));
                    const d4 dd = w[t] - cw0;
                    cwt = cw0;
                    if (t == N) {
                        const d4 s = dd - cw0;
#if SPCIES_MFMA4U_RC_REGS
                        x = rdN * s - mu[N - 1];
#else
                        x = RC(Mfma4uRC::RD_N) * s - mu[N - 1];
#endif
                        return dd - A3(N);
                    } else if (t == 0) {
#if SPCIES_MFMA4U_RC_REGS
                        const d4 a10 = one(a10s);
#else
                        const d4 a10 = RC(Mfma4uRC::A1_0);
#endif
                        x = mu[0];
                        return (a10 + 1.0) * dd - a10 * cw0 - A3(0);
                    }
                    z = a1p * dd - a3m;
                    z = z - a1m * cw0;
                    z = z - a2m * mu[t - 1];
                    x = mu[t];
                    return z;
                }
                const d4 s = qhat(t, cwt);
                if (t == N) {
#if SPCIES_MFMA4U_RC_REGS
                    x = rdN * s - mu[N - 1];
#else
                    x = RC(Mfma4uRC::RD_N) * s - mu[N - 1];
#endif
                    z = -A3(N);
                } else if (t == 0) {
#if SPCIES_MFMA4U_RC_REGS
                    z = one(a10s) * s - A3(0);
#else
                    z = RC(Mfma4uRC::A1_0) * s - A3(0);
#endif
                    x = mu[0];
                } else {
                    z = a1m * s - a3m;
                    z = z - a2m * mu[t - 1];  // (a2 is zero on the u rows; mu has x rows only)
                    x = mu[t];
                }
                return z;
            };
            auto stage_prod = [&](int t, d4 &z, const d4 &x) {
                if (t == N) prod(z, x, LL.ZN());
                else if (t == 0) prod(z, x, LL.Z0());
                else prod(z, x, LL.Zmid());
            };
            auto stage_w = [&](int t, const d4 &zin, const d4 &cwin) {
                d4 wn, z = zin, cwt = cwin;
                if constexpr (FIRST) wn = z + KAP(t);
                else if constexpr (MERGED) wn = zin;  // the products ran on top of d: this IS w^+
                else wn = z + (w[t] - cwt);
                if constexpr (MERGED) {
                    // z' and the old clamp are needed by the residual checks and the record only: rebuilt from the old state (cold path)
                    if (WANT_SOL || !all_hit) {
                        cwt = clamp01(w[t] - KAP(t));
                        z = wn - (w[t] - cwt);
                    }
                }
                if (!all_hit) {
                    const d4 vn = clamp01(wn - KAP(t));
                    const d4 told = tol * RC(t == 0 ? Mfma4uRC::ID_0 : (t == N ? Mfma4uRC::ID_N : Mfma4uRC::ID_MID));
#pragma unroll
                    for (int r = 0; r < 4; r++) res |= (fabs(cwt[r] - vn[r]) > told[r]) | (fabs(z[r] - vn[r]) > told[r]);
                    unsigned long long hb = __ballot(res);
                    hb |= hb >> 32;
                    hb |= hb >> 16;
                    all_hit = (hb & 0xFFFFull) == 0xFFFFull;
                }
                w[t] = wn;
                if constexpr (WANT_SOL) {
                    const int off = (t == 0) ? -n : (m + (t - 1) * nm);
                    const d4 zo = DD(t) * z + LBo(t);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 4 * r + g;
                        const bool in = (t == 0) ? (row >= n && row < nm) : (t == N ? row < n : row < nm);
                        double *ptr = (in && active) ? (zp + off + 4 * r) : dump;
                        *ptr = zo[r];
                    }
                }
            };
            d4 zc = {0, 0, 0, 0}, cwc = {0, 0, 0, 0};
#pragma unroll
            for (int l = N - 1; l >= 0; l--) {
                const int tp = l + 3, t = l + 2;
                if (LL.stage_exists(tp)) stage_w(tp, zc, cwc);
                d4 xz = {0, 0, 0, 0};
                if (LL.stage_exists(t)) {
                    zc = stage_seed(t, cwc, xz);
                    SPCIES_SEED_TO_ACC(zc);
                }
                SPCIES_SEG_SPLIT;
                d4 acc = {0, 0, 0, 0};
                prod(acc, mu[l], LL.B1());
                if (l < N - 1) prod(acc, mu[l + 1], LL.B2());
                if (LL.stage_exists(t)) stage_prod(t, zc, xz);
                mu[l] = acc;
                SPCIES_SEG_BARRIER;
            }
            {
                asm volatile("" : "+v"(go));
                d4 cw1, cw0, x1, x0s;
                stage_w(2, zc, cwc);
                d4 z1 = stage_seed(1, cw1, x1);
                d4 z0 = stage_seed(0, cw0, x0s);
                SPCIES_SEG_SPLIT;
                stage_prod(1, z1, x1);
                stage_prod(0, z0, x0s);
                SPCIES_SEG_SPLIT;
                stage_w(1, z1, cw1);
                stage_w(0, z0, cw0);
                SPCIES_SEG_BARRIER;
            }
            // ============ exit test per instance (code_laxMPC_ADMM_C.c:572-631) ============
            unsigned long long bal = __ballot(res);
            bal |= bal >> 32;
            bal |= bal >> 16;
            const bool res_inst = (bal >> c) & 1ull;
            const bool done_now = active && (!res_inst || kk >= p.k_max);
            if (__any(done_now)) {
                if (done_now) {
                    const d4 v0 = DD(0) * clamp01(w[0] - KAP(0)) + LBo(0);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 4 * r + g;
                        if (row >= n && row < nm) u_out[il * m + (row - n)] = v0[r];
                    }
                    if (g == 0) {
                        k_out[il] = kk;
                        e_out[il] = res_inst ? -1 : 1;
                    }
                    if constexpr (WANT_SOL) {
#pragma unroll
                        for (int t = 0; t <= N; t++) {
                            if (t == N && !TERMINAL) continue;
                            const int off = (t == 0) ? -n : (m + (t - 1) * nm);
                            const d4 wp = w[t] - KAP(t), vt = clamp01(wp), Dt = DD(t);
                            const d4 vo = Dt * vt + LBo(t), lt = rho * Dt * (wp - vt);
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int row = 4 * r + g;
                                const bool in = (t == 0) ? (row >= n && row < nm) : (t == N ? row < n : row < nm);
                                if (in) {
                                    v_out[il * dim + off + row] = vo[r];
                                    lam_out[il * dim + off + row] = lt[r];
                                }
                            }
                        }
                    }
                    active = false;
                }
            }
            return __any(active);
        };
        bool more = iteration(Mfma4uTag<true>{});
        while (more) more = iteration(Mfma4uTag<false>{});
    }
#undef MFMA4
}
// [rtc-end]

}  // namespace spcies
