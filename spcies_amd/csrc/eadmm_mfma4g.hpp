// Variant MFMA4G of the MPCT EADMM solver (diagonal Q, R), see mfma4g.hpp for the design.
// Algorithm: formulations/+MPCT/code_MPCT_EADMM_C.c:85-457.  In block form, z1_l, z3_l (n+m rows, l = 0..N),
// z2 (n+m rows), lambda_b (b = 0..N+2), Bi_l = Beta_l^-1, K_l = per-stage diagonals (rho_l, H1i_l, H3i_l):
//   P1 (:97-117)   z1_l = clamp( H1i_l o ( rho_l (z3_l + z2) + lambda_{l+1} [+ rho_0 o x0 - lambda_0 | l = 0] ) ),
//                  z1_N = clamp( H1i_N o ( rho_N z3_N + (rho_N + rho_s) z2 + lambda_{N+1} + lambda_{N+2} ) )
//   P2 (:123-151)  z2 = W2 ( [T xr; S ur] + sum_l rho_l (z3_l - z1_l) + lambda_{l+1}  (- rho_s z1_N + lambda_{N+2}) )
//   P3 (:157-320)  q3_l = rho_l (z2 - z1_l) + lambda_{l+1};  y_l = (H3i q3)_{l+1}[x] - AB (H3i_l o q3_l)
//                  mu_l = Bi_l' y_l - Bi_l' Alpha_{l-1}' mu_{l-1};   mu_l = Bi_l mu_l - Bi_l Alpha_l mu_{l+1}
//                  z3_l = -H3i_l o ( q3_l - [mu_{l-1}; 0] + AB' mu_l )
//   duals (:371-405) lambda_{l+1} += rho_l (z2 + z3_l - z1_l), lambda_0 += rho_0 (z1_0[x] - x0), lambda_{N+2} += rho_s (z2 - z1_N)
// Two sweeps per iteration (B: z1, q3, forward substitution; C: backward substitution, z1, z3, duals and - fused - P1 + q2 of
// the next iteration); a stand-alone P1 sweep (A) runs once before iteration 1.
// State in HBM per 16 instances: z3 ((N+1) KS slab vectors), lambda ((N+3) KS), mu (N KX), KS = ceil((n+m)/4),
// KX = ceil(n/4) (z1 only when the record asks for it); traffic per iteration and stage: (6 KS + 2 KX) x 512 B.
#pragma once
#include "mfma4g.hpp"

namespace spcies {
namespace g4 {

#pragma clang fp contract(fast)  // (the STREAM headers included before this one switch contraction off)

// DIAG: diagonal Q, R (IS_DIAG == 1: H3^-1 is the vector H3i).  !DIAG: general Q, R (code_MPCT_EADMM_C.c:184-217, 321-366) - H3^-1 is
// block diagonal, blkdiag(Q_mi | Q_bi, R_bi | R_mi) per stage, and three more block products per stage replace the elementwise
// scalings: the sweep-B chunk of block l carries Qinv_{l+1} (KX x KX) and -AB H3inv_l (KX x KS, the reference's AB_mi / AB_bi),
// the sweep-C chunk -H3inv_{l+1} (KS x KS); -H3inv_0 sits in the stage-invariant region.
template <int KX, int KS, bool DIAG = true>
struct EadmmGLayout {
    static constexpr int RC = 4 * KS;
    // stage-invariant blocks: -AB (KX x KS), AB' (KS x KX), W2 (KS x KS), blkdiag(T, S) (KS x KS) [, -H3inv_0 (KS x KS)]
    static constexpr int T_NAB = 0, T_ABT = KX * KS, T_W2 = 2 * KX * KS, T_TS = 2 * KX * KS + KS * KS, T_H0 = 2 * KX * KS + 2 * KS * KS;
    static constexpr int INV_TILES = (2 * KX * KS + 2 * KS * KS + (DIAG ? 0 : KS * KS) + 1) / 2 * 2;
    enum { C_RHO0, C_RHOS, C_COUNT };
    static constexpr int INV_D = INV_TILES * 16 + C_COUNT * RC;
    // per-stage diagonals and bounds
    enum { K_RHO, K_H1I, K_H3I, K_LB, K_UB, K_COUNT };
    static constexpr int KD = K_COUNT * RC;
    static constexpr int NTS = blk_count(KX, KX, LOWER) + KX * KX;  // triangular + dense block of the substitution
    static constexpr int NTX_B = KX * KX + KX * KS, NTX_C = KS * KS;  // general Q, R: extra blocks of a sweep-B / sweep-C chunk
    static constexpr int NT = NTS + (DIAG ? 0 : (NTX_B > NTX_C ? NTX_B : NTX_C)), NT_PAD = (NT + 1) / 2 * 2;
    static constexpr int CHD = NT_PAD * 16 + 2 * KD;  // blocks, K_a, K_b
    static constexpr int LDS_D = INV_D + 2 * CHD;
    static int n_seq(int N) { return 3 * N + 1; }
    static size_t table_doubles(int N) { return (size_t)INV_D + (size_t)n_seq(N) * CHD; }
};

struct EadmmGHost {
    const std::vector<double> *rho, *rho0, *rhos, *LB0, *UB0, *LBs, *UBs, *S, *H1i, *W2, *H3i;
    bool diag = true;
    const std::vector<double> *Q_bi = nullptr, *Q_mi = nullptr, *R_bi = nullptr, *R_mi = nullptr;  // general Q, R
};

template <int KX, int KS, bool DIAG>
inline int eadmm_plan_build_shape(Plan &p, const AdmmHost &a, const EadmmGHost &h) {
    using LY = EadmmGLayout<KX, KS, DIAG>;
    const int n = a.n, m = a.m, N = a.N, nm = n + m;
    std::vector<double> tab(LY::table_doubles(N), 0.0);
    DM AB(n, nm), W2(nm, nm), TS(nm, nm);
    for (int i = 0; i < n; i++)
        for (int j = 0; j < nm; j++) AB(i, j) = a.AB[(size_t)i * nm + j];
    for (int i = 0; i < nm; i++)
        for (int j = 0; j < nm; j++) W2(i, j) = (*h.W2)[(size_t)i * nm + j];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) TS(i, j) = a.T[(size_t)i * n + j];
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) TS(n + i, n + j) = (*h.S)[(size_t)i * m + j];
    // H3^-1 of stage l (general Q, R): blkdiag(Q_mi at l = 0 and N, else Q_bi; R_mi at l = N, else R_bi)
    auto H3inv = [&](int l) {
        DM M(nm, nm);
        if (!DIAG) {
            const std::vector<double> &Qi = (l == 0 || l == N) ? *h.Q_mi : *h.Q_bi, &Ri = (l == N) ? *h.R_mi : *h.R_bi;
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) M(i, j) = Qi[(size_t)i * n + j];
            for (int i = 0; i < m; i++)
                for (int j = 0; j < m; j++) M(n + i, n + j) = Ri[(size_t)i * m + j];
        }
        return M;
    };
    auto xblock = [&](const DM &M) {  // the state rows / columns
        DM X(n, n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) X(i, j) = M(i, j);
        return X;
    };
    bool ok = true;
    {
        BlockWriter w(tab, 0);
        w.emit(neg(AB), KX, KS, DENSE);
        w.emit(tr(AB), KS, KX, DENSE);
        w.emit(W2, KS, KS, DENSE);
        w.emit(TS, KS, KS, DENSE);
        if (!DIAG) w.emit(neg(H3inv(0)), KS, KS, DENSE);
        ok = ok && w.structure_ok && w.cursor == 2 * KX * KS + 2 * KS * KS + (DIAG ? 0 : KS * KS);
        double *rc = tab.data() + LY::INV_TILES * 16;
        for (int j = 0; j < nm; j++) {
            rc[LY::C_RHO0 * LY::RC + j] = (*h.rho0)[j];
            rc[LY::C_RHOS * LY::RC + j] = (*h.rhos)[j];
        }
    }
    auto put_K = [&](double *dst, int l) {
        const double *lb = (l == 0) ? h.LB0->data() : (l == N ? h.LBs->data() : a.LB.data());
        const double *ub = (l == 0) ? h.UB0->data() : (l == N ? h.UBs->data() : a.UB.data());
        for (int j = 0; j < nm; j++) {
            dst[LY::K_RHO * LY::RC + j] = (*h.rho)[(size_t)l * nm + j];
            dst[LY::K_H1I * LY::RC + j] = (*h.H1i)[(size_t)l * nm + j];
            dst[LY::K_H3I * LY::RC + j] = DIAG ? (*h.H3i)[(size_t)l * nm + j] : 0.0;
            dst[LY::K_LB * LY::RC + j] = lb[j];
            dst[LY::K_UB * LY::RC + j] = ub[j];
        }
    };
    std::vector<DM> Bi(N), Al(N - 1);
    for (int l = 0; l < N; l++) Bi[l] = beta_inverse(a.Beta.data() + (size_t)l * n * n, n);
    for (int l = 0; l < N - 1; l++) {
        Al[l] = DM(n, n);
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Al[l](i, j) = a.Alpha[((size_t)l * n + i) * n + j];
    }
    const DM Zero(n, n);
    for (int s = 0; s < LY::n_seq(N); s++) {
        const size_t base = (size_t)LY::INV_D + (size_t)s * LY::CHD;
        double *Ka = tab.data() + base + LY::NT_PAD * 16, *Kb = Ka + LY::KD;
        BlockWriter w(tab, base);
        if (s <= N) {  // sweep A: stage N first, then 0..N-1
            put_K(Ka, s == 0 ? N : s - 1);
            continue;
        }
        if (s <= 2 * N) {  // sweep B: block l
            const int l = s - N - 1;
            const DM BiT = tr(Bi[l]);
            w.emit(BiT, KX, KX, LOWER);
            w.emit(l >= 1 ? neg(mul(BiT, tr(Al[l - 1]))) : Zero, KX, KX, DENSE);
            if (!DIAG) {
                w.emit(xblock(H3inv(l + 1)), KX, KX, DENSE);
                w.emit(neg(mul(AB, H3inv(l))), KX, KS, DENSE);
            }
            put_K(Ka, l);
            put_K(Kb, l + 1);
            ok = ok && w.structure_ok && w.cursor == LY::NTS + (DIAG ? 0 : LY::NTX_B);
        } else {  // sweep C: block l
            const int l = 3 * N - s;
            w.emit(Bi[l], KX, KX, UPPER);
            w.emit(l < N - 1 ? neg(mul(Bi[l], Al[l])) : Zero, KX, KX, DENSE);
            if (!DIAG) w.emit(neg(H3inv(l + 1)), KS, KS, DENSE);
            put_K(Ka, l);
            put_K(Kb, l + 1);
            ok = ok && w.structure_ok && w.cursor == LY::NTS + (DIAG ? 0 : LY::NTX_C);
        }
    }
    if (!ok) { p.why = "MFMA4G packer: block structure mismatch"; return 0; }
    p.KX = KX;
    p.KS = KS;
    p.general = !DIAG;
    return plan_upload(p, tab);
}

// -------------------------------------------------------------------------------------------------
// Two sweeps per iteration.  P1 is elementwise, so z1 is never stored: sweep B and sweep C rebuild z1^k_l from
// (z3^{k-1}_l, lambda^{k-1}_{l+1}, z2^{k-1}) where they need it, and sweep C - which has z3^k_l, lambda^k_{l+1} and z2^k in
// registers - evaluates P1 of iteration k + 1 on the spot to accumulate q2^{k+1}.  The stand-alone P1 sweep ("A") runs
// once, before iteration 1.  Traffic per iteration and stage: (6 KS + 2 KX) x 512 B (was 10 KS + 2 KX).
// WANT_Z1: the record's z1 is requested, sweep C then also writes z1^k.
template <int KX, int KS, int WG_PER_CU, bool WANT_Z1, bool DIAG>
__global__ __launch_bounds__(256, WG_PER_CU) void eadmm_g_kernel(Args p, const double *__restrict__ tab,
                                                                 const double *__restrict__ x0g,
                                                                 const double *__restrict__ xrg,
                                                                 const double *__restrict__ urg, double *__restrict__ Z1g,
                                                                 double *__restrict__ Z3g, double *__restrict__ LAMg,
                                                                 double *__restrict__ MUg, double *__restrict__ C2g,
                                                                 double *__restrict__ u_out,
                                                                 int *__restrict__ k_out, int *__restrict__ e_out,
                                                                 double *__restrict__ z2_out) {
    using LY = EadmmGLayout<KX, KS, DIAG>;
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = p.n, m = p.m, nm = n + m, N = p.N;
    for (int i = threadIdx.x; i < LY::INV_D / 2; i += 256)
        reinterpret_cast<double2 *>(lds)[i] = reinterpret_cast<const double2 *>(tab)[i];
    double *ring = lds + LY::INV_D;
    const double *seq = tab + LY::INV_D;
    const double *inv_rc = lds + LY::INV_TILES * 16;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // uniform: state addresses are SGPR base + lane offset
    const int g = lane >> 4, c = lane & 15;
    const int ao = g * 4 + (lane & 3);
    const long n_tiles = (p.B + 15) / 16, n_groups = (n_tiles + 3) / 4;
    const double tol = p.tol;
    const int n_seq = 3 * N + 1;
    Stager<LY::CHD> stg;
#define SPCIES_K(K, which, s) (K)[(which) * LY::RC + 4 * (s) + g]

    for (long group = blockIdx.x; group < n_groups; group += gridDim.x) {
        const long tile = group * 4 + wave;
        const long inst = tile * 16 + c;
        const bool valid = inst < p.B;
        const SlabBuf Z1(Z1g + (WANT_Z1 ? tile * (long)(N + 1) * KS * 64 : 0), WANT_Z1 ? (long)(N + 1) * KS : 0);
        const SlabBuf Z3(Z3g + tile * (long)(N + 1) * KS * 64, (long)(N + 1) * KS);
        const SlabBuf LAM(LAMg + tile * (long)(N + 3) * KS * 64, (long)(N + 3) * KS), MU(MUg + tile * (long)N * KX * 64, (long)N * KX);
        const int voff = lane * 8;
#define SPCIES_V(P, blk, s) (P).ld((blk) * KS + (s), voff)
#define SPCIES_VST(P, blk, s, x) (P).st((blk) * KS + (s), voff, (x))
        // ---- per-instance setup: x0 and c2 = [T xr; S ur] (:128-137)
        double z2[KS];  // z2 of this iteration; the previous one (the one P1 of this iteration saw) is parked in LDS
        double *z2park = lds + LY::LDS_D + wave * (KS * 64) + lane;
#define Z2O(s) z2park[(s) * 64]
        const SlabBuf C2(C2g + tile * (long)KS * 64, KS);  // c2 = [T xr; S ur], parked in HBM (read once per iteration)
        auto load_x0 = [&](int s) -> double {
            const int row = 4 * s + g;
            return (valid && row < n) ? x0g[inst * n + row] : 0.0;
        };
        {
            double xu[KS], c2[KS];
            const double *xrp = p.ref_stride ? xrg + inst * n : xrg;
            const double *urp = p.ref_stride ? urg + inst * m : urg;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int row = 4 * s + g;
                xu[s] = 0.0;
                if (valid && row < n) {
                    xu[s] = xrp[row];
                } else if (valid && row < nm) {
                    xu[s] = urp[row - n];
                }
                c2[s] = 0.0;
                z2[s] = 0.0;
            }
            __syncthreads();  // invariant region visible / previous group done with the ring
            int tix = LY::T_TS;
            double2 cur;
            prod<KS, KS, DENSE>(c2, xu, lds, ao, tix, cur);
#pragma unroll
            for (int s = 0; s < KS; s++) C2.st(s, lane * 8, c2[s]);
        }
        stg.issue(seq);
        stg.commit(ring);
        __syncthreads();
        int slot = 0, sq = 0;  // sq: index of the chunk in the current slot
        int ao_l = ao;  // laundered once per stage: keeps LICM from hoisting the stage-invariant LDS block reads
        auto next_chunk = [&]() __attribute__((always_inline)) {  // prefetch the chunk after sq (the sweep-A chunks 0..N are visited once: 3N wraps to N + 1)
            asm volatile("" : "+v"(ao_l));
            const int nx = (sq + 1 == n_seq) ? N + 1 : sq + 1;
            stg.issue(seq + (long)nx * LY::CHD);
        };
        auto commit_chunk = [&]() __attribute__((always_inline)) {
            stg.commit(ring + (slot ^ 1) * LY::CHD);
            __syncthreads();
            slot ^= 1;
            sq = (sq + 1 == n_seq) ? N + 1 : sq + 1;
        };
        // P1 of one row (:97-117): stage 0 carries `add` = rho_0 x0 - lambda_0, stage N the x_s = x_N, u_s = u_N rows
        auto p1 = [&](const double *K, int s, double z3v, double l1, double add, double zz) -> double {
            const double v = (SPCIES_K(K, LY::K_RHO, s) * (z3v + zz) + l1 + add) * SPCIES_K(K, LY::K_H1I, s);
            return fmin(fmax(v, SPCIES_K(K, LY::K_LB, s)), SPCIES_K(K, LY::K_UB, s));
        };
        auto p1_mid = [&](const double *K, int s, double z3v, double l1, double zz) -> double {
            const double v = (SPCIES_K(K, LY::K_RHO, s) * (z3v + zz) + l1) * SPCIES_K(K, LY::K_H1I, s);
            return fmin(fmax(v, SPCIES_K(K, LY::K_LB, s)), SPCIES_K(K, LY::K_UB, s));
        };
        auto p1_N = [&](const double *K, int s, double z3v, double l1, double l2, double zz) -> double {
            const double r = SPCIES_K(K, LY::K_RHO, s), rs = SPCIES_K(inv_rc, LY::C_RHOS, s);
            const double v = (r * z3v + (r + rs) * zz + l1 + l2) * SPCIES_K(K, LY::K_H1I, s);
            return fmin(fmax(v, SPCIES_K(K, LY::K_LB, s)), SPCIES_K(K, LY::K_UB, s));
        };

        bool active = valid;
        int kk = 0;
        double q2[KS];
        // ======================= sweep A, once: q2 of iteration 1 (z3 = lambda = z2 = 0: zero-filled by the launcher) ===========
        {  // stage N
            next_chunk();
            const double *K = ring + slot * LY::CHD + LY::NT_PAD * 16;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const double z3v = SPCIES_V(Z3, N, s);
                const double l1 = SPCIES_V(LAM, N + 1, s), l2 = SPCIES_V(LAM, N + 2, s);
                const double r = SPCIES_K(K, LY::K_RHO, s), rs = SPCIES_K(inv_rc, LY::C_RHOS, s);
                const double v = p1_N(K, s, z3v, l1, l2, z2[s]);
                q2[s] = r * z3v - (r + rs) * v + l1 + l2 + C2.ld(s, voff);
            }
            commit_chunk();
        }
        for (int l = 0; l < N; l++) {
            next_chunk();
            const double *K = ring + slot * LY::CHD + LY::NT_PAD * 16;
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const double z3v = SPCIES_V(Z3, l, s), l1 = SPCIES_V(LAM, l + 1, s);
                double v;
                if (l == 0) {
                    v = p1(K, s, z3v, l1, SPCIES_K(inv_rc, LY::C_RHO0, s) * load_x0(s) - SPCIES_V(LAM, 0, s), z2[s]);
                } else {
                    v = p1_mid(K, s, z3v, l1, z2[s]);
                }
                q2[s] += SPCIES_K(K, LY::K_RHO, s) * (z3v - v) + l1;
            }
            commit_chunk();
        }
        while (true) {
            kk += 1;
            bool res = false;
            {  // z2 = W2 q2 (:145-151) and its part of the exit test (:408-415)
                double z2n[KS];
#pragma unroll
                for (int s = 0; s < KS; s++) z2n[s] = 0.0;
                int tix = LY::T_W2;
                double2 cur;
                prod<KS, KS, DENSE>(z2n, q2, lds, ao_l, tix, cur);
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    res |= fabs(z2[s] - z2n[s]) > tol;
                    Z2O(s) = z2[s];
                    z2[s] = z2n[s];
                }
            }
            // ======================= sweep B: z1, q3, right-hand side, forward substitution =======================
            {
                double q3c[KS], mup[KX], z3n[KS], ln[KS];
#pragma unroll
                for (int s = 0; s < KX; s++) mup[s] = 0.0;
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    z3n[s] = SPCIES_V(Z3, 1, s);
                    ln[s] = SPCIES_V(LAM, 2, s);
                }
                for (int l = 0; l < N; l++) {
                    next_chunk();
                    const double *ch = ring + slot * LY::CHD;
                    const double *Ka = ch + LY::NT_PAD * 16, *Kb = Ka + LY::KD;
                    if (l == 0) {
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            const double l1 = SPCIES_V(LAM, 1, s);
                            const double z1v = p1(Ka, s, SPCIES_V(Z3, 0, s), l1,
                                                  SPCIES_K(inv_rc, LY::C_RHO0, s) * load_x0(s) - SPCIES_V(LAM, 0, s), Z2O(s));
                            q3c[s] = SPCIES_K(Ka, LY::K_RHO, s) * (z2[s] - z1v) + l1;
                        }
                    }
                    double q3n[KS], t[KS], y[KX];
                    if (l + 1 < N) {
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            const double z1v = p1_mid(Kb, s, z3n[s], ln[s], Z2O(s));
                            q3n[s] = SPCIES_K(Kb, LY::K_RHO, s) * (z2[s] - z1v) + ln[s];
                            if (l + 2 <= N) {
                                z3n[s] = SPCIES_V(Z3, l + 2, s);
                                ln[s] = SPCIES_V(LAM, l + 3, s);
                            }
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            const double z1v = p1_N(Kb, s, z3n[s], ln[s], SPCIES_V(LAM, N + 2, s), Z2O(s));
                            q3n[s] = SPCIES_K(Kb, LY::K_RHO, s) * (z2[s] - z1v) + ln[s];
                        }
                    }
                    if constexpr (DIAG) {
#pragma unroll
                        for (int s = 0; s < KS; s++) t[s] = SPCIES_K(Ka, LY::K_H3I, s) * q3c[s];
#pragma unroll
                        for (int s = 0; s < KX; s++) y[s] = (4 * s + g < n) ? SPCIES_K(Kb, LY::K_H3I, s) * q3n[s] : 0.0;
                        int tix = LY::T_NAB;
                        double2 cur;
                        prod<KX, KS, DENSE>(y, t, lds, ao_l, tix, cur);
                    } else {  // y = Qinv_{l+1} q3_{l+1}[x] - AB H3inv_l q3_l (:184-217; the padded columns of Qinv are zero)
                        double qx[KX];
#pragma unroll
                        for (int s = 0; s < KX; s++) {
                            y[s] = 0.0;
                            qx[s] = q3n[s];
                        }
                        int tix = LY::NTS;
                        double2 cur;
                        prod<KX, KX, DENSE>(y, qx, ch, ao_l, tix, cur);
                        prod<KX, KS, DENSE>(y, q3c, ch, ao_l, tix, cur);
                    }
                    double mu[KX];
#pragma unroll
                    for (int s = 0; s < KX; s++) mu[s] = 0.0;
                    {
                        int tix = 0;
                        double2 cur;
                        prod<KX, KX, LOWER>(mu, y, ch, ao_l, tix, cur);
                        prod<KX, KX, DENSE>(mu, mup, ch, ao_l, tix, cur);
                    }
#pragma unroll
                    for (int s = 0; s < KX; s++) {
                        MU.st(l * KX + s, voff, mu[s]);
                        mup[s] = mu[s];
                    }
#pragma unroll
                    for (int s = 0; s < KS; s++) q3c[s] = q3n[s];
                    commit_chunk();
                }
            }
            // ============ sweep C: backward substitution, z1 again, z3, residuals, duals; P1 and q2 of the next iteration ============
            {
                double mun[KX];
#pragma unroll
                for (int s = 0; s < KX; s++) mun[s] = 0.0;
                // z3_t, residual, lambda_{t+1} of one stage (:289-320, :371-402) and its term of the next q2 (:97-143)
                auto finish_stage = [&](int t, const double *K, const double *hblk, const double (&mu_sub)[KX], const double (&mu_abt)[KX],
                                        const double (&lam)[KS], const double (&z3o)[KS]) __attribute__((always_inline)) {
                    double v[KS], z1v[KS], ex[KS];  // ex: lambda_{N+2} (t = N) / lambda_0 (t = 0)
                    if (t == N) {
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            ex[s] = SPCIES_V(LAM, N + 2, s);
                            z1v[s] = p1_N(K, s, z3o[s], lam[s], ex[s], Z2O(s));
                        }
                    } else if (t == 0) {
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            ex[s] = SPCIES_V(LAM, 0, s);
                            z1v[s] = p1(K, s, z3o[s], lam[s], SPCIES_K(inv_rc, LY::C_RHO0, s) * load_x0(s) - ex[s], Z2O(s));
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < KS; s++) z1v[s] = p1_mid(K, s, z3o[s], lam[s], Z2O(s));
                    }
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        v[s] = SPCIES_K(K, LY::K_RHO, s) * (z2[s] - z1v[s]) + lam[s];
                        if (s < KX) v[s] -= mu_sub[s < KX ? s : 0];
                    }
                    {
                        int tix = LY::T_ABT;
                        double2 cur;
                        prod<KS, KX, DENSE>(v, mu_abt, lds, ao_l, tix, cur);
                    }
                    double z3v[KS];
                    if constexpr (DIAG) {
#pragma unroll
                        for (int s = 0; s < KS; s++) z3v[s] = -SPCIES_K(K, LY::K_H3I, s) * v[s];
                    } else {  // z3 = -H3inv_t v (:321-366); t = 0 reads the stage-invariant copy
#pragma unroll
                        for (int s = 0; s < KS; s++) z3v[s] = 0.0;
                        double2 cur;
                        if (t == 0) {
                            int tix = LY::T_H0;
                            prod<KS, KS, DENSE>(z3v, v, lds, ao_l, tix, cur);
                        } else {
                            int tix = LY::NTS;
                            prod<KS, KS, DENSE>(z3v, v, hblk, ao_l, tix, cur);
                        }
                    }
                    double ln_[KS];
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        const double z3 = z3v[s];
                        const double r = z2[s] + z3 - z1v[s];
                        ln_[s] = lam[s] + SPCIES_K(K, LY::K_RHO, s) * r;
                        res |= (fabs(r) > tol) | (fabs(z3o[s] - z3) > tol);
                        v[s] = z3;
                    }
                    if (active) {
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            SPCIES_VST(Z3, t, s, v[s]);
                            SPCIES_VST(LAM, t + 1, s, ln_[s]);
                            if constexpr (WANT_Z1) SPCIES_VST(Z1, t, s, z1v[s]);
                        }
                    }
                    // first / last residual rows and their multipliers (:374-376, 386-388, 391-393, 403-405), then P1 of
                    // iteration k + 1 from (z3, lambda, z2) of iteration k and this stage's term of q2
                    if (t == N) {
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            const double rN = z2[s] - z1v[s];
                            ex[s] = ex[s] + SPCIES_K(inv_rc, LY::C_RHOS, s) * rN;
                            res |= fabs(rN) > tol;
                            const double r = SPCIES_K(K, LY::K_RHO, s), rs = SPCIES_K(inv_rc, LY::C_RHOS, s);
                            const double z1n = p1_N(K, s, v[s], ln_[s], ex[s], z2[s]);
                            q2[s] = r * v[s] - (r + rs) * z1n + ln_[s] + ex[s] + C2.ld(s, voff);
                        }
                        if (active) {
#pragma unroll
                            for (int s = 0; s < KS; s++) SPCIES_VST(LAM, N + 2, s, ex[s]);
                        }
                    } else if (t == 0) {
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            const int row = 4 * s + g;
                            const double x0v = load_x0(s);
                            const double r0 = (row < n) ? z1v[s] - x0v : 0.0;
                            ex[s] = ex[s] + SPCIES_K(inv_rc, LY::C_RHO0, s) * r0;  // rows >= n stay 0
                            res |= fabs(r0) > tol;
                            if (active && row >= n && row < nm) u_out[inst * m + (row - n)] = z1v[s];  // u_opt (:477), last write wins
                            const double z1n = p1(K, s, v[s], ln_[s], SPCIES_K(inv_rc, LY::C_RHO0, s) * x0v - ex[s], z2[s]);
                            q2[s] += SPCIES_K(K, LY::K_RHO, s) * (v[s] - z1n) + ln_[s];
                        }
                        if (active) {
#pragma unroll
                            for (int s = 0; s < KS; s++) SPCIES_VST(LAM, 0, s, ex[s]);
                        }
                    } else {
#pragma unroll
                        for (int s = 0; s < KS; s++) {
                            const double z1n = p1_mid(K, s, v[s], ln_[s], z2[s]);
                            q2[s] += SPCIES_K(K, LY::K_RHO, s) * (v[s] - z1n) + ln_[s];
                        }
                    }
                };
                double lamn[KS], z3on[KS], mufn[KX];  // stage l + 1 vectors / mu_l, in flight
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    lamn[s] = SPCIES_V(LAM, N + 1, s);
                    z3on[s] = SPCIES_V(Z3, N, s);
                }
#pragma unroll
                for (int s = 0; s < KX; s++) mufn[s] = MU.ld((N - 1) * KX + s, voff);
                for (int l = N - 1; l >= 0; l--) {
                    next_chunk();
                    const double *ch = ring + slot * LY::CHD;
                    const double *Ka = ch + LY::NT_PAD * 16, *Kb = Ka + LY::KD;
                    double lam[KS], z3o[KS], muf[KX];
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        lam[s] = lamn[s];
                        z3o[s] = z3on[s];
                        lamn[s] = SPCIES_V(LAM, l + 1, s);  // stage l: next pass, or finish_stage(0)
                        z3on[s] = SPCIES_V(Z3, l, s);
                    }
#pragma unroll
                    for (int s = 0; s < KX; s++) {
                        muf[s] = mufn[s];
                        if (l > 0) mufn[s] = MU.ld((l - 1) * KX + s, voff);
                    }
                    double mu[KX];
#pragma unroll
                    for (int s = 0; s < KX; s++) mu[s] = 0.0;
                    {
                        int tix = 0;
                        double2 cur;
                        prod<KX, KX, UPPER>(mu, muf, ch, ao_l, tix, cur);
                        prod<KX, KX, DENSE>(mu, mun, ch, ao_l, tix, cur);
                    }
                    finish_stage(l + 1, Kb, ch, mu, mun, lam, z3o);
#pragma unroll
                    for (int s = 0; s < KX; s++) mun[s] = mu[s];
                    if (l == 0) {
                        double zero[KX];
#pragma unroll
                        for (int s = 0; s < KX; s++) zero[s] = 0.0;
                        finish_stage(0, Ka, ch, zero, mun, lamn, z3on);
                    }
                    commit_chunk();
                }
            }
            // ======================= exit (:408-449) =======================
            const bool res_inst = or_over_rows(res, c);
            const bool done_now = active && (!res_inst || kk >= p.k_max);
            if (done_now) {
#pragma unroll
                for (int s = 0; s < KS; s++) {
                    const int row = 4 * s + g;
                    if (z2_out && row < nm) z2_out[inst * nm + row] = z2[s];
                }
                if (g == 0) {
                    k_out[inst] = kk;
                    e_out[inst] = res_inst ? -1 : 1;
                }
                active = false;
            }
            if (!__syncthreads_or(active ? 1 : 0)) break;
        }
    }
#undef Z2O
#undef SPCIES_K
#undef SPCIES_V
#undef SPCIES_VST
}

// lambda copy-out with the reference's packing (code_MPCT_EADMM_C.c:495-513): the first n entries of every
// (n+m)-wide block, written contiguously; the rest of the (N+3)(n+m) record stays zero.
__global__ __launch_bounds__(256) void eadmm_g_pack_lambda_kernel(const double *__restrict__ LAM, long B, int N, int n, int nm,
                                                                   int KS, double *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long rows = (long)(N + 3) * nm;
    if (i >= B * rows) return;
    const long inst = i / rows;
    const int e = (int)(i % rows);
    double v = 0.0;
    if (e < (N + 3) * n) {
        const int b = e / n, row = e % n;
        const long tile = inst / 16;
        v = LAM[((tile * (N + 3) + b) * KS + row / 4) * 64 + 16 * (row % 4) + (inst % 16)];
    }
    out[i] = v;
}

#define SPCIES_G4_EADMM_SHAPES(X) X(1, 1) X(1, 2) X(2, 2) X(2, 3) X(3, 3) X(3, 4) X(4, 4) X(4, 5) X(5, 5) X(5, 6) X(6, 6)

inline int eadmm_plan_build(Plan &p, const AdmmHost &a, const EadmmGHost &h) {
    p.ok = false;
    const int KX = (a.n + 3) / 4, KS = (a.n + a.m + 3) / 4;
    if (a.N < 2) { p.why = "N < 2"; return 0; }
#define X(KKX, KKS)               \
    if (KX == KKX && KS == KKS)   \
        return h.diag ? eadmm_plan_build_shape<KKX, KKS, true>(p, a, h) : eadmm_plan_build_shape<KKX, KKS, false>(p, a, h);
    SPCIES_G4_EADMM_SHAPES(X)
#undef X
    p.why = "MFMA4G EADMM kernel not instantiated for this (ceil(n/4), ceil((n+m)/4))";
    return 0;
}

inline size_t eadmm_state_bytes(const Plan &p, const AdmmHost &a, long B) {
    return (size_t)padded_tiles(B) * ((size_t)(3 * a.N + 6) * p.KS + (size_t)a.N * p.KX) * 64 * sizeof(double);
}

template <int KX, int KS, bool DIAG>
static int launch_eadmm_g_shape(Plan &pl, const AdmmHost &a, const Args &args, const double *x0, const double *xr,
                                const double *ur, double *state, double *u, int *k, int *e, double *z1, double *z2,
                                double *z3, double *lam, hipStream_t st) {
    using LY = EadmmGLayout<KX, KS, DIAG>;
#ifndef SPCIES_G4_EADMM_WGS_BIG
#define SPCIES_G4_EADMM_WGS_BIG 2
#endif
    constexpr int WGS = (KS >= 5) ? SPCIES_G4_EADMM_WGS_BIG : 2;
    const long tiles = padded_tiles(args.B);
    const int N = a.N, nm = a.n + a.m;
    double *Z1 = state, *Z3 = Z1 + tiles * (long)(N + 1) * KS * 64, *LAM = Z3 + tiles * (long)(N + 1) * KS * 64;
    double *MU = LAM + tiles * (long)(N + 3) * KS * 64, *C2 = MU + tiles * (long)N * KX * 64;
    const long wgs = std::min(tiles / 4, (long)pl.num_cu * pick_wgs(tiles / 4, pl.num_cu, WGS));
    const size_t shmem = (LY::LDS_D + 4 * KS * 64) * sizeof(double);  // + z2 of the previous iteration, per wavefront
    // iteration 1 starts from z3 = lambda = 0 (:85-95); Z3 and LAM are adjacent
    SPCIES_HIP_CHECK(hipMemsetAsync(Z3, 0, (size_t)tiles * (size_t)(2 * N + 4) * KS * 64 * sizeof(double), st));
    if (shmem > 64 * 1024) {
        if (z1) SPCIES_HIP_CHECK(hipFuncSetAttribute((const void *)eadmm_g_kernel<KX, KS, WGS, true, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
        else SPCIES_HIP_CHECK(hipFuncSetAttribute((const void *)eadmm_g_kernel<KX, KS, WGS, false, DIAG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
    }
    if (z1)
        hipLaunchKernelGGL((eadmm_g_kernel<KX, KS, WGS, true, DIAG>), dim3((unsigned)wgs), dim3(256), shmem, st, args, pl.d_table, x0, xr,
                           ur, Z1, Z3, LAM, MU, C2, u, k, e, z2);
    else
        hipLaunchKernelGGL((eadmm_g_kernel<KX, KS, WGS, false, DIAG>), dim3((unsigned)wgs), dim3(256), shmem, st, args, pl.d_table, x0, xr,
                           ur, Z1, Z3, LAM, MU, C2, u, k, e, z2);
    SPCIES_HIP_CHECK(hipGetLastError());
    const long tz = args.B * (long)(N + 1) * nm;
    if (z1)
        hipLaunchKernelGGL(tile_state_to_aos_kernel, dim3((unsigned)((tz + 255) / 256)), dim3(256), 0, st, Z1, args.B, N + 1, KS,
                           nm, z1);
    if (z3)
        hipLaunchKernelGGL(tile_state_to_aos_kernel, dim3((unsigned)((tz + 255) / 256)), dim3(256), 0, st, Z3, args.B, N + 1, KS,
                           nm, z3);
    if (lam) {
        const long tl = args.B * (long)(N + 3) * nm;
        hipLaunchKernelGGL(eadmm_g_pack_lambda_kernel, dim3((unsigned)((tl + 255) / 256)), dim3(256), 0, st, LAM, args.B, N, a.n,
                           nm, KS, lam);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

inline int launch_eadmm_g(Plan &pl, const AdmmHost &a, const double *x0, const double *xr, const double *ur, int ref_stride,
                          long B, double *state, double *u, int *k, int *e, double *z1, double *z2, double *z3, double *lam,
                          hipStream_t st) {
    if (!pl.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4G variant unavailable: %s", pl.why.c_str());
    Args args{a.n, a.m, a.N, a.k_max, a.tol, B, ref_stride};
#define X(KKX, KKS)                                                                                                        \
    if (pl.KX == KKX && pl.KS == KKS)                                                                                      \
        return pl.general ? launch_eadmm_g_shape<KKX, KKS, false>(pl, a, args, x0, xr, ur, state, u, k, e, z1, z2, z3, lam, st) \
                          : launch_eadmm_g_shape<KKX, KKS, true>(pl, a, args, x0, xr, ur, state, u, k, e, z1, z2, z3, lam, st);
    SPCIES_G4_EADMM_SHAPES(X)
#undef X
    return fail(SPCIES_HIP_ENOSUP, "MFMA4G EADMM kernel not instantiated for KX=%d KS=%d", pl.KX, pl.KS);
}

}  // namespace g4
}  // namespace spcies
