// Variant MFMA4 of the banded-Cholesky ADMM solver: same design as admm_mfma.hpp (16 instances per
// wavefront, state in registers, folded block matrices in LDS) on v_mfma_f64_4x4x4_4b_f64.
//
// Why the small instruction: on gfx950 the 16x16x4 FP64 MFMA issues every ~75-80 cycles for 2048 flop
// and the 4x4x4 (4 blocks) one every ~17 cycles for 512 flop (profiles/r01_microbench_f64_v3.txt) - the
// same rate per flop - but with 4x4 blocks nothing is padded to 16 rows (n = 12 is 3 slabs, not 3/4 of a
// 16-row tile) and structurally zero blocks are skipped (Beta^-1 is triangular):  786 small MFMAs per
// iteration at C2 = 13.2 k matrix-pipe cycles against 279 large ones = 21-22 k.
//
// Lane layout, measured with tools/probe_mfma4x4x4.hip: block b = (lane%16)/4, and
//     A[i][k] at lane 16k + 4b + i,   B[k][j] at lane 16k + 4b + j,   D[i][j] at lane 16i + 4b + j.
// With instance c = lane%16 (= 4b + j) and g = lane/16 a register holds rows 4s+g of "slab" s of a
// vector for 16 instances - EXACTLY the register layout of the 16x16x4 variant (VR[r] = slab r), and
// again D layout == B layout, so products chain with no data movement.  The A operand is the 4x4 block
// M[4I+i][4J+k], replicated over the 4 blocks b (CBSZ/ABID broadcast is ignored for f64 on gfx950).
//
// A operands are consumed strictly in stream order: the host lays the blocks out in the exact order
// the kernel issues them (one pass per iteration), so the kernel walks one running index through LDS.
#pragma once
#include <cmath>

#include "admm_mfma.hpp"

namespace spcies {

// [rtc-begin]
// Shapes of the block products.  pattern: 0 dense, 1 lower (J <= I), 2 upper (J >= I).
struct Prod4 {
    int I0, I1, J0, J1, pat;
    __host__ __device__ constexpr bool nz(int I, int J) const {
        return I >= I0 && I < I1 && J >= J0 && J < J1 && (pat == 0 || (pat == 1 ? J <= I : J >= I));
    }
    __host__ __device__ constexpr int count() const {
        int c = 0;
        for (int J = 0; J < 4; J++)
            for (int I = 0; I < 4; I++) c += nz(I, J) ? 1 : 0;
        return c;
    }
};

struct Mfma4Layout {
    int N, KX, KS;
    bool terminal;
    __host__ __device__ constexpr bool stage_exists(int t) const { return t >= 0 && (t < N || (t == N && terminal)); }
    __host__ __device__ constexpr Prod4 F2(int l) const { return Prod4{0, KX, l == 0 ? KS - 1 : 0, KS, 0}; }
    __host__ __device__ constexpr Prod4 F1(int l) const { return Prod4{0, KX, 0, KX, (l + 1 == N) ? 0 : 1}; }
    __host__ __device__ constexpr bool hasF1(int l) const { return terminal || l < N - 1; }
    __host__ __device__ constexpr Prod4 F3() const { return Prod4{0, KX, 0, KX, 0}; }
    __host__ __device__ constexpr Prod4 B1() const { return Prod4{0, KX, 0, KX, 2}; }
    __host__ __device__ constexpr Prod4 B2() const { return Prod4{0, KX, 0, KX, 0}; }
    __host__ __device__ constexpr Prod4 Zmid() const { return Prod4{0, KS, 0, KX, 0}; }
    __host__ __device__ constexpr Prod4 ZN() const { return Prod4{0, KX, 0, KX, 0}; }
    __host__ __device__ constexpr Prod4 Z0() const { return Prod4{KS - 1, KS, 0, KX, 0}; }
    __host__ __device__ constexpr Prod4 S() const { return Prod4{0, KX, 0, KX, 0}; }
    // number of blocks in the per-iteration stream
    __host__ __device__ constexpr int stream_tiles() const {
        int c = 0;
        for (int l = 0; l < N; l++) c += F2(l).count() + (hasF1(l) ? F1(l).count() : 0) + (l >= 1 ? F3().count() : 0);
        for (int l = N - 1; l >= 0; l--) {
            c += B1().count() + (l < N - 1 ? B2().count() : 0);
            const int t = l + 2;
            if (stage_exists(t)) c += (t == N) ? ZN().count() : Zmid().count();
        }
        c += Zmid().count() + Z0().count();
        return c;
    }
    // the stream is padded to an even number of blocks (blocks are stored in pairs)
    __host__ __device__ constexpr int stream_pairs() const { return (stream_tiles() + 1) / 2; }
    __host__ __device__ constexpr int setup_base() const { return 2 * stream_pairs(); }  // S0, then ST (lax) or SN (equ)
    __host__ __device__ constexpr int n_tiles() const { return setup_base() + 2 * S().count(); }
    enum { RC_NEGHD_MID = 0, RC_NEGHD_0, RC_LB_MID, RC_UB_MID, RC_LB_0, RC_UB_0, RC_LB_N, RC_UB_N, RC_QR, RC_COUNT };
    __host__ __device__ constexpr int rc_off(int i) const { return n_tiles() * 16 + i * 16; }
    __host__ __device__ constexpr int total_doubles() const { return n_tiles() * 16 + RC_COUNT * 16; }
};
// [rtc-end]

struct Mfma4Plan {
    bool ok = false;
    bool unit = false;       // unit-box coordinates (admm_mfma4u.hpp): the table holds scaled blocks and Mfma4uRC row constants
    bool needs_rtc = false;  // tables are packed, but no kernel of this shape was instantiated at build time (mfma4_rtc.hpp)
    std::string why = "not built";
    bool build_failed = false;  // the variant applies to this controller but its run-time specialisation failed (hiprtc missing, compile error): what SPCIES_HIP_STRICT reacts to
    Mfma4Layout lay{};
    double *d_table = nullptr;
    size_t table_bytes = 0;
    int num_cu = 256;
};

inline void mfma4_plan_free(Mfma4Plan &p) {
    if (p.d_table) hipFree(p.d_table);
    p.d_table = nullptr;
}

inline bool mfma4_shape_instantiated(int N, int KX, int KS);

// unit-box coordinates (admm_mfma4u.hpp): row scalings D = ub - lb and shifts lb per kind of stage, when the bounds allow them
struct Mfma4uScaling {
    bool ok = false;
    std::string why;
    std::vector<double> D_mid, D_0, D_N, lb_mid, lb_0, lb_N;  // 16 entries each; D = 1, lb = 0 on rows that do not exist
};
inline Mfma4uScaling mfma4u_scaling(const AdmmHost &a);
inline void mfma4u_row_constants(const AdmmHost &a, const Mfma4uScaling &sc, const std::vector<double> &hd_mid, const std::vector<double> &hd_0,
                                 double *rc0);
inline int mfma4u_rc_count();

inline int mfma4_plan_build(Mfma4Plan &p, const AdmmHost &a) {
    using namespace hostla;
    const int n = a.n, m = a.m, N = a.N, nm = n + m;
    p.ok = false;
    if (nm > 16) { p.why = "n+m > 16 needs more than 4 slabs (not built yet)"; return 0; }
    Mfma4Layout L{N, (n + 3) / 4, (nm + 3) / 4, a.terminal};
    if (n / 4 != L.KS - 1) { p.why = "u rows must sit inside the last slab (n % 4 + m <= 4)"; return 0; }
    if (N < 3) { p.why = "N < 3"; return 0; }
    p.needs_rtc = !mfma4_shape_instantiated(N, L.KX, L.KS);
    // registers: one slab (2 VGPRs) per stage vector and block of y / mu; 109 slabs at C2 leave ~35 registers of working set
    if (p.needs_rtc && (N + 1) * L.KS + N * L.KX > 112) { p.why = "state does not fit the register file (use MFMA4G)"; p.needs_rtc = false; return 0; }
    for (int l = 1; l < N - 1; l++)
        for (int j = 0; j < nm; j++)
            if (a.Hi[(size_t)l * nm + j] != a.Hi[j]) { p.why = "Hi differs between stages (vector rho?)"; return 0; }
    p.lay = L;
    const Mfma4uScaling sc = mfma4u_scaling(a);
    { const char *ev = getenv("SPCIES_MFMA4_UNIT"); p.unit = sc.ok && !(ev && ev[0] == '0'); }
    // the unit-box table carries 20 rows of constants against 9 (1.4 KB more): a shape whose plain table just fits the LDS keeps MFMA4
    // in the plain form instead of losing it
    auto table_bytes_of = [&](bool unit) { return ((size_t)L.n_tiles() * 16 + (size_t)(unit ? mfma4u_rc_count() : (int)Mfma4Layout::RC_COUNT) * 16) * sizeof(double); };
    if (p.unit && table_bytes_of(true) > 160 * 1024 - 512 && table_bytes_of(false) <= 160 * 1024 - 512) p.unit = false;
    std::vector<double> tab((size_t)L.n_tiles() * 16 + (size_t)(p.unit ? mfma4u_rc_count() : (int)Mfma4Layout::RC_COUNT) * 16, 0.0);
    int cursor = 0;
    // append the non-zero 4x4 blocks of M in issue order (J outer, I inner); a block that the pattern
    // declares zero must really be zero
    bool structure_ok = true;
    auto emit = [&](const Mat &M, const Prod4 &P) {
        for (int J = 0; J < 4; J++)
            for (int I = 0; I < 4; I++) {
                bool in = P.nz(I, J);
                if (!in) {
                    if (I >= P.I0 && I < P.I1 && J >= P.J0 && J < P.J1)
                        for (int i = 0; i < 4; i++)
                            for (int k = 0; k < 4; k++)
                                if (M[(4 * I + i) * 16 + 4 * J + k] != 0.0) structure_ok = false;
                    continue;
                }
                // tiles are stored in PAIRS, element-interleaved, so one ds_read_b128 per lane fetches its
                // element of two consecutive blocks: pair p, element e = 4k+i, tile 2p+h at double 32p + 2e + h
                double *t = tab.data() + (size_t)(cursor / 2) * 32 + (cursor % 2);
                for (int k = 0; k < 4; k++)
                    for (int i = 0; i < 4; i++) t[2 * (k * 4 + i)] = M[(4 * I + i) * 16 + 4 * J + k];
                cursor++;
            }
    };
    Mat AB = zeros(), A = zeros(), HiN = zeros(), Tm = zeros();
    for (int i = 0; i < n; i++)
        for (int j = 0; j < nm; j++) AB[i * 16 + j] = a.AB[(size_t)i * nm + j];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            A[i * 16 + j] = a.AB[(size_t)i * nm + j];
            HiN[i * 16 + j] = a.Hi_N[(size_t)i * n + j];
            Tm[i * 16 + j] = a.T[(size_t)i * n + j];
        }
    std::vector<double> hd_mid(16, 0.0), hd_0(16, 0.0), hdx_mid(16, 0.0);
    for (int j = 0; j < nm; j++) hd_mid[j] = a.Hi[j];
    for (int j = 0; j < n; j++) hdx_mid[j] = a.Hi[j];
    for (int j = 0; j < m; j++) hd_0[n + j] = a.Hi_0[j];
    std::vector<Mat> Bi(N), Al(N - 1);
    for (int l = 0; l < N; l++) {
        Mat U = zeros();
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) {
                double v = a.Beta[((size_t)l * n + i) * n + j];
                U[i * 16 + j] = (i == j) ? 1.0 / v : v;
            }
        Bi[l] = inv_upper(U, n);
    }
    for (int l = 0; l < N - 1; l++) {
        Al[l] = zeros();
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Al[l][i * 16 + j] = a.Alpha[((size_t)l * n + i) * n + j];
    }
    // unit-box coordinates: the blocks that multiply q_hat = rho D s get their columns scaled by rho D, the blocks that produce z
    // their rows by 1 / D
    std::vector<double> cs_mid(16, 1.0), cs_0(16, 1.0), cs_N(16, 1.0), rs_mid(16, 1.0), rs_0(16, 1.0), rs_N(16, 1.0);
    if (p.unit)
        for (int j = 0; j < 16; j++) {
            cs_mid[j] = a.rho * sc.D_mid[j]; cs_0[j] = a.rho * sc.D_0[j]; cs_N[j] = a.rho * sc.D_N[j];
            rs_mid[j] = 1.0 / sc.D_mid[j]; rs_0[j] = 1.0 / sc.D_0[j]; rs_N[j] = 1.0 / sc.D_N[j];
        }
    Mat ABt = transpose(AB);
    const Mat Zmid = scale_rows(neg(scale_rows(ABt, hd_mid)), rs_mid);
    // ---- forward blocks
    for (int l = 0; l < N; l++) {
        Mat BiT = transpose(Bi[l]);
        emit(neg(mul(BiT, scale_cols(scale_cols(AB, l == 0 ? hd_0 : hd_mid), l == 0 ? cs_0 : cs_mid))), L.F2(l));
        if (L.hasF1(l)) {
            Mat Dx = zeros();
            if (l + 1 == N) Dx = scale_cols(HiN, cs_N);
            else for (int j = 0; j < n; j++) Dx[j * 16 + j] = hdx_mid[j] * cs_mid[j];
            emit(mul(BiT, Dx), L.F1(l));
        }
        if (l >= 1) emit(neg(mul(BiT, transpose(Al[l - 1]))), L.F3());
    }
    // ---- backward blocks, each followed by the Z product of stage l+2
    for (int l = N - 1; l >= 0; l--) {
        emit(Bi[l], L.B1());
        if (l < N - 1) emit(neg(mul(Bi[l], Al[l])), L.B2());
        const int t = l + 2;
        if (L.stage_exists(t)) {
            if (t == N) emit(scale_rows(neg(HiN), rs_N), L.ZN());
            else emit(Zmid, L.Zmid());
        }
    }
    emit(Zmid, L.Zmid());                                // stage 1
    emit(scale_rows(neg(scale_rows(ABt, hd_0)), rs_0), L.Z0());  // stage 0
    if (cursor != L.stream_tiles()) return fail(SPCIES_HIP_EINVAL, "MFMA4 packer/stream mismatch (%d vs %d)", cursor, L.stream_tiles());
    cursor = L.setup_base();
    emit(mul(transpose(Bi[0]), A), L.S());               // x0 -> c0
    if (a.terminal) emit(Tm, L.S());                     // xr -> qT
    else emit(neg(transpose(Bi[N - 1])), L.S());         // xr -> cN
    if (!structure_ok) { p.why = "a block expected to be structurally zero is not"; return 0; }
    auto rc = [&](int i) { return tab.data() + L.rc_off(i); };
    if (p.unit) mfma4u_row_constants(a, sc, hd_mid, hd_0, tab.data() + (size_t)L.n_tiles() * 16);
    for (int j = 0; j < 16 && !p.unit; j++) {
        rc(Mfma4Layout::RC_NEGHD_MID)[j] = -hd_mid[j];
        rc(Mfma4Layout::RC_NEGHD_0)[j] = -hd_0[j];
    }
    for (int j = 0; j < nm && !p.unit; j++) {
        rc(Mfma4Layout::RC_LB_MID)[j] = a.LB[j];
        rc(Mfma4Layout::RC_UB_MID)[j] = a.UB[j];
    }
    for (int j = 0; j < m && !p.unit; j++) {
        rc(Mfma4Layout::RC_LB_0)[n + j] = a.LB[n + j];
        rc(Mfma4Layout::RC_UB_0)[n + j] = a.UB[n + j];
    }
    for (int j = 0; j < n && !p.unit; j++) {
        rc(Mfma4Layout::RC_LB_N)[j] = a.LB[j];
        rc(Mfma4Layout::RC_UB_N)[j] = a.UB[j];
        rc(Mfma4Layout::RC_QR)[j] = a.Q[j];
    }
    for (int j = 0; j < m && !p.unit; j++) rc(Mfma4Layout::RC_QR)[n + j] = a.R[j];
    for (double x : tab)
        if (!std::isfinite(x)) { p.why = "non-finite folded constant (singular Beta block?)"; return 0; }
    p.table_bytes = tab.size() * sizeof(double);
    if (p.table_bytes > 160 * 1024 - 512) { p.why = "block table exceeds the 160 KB LDS"; return 0; }
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_table, p.table_bytes + 64));  // + a dump word for masked-off stores
    SPCIES_HIP_CHECK(hipMemcpy(p.d_table, tab.data(), p.table_bytes, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    int dev = 0;
    SPCIES_HIP_CHECK(hipGetDevice(&dev));
    SPCIES_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    p.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (p.needs_rtc) {
        p.why = "MFMA4 kernel not instantiated for this (N, n, m): select the MFMA4 variant (or set SPCIES_HIP_RTC=1) to compile it at run time";
        return 0;
    }
    p.ok = true;
    p.why.clear();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Device
// ---------------------------------------------------------------------------------------------
// [rtc-begin]
template <int N, int KX, int KS, bool TERMINAL, bool WANT_SOL>
__global__ __launch_bounds__(256, 1) void admm_mfma4_kernel(MfmaArgs p, const double *__restrict__ table_g,
                                                            const double *__restrict__ x0g,
                                                            const double *__restrict__ xrg,
                                                            const double *__restrict__ urg, double *__restrict__ u_out,
                                                            int *__restrict__ k_out, int *__restrict__ e_out,
                                                            double *__restrict__ z_out, double *__restrict__ v_out,
                                                            double *__restrict__ lam_out, double *__restrict__ dump) {
    constexpr Mfma4Layout LL{N, KX, KS, TERMINAL};
#ifdef SPCIES_RTC_STATIC_LDS  // run-time compiled for one shape: the table size is a constant
    __shared__ __attribute__((aligned(16))) double lds[LL.total_doubles()];
#else
    extern __shared__ __attribute__((aligned(16))) double lds[];
#endif
    const int n = p.n, m = p.m, nm = n + m;
    {
        constexpr int total = LL.total_doubles();
        const double2 *src = reinterpret_cast<const double2 *>(table_g);
        double2 *dst = reinterpret_cast<double2 *>(lds);
        for (int i = threadIdx.x; i < total / 2; i += 256) dst[i] = src[i];
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, c = lane & 15;
    const long n_tiles = (p.B + 15) / 16;
    const double rho = p.rho, tol = p.tol;
    const int dim = TERMINAL ? N * nm : N * nm - n;

    // (laundered once per iteration: keeps LICM from hoisting every LDS read out of the iteration loop)
    int ao = g * 4 + (lane & 3), go = g;  // A operand: element [i = lane%4][k = lane/16] of a 16-double block
    auto BLK = [&](int t) -> double { return lds[(t / 2) * 32 + 2 * ao + (t % 2)]; };                 // single block
    // ds_read_b128; two bases kept in registers (laundered once per iteration with ao): the stream is longer than the 64 KB an LDS
    // instruction's immediate offset reaches, and a vector add in front of a read - an isolated vector instruction between MFMAs - costs
    // 12 clocks (profiles/r03_microbench_issue.txt, admm_mfma4u.hpp)
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
        const double *r = lds + LL.rc_off(i);
        return d4{r[go], r[4 + go], r[8 + go], r[12 + go]};
    };
#ifdef SPCIES_NO_SEG_BARRIER
#define SPCIES_SEG_BARRIER
#else
#define SPCIES_SEG_BARRIER __builtin_amdgcn_sched_barrier(0)
#endif
    // requested issue pattern of a segment: after every MFMA, SPCIES_MFMA4_IL other VALU instructions and one LDS read -
    // register moves, integer work and the block-stream reads then issue in the shadow of the 16-cycle MFMA before them
#if defined(SPCIES_MFMA4_IL) && SPCIES_MFMA4_IL > 0
#define SPCIES_SEG_PATTERN                                                   \
    _Pragma("unroll") for (int il_ = 0; il_ < 30; il_++) {                   \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                   \
        __builtin_amdgcn_sched_group_barrier(0x002, SPCIES_MFMA4_IL, 0);     \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                   \
    }
#else
#define SPCIES_SEG_PATTERN
#endif
#define MFMA4(acc, a, b) acc = __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (acc), 0, 0, 0)

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
        const d4 qm = RC(Mfma4Layout::RC_QR) * xuv;  // [Q o xr; R o ur]  (negated weights)
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

        // state between iterations: w_t = z_t + lambda_t / rho per stage (see admm_mfma.hpp), y / mu per block
        d4 w[N + 1], mu[N];
#pragma unroll
        for (int t = 0; t <= N; t++) w[t] = d4{0, 0, 0, 0};
        bool active = valid;
        int kk = 0;
        d4 lbm = RC(Mfma4Layout::RC_LB_MID), ubm = RC(Mfma4Layout::RC_UB_MID), nhm = RC(Mfma4Layout::RC_NEGHD_MID);
        auto LBt = [&](int t) -> d4 { return t == 0 ? RC(Mfma4Layout::RC_LB_0) : (t == N ? RC(Mfma4Layout::RC_LB_N) : lbm); };
        auto UBt = [&](int t) -> d4 { return t == 0 ? RC(Mfma4Layout::RC_UB_0) : (t == N ? RC(Mfma4Layout::RC_UB_N) : ubm); };
        auto clampv = [&](const d4 &x, const d4 &lb, const d4 &ub) -> d4 {
            d4 r;
#pragma unroll
            for (int i = 0; i < 4; i++) r[i] = fmin(fmax(x[i], lb[i]), ub[i]);
            return r;
        };

        // A-operand stream: a ring of block pairs kept PF pairs ahead of the MFMA that consumes them (the LDS
        // round trip is ~130 cycles = 8 small MFMAs); it carries over from one iteration to the next.
#ifndef SPCIES_MFMA4_PF
#define SPCIES_MFMA4_PF 8  // (measured at C2 after the residual short-circuit: 6 -> 7.31 ms, 8 -> 7.21, 10 -> 7.31, 12 -> 7.35)
#endif
        constexpr int NP = LL.stream_pairs(), PF = SPCIES_MFMA4_PF;
        static_assert(NP > PF, "ring");
        // a shift register of named values (an indexed array is not promoted to registers here): after
        // unrolling the shifts are register renames
#define SPCIES_R(i) r##i = PAIR(i)
        double2 SPCIES_R(0), SPCIES_R(1), SPCIES_R(2), SPCIES_R(3), cur = r0;
#if SPCIES_MFMA4_PF >= 6
        double2 SPCIES_R(4), SPCIES_R(5);
#endif
#if SPCIES_MFMA4_PF >= 8
        double2 SPCIES_R(6), SPCIES_R(7);
#endif
#if SPCIES_MFMA4_PF >= 10
        double2 SPCIES_R(8), SPCIES_R(9);
#endif
#if SPCIES_MFMA4_PF >= 12
        double2 SPCIES_R(10), SPCIES_R(11);
#endif
#undef SPCIES_R

        while (true) {
            kk += 1;
            const double fz = (kk == 1) ? 0.0 : 1.0, rf = rho * fz;  // cold start: v = lambda = 0 in iteration 1
            asm volatile("" : "+v"(ao), "+v"(go), "+v"(pb0), "+v"(pb1));
            long il = inst;
            asm volatile("" : "+v"(il));
            double *zp = WANT_SOL ? z_out + il * dim + g : nullptr;
            LAUNDER4(lbm); LAUNDER4(ubm);
            int tix = 0;  // running index into the block stream (a compile-time constant at every use once unrolled)
            // acc[I] += M[I][J] x[J] over the non-zero blocks of P, in stream order
            auto prod = [&](d4 &acc, const d4 &x, const Prod4 P) {
#pragma unroll
                for (int J = 0; J < 4; J++)
#pragma unroll
                    for (int I = 0; I < 4; I++)
                        if (P.nz(I, J)) {
                            if (tix % 2 == 0) {
                                const double2 nw = PAIR((tix / 2 + PF) % NP);
                                cur = r0, r0 = r1, r1 = r2, r2 = r3;
#if SPCIES_MFMA4_PF == 4
                                r3 = nw;
#elif SPCIES_MFMA4_PF == 6
                                r3 = r4, r4 = r5, r5 = nw;
#elif SPCIES_MFMA4_PF == 8
                                r3 = r4, r4 = r5, r5 = r6, r6 = r7, r7 = nw;
#elif SPCIES_MFMA4_PF == 10
                                r3 = r4, r4 = r5, r5 = r6, r6 = r7, r7 = r8, r8 = r9, r9 = nw;
#elif SPCIES_MFMA4_PF == 12
                                r3 = r4, r4 = r5, r5 = r6, r6 = r7, r7 = r8, r8 = r9, r9 = r10, r10 = r11, r11 = nw;
#else
#error "SPCIES_MFMA4_PF must be 4, 6, 8, 10 or 12"
#endif
                            }
                            MFMA4(acc[I], (tix % 2 == 0) ? cur.x : cur.y, x[J]);
                            tix++;
                        }
            };
            auto qhat = [&](int t, d4 &cw) -> d4 {
                cw = clampv(w[t], LBt(t), UBt(t));
                return ((t == N) ? qT : qm) + rf * (w[t] - 2.0 * cw);
            };
            d4 cw;
            // ============ forward sweep (software-pipelined: q_hat_{l+2} is formed during block l) ============
            d4 qh = qhat(0, cw);
            d4 qn = qhat(1, cw);
#pragma unroll
            for (int l = 0; l < N; l++) {
                d4 acc = (l == 0) ? c0 : d4{0, 0, 0, 0};
                if constexpr (!TERMINAL) {
                    if (l == N - 1) acc = cN;
                }
                // (q_hat of stage l + 2 in ONE run of vector instructions in front of the trip's products, not spread between them)
                d4 qnn = qn;
                if (LL.stage_exists(l + 2)) qnn = qhat(l + 2, cw);
                SPCIES_SEG_BARRIER;
                prod(acc, qh, LL.F2(l));
                if (LL.hasF1(l)) prod(acc, qn, LL.F1(l));
                if (l >= 1) prod(acc, mu[l - 1], LL.F3());
                mu[l] = acc;
                qh = qn;
                qn = qnn;
                SPCIES_SEG_PATTERN
                SPCIES_SEG_BARRIER;
            }
            // ============ backward sweep ============
            asm volatile("" : "+v"(go));
            LAUNDER4(lbm); LAUNDER4(ubm);
            bool res = false;
            // once every instance of the wavefront has a residual above tol the remaining checks of this iteration cannot change
            // the outcome (the reference leaves its residual loops at the first hit, code_laxMPC_ADMM_C.c:575-620): skipped under a
            // wave-uniform branch - a quarter of the iteration's FP64 vector instructions while the batch is far from converged
            bool all_hit = false;
            auto stage_z = [&](int t, d4 &cwt) -> d4 {
                d4 z;
                const d4 qq = qhat(t, cwt);
                if (t == N) {
                    const d4 wv = qq - mu[N - 1];
                    z = d4{0, 0, 0, 0};
                    prod(z, wv, LL.ZN());
                } else if (t == 0) {
                    z = RC(Mfma4Layout::RC_NEGHD_0) * qq;
                    prod(z, mu[0], LL.Z0());
                } else {
                    z = nhm * (qq - mu[t - 1]);
                    prod(z, mu[t], LL.Zmid());
                }
                return z;
            };
            auto stage_w = [&](int t, const d4 &z, const d4 &cwt) {
                const d4 wn = z + fz * (w[t] - cwt);  // z + lambda/rho
                const d4 vn = clampv(wn, LBt(t), UBt(t));
                // v_old = fz * clamp(w_old); the product folds into the subtraction (exact: fz is 0 or 1)
                if (!all_hit) {
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        res |= (fabs(__builtin_fma(fz, cwt[r], -vn[r])) > tol) | (fabs(z[r] - vn[r]) > tol);
                    unsigned long long hb = __ballot(res);
                    hb |= hb >> 32;
                    hb |= hb >> 16;
                    all_hit = (hb & 0xFFFFull) == 0xFFFFull;
                }
                w[t] = wn;
                if constexpr (WANT_SOL) {
                    const int off = (t == 0) ? -n : (m + (t - 1) * nm);
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 4 * r + g;
                        const bool in = (t == 0) ? (row >= n && row < nm) : (t == N ? row < n : row < nm);
                        double *ptr = (in && active) ? (zp + off + 4 * r) : dump;
                        *ptr = z[r];
                    }
                }
            };
            d4 zc = {0, 0, 0, 0}, cwc = {0, 0, 0, 0};
#pragma unroll
            for (int l = N - 1; l >= 0; l--) {
                const int tp = l + 3;
                if (LL.stage_exists(tp)) stage_w(tp, zc, cwc);
                d4 acc = {0, 0, 0, 0};
                prod(acc, mu[l], LL.B1());
                if (l < N - 1) prod(acc, mu[l + 1], LL.B2());
                const int t = l + 2;
                if (LL.stage_exists(t)) zc = stage_z(t, cwc);
                mu[l] = acc;
                SPCIES_SEG_PATTERN
                SPCIES_SEG_BARRIER;
            }
            {
                asm volatile("" : "+v"(go));
                d4 cw1, cw0;
                const d4 z1 = stage_z(1, cw1);
                const d4 z0 = stage_z(0, cw0);
                stage_w(2, zc, cwc);
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
                    const d4 v0 = clampv(w[0], LBt(0), UBt(0));
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
                            const d4 vt = clampv(w[t], LBt(t), UBt(t));
                            const d4 lt = rho * (w[t] - vt);
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const int row = 4 * r + g;
                                const bool in = (t == 0) ? (row >= n && row < nm) : (t == N ? row < n : row < nm);
                                if (in) {
                                    v_out[il * dim + off + row] = vt[r];
                                    lam_out[il * dim + off + row] = lt[r];
                                }
                            }
                        }
                    }
                    active = false;
                }
            }
            if (!__any(active)) break;
        }
    }
#undef MFMA4
}
// [rtc-end]

// (admm_mfma4u.hpp: the same kernel in unit-box coordinates, chosen when Mfma4Plan::unit; its build-time instantiations live in
// admm_mfma4u.hip - compiled with -amdgpu-mfma-vgpr-form - behind this launcher)
int mfma4u_launch_builtin(int N, int KX, int KS, bool terminal, bool want_sol, dim3 grid, dim3 block, size_t shmem, hipStream_t st,
                          const MfmaArgs &args, const double *table, const double *x0, const double *xr, const double *ur, double *u, int *k,
                          int *e, double *z, double *v, double *lam, double *dump);

#define SPCIES_MFMA4_SHAPES(X) X(10, 2, 2) X(15, 3, 4)

inline bool mfma4_shape_instantiated(int N, int KX, int KS) {
#define X(NN, KKX, KKS) \
    if (N == NN && KX == KKX && KS == KKS) return true;
    SPCIES_MFMA4_SHAPES(X)
#undef X
    return false;
}

#ifndef SPCIES_NO_BUILTIN_LAUNCHERS  // (admm_mfma4u.hip includes these headers for the host-side declarations only: no second copy of the kernels there)
template <int N, int KX, int KS>
static int launch_mfma4_shape(Mfma4Plan &pl, const AdmmHost &a, const MfmaArgs &args, const double *x0, const double *xr,
                              const double *ur, double *u, int *k, int *e, double *z, double *v, double *lam,
                              hipStream_t st) {
    const bool want_sol = (z || v || lam);
    if (want_sol && !(z && v && lam)) return fail(SPCIES_HIP_EINVAL, "MFMA4 variant: pass all of z, v, lambda or none");
    const long n_tiles = (args.B + 15) / 16;
    long wgs = (n_tiles + 3) / 4;
    if (wgs > pl.num_cu) wgs = pl.num_cu;
    const size_t shmem = pl.table_bytes;
    dim3 grid((unsigned)wgs), block(256);
    if (pl.unit) {
        int rc = mfma4u_launch_builtin(N, KX, KS, a.terminal, want_sol, grid, block, shmem, st, args, pl.d_table, x0, xr, ur, u, k, e, z, v, lam,
                                       pl.d_table + pl.table_bytes / sizeof(double));
        if (rc) return rc;
        SPCIES_HIP_CHECK(hipGetLastError());
        return 0;
    }
#define SPCIES_LAUNCH(TERM, SOL)                                                                                     \
    do {                                                                                                             \
        auto kern = admm_mfma4_kernel<N, KX, KS, TERM, SOL>;                                                         \
        /* per device, so set before every launch (a handle may live on any GPU of the process) */                   \
        SPCIES_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                                 160 * 1024));                                                      \
        hipLaunchKernelGGL(kern, grid, block, shmem, st, args, pl.d_table, x0, xr, ur, u, k, e, z, v, lam,           \
                           pl.d_table + pl.table_bytes / sizeof(double));                                            \
    } while (0)
    if (a.terminal) {
        if (want_sol) SPCIES_LAUNCH(true, true); else SPCIES_LAUNCH(true, false);
    } else {
        if (want_sol) SPCIES_LAUNCH(false, true); else SPCIES_LAUNCH(false, false);
    }
#undef SPCIES_LAUNCH
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

inline int launch_mfma4(Mfma4Plan &pl, const AdmmHost &a, const double *x0, const double *xr, const double *ur,
                        int ref_stride, long B, double *u, int *k, int *e, double *z, double *v, double *lam,
                        hipStream_t st) {
    if (!pl.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4 variant unavailable: %s", pl.why.c_str());
    MfmaArgs args{a.n, a.m, a.k_max, a.tol, a.rho, a.rho_i, B, ref_stride};
#define X(NN, KKX, KKS)                                         \
    if (pl.lay.N == NN && pl.lay.KX == KKX && pl.lay.KS == KKS) \
        return launch_mfma4_shape<NN, KKX, KKS>(pl, a, args, x0, xr, ur, u, k, e, z, v, lam, st);
    SPCIES_MFMA4_SHAPES(X)
#undef X
    return fail(SPCIES_HIP_ENOSUP, "MFMA4 kernel not instantiated for N=%d KX=%d KS=%d", pl.lay.N, pl.lay.KX, pl.lay.KS);
}

#endif  // SPCIES_NO_BUILTIN_LAUNCHERS

}  // namespace spcies
