// Variant MFMA of the banded-Cholesky ADMM solver (laxMPC / equMPC): 16 INSTANCES PER WAVEFRONT on
// v_mfma_f64_16x16x4_f64, all solver state in registers, controller constants in LDS.
//
// Why a matrix formulation here: with 16 instances side by side every block
// product of the reference's iteration - AB, Alpha_l, Beta_l times an n-vector
// (code_laxMPC_ADMM_C.c:355-485) - is a true dense contraction [16 x K] x [K x 16 instances].
//
// Register layout ("VR" = 4 doubles per lane): lane = 16*g + c holds, for instance c of the tile,
// rows 4r+g (r = 0..3) of a 16-row padded stage vector [x (n rows); u (m rows); 0].  That is the
// C/D layout of v_mfma_f64_16x16x4_f64 (row = 4*reg + lane/16, col = lane%16) AND, register r taken
// alone, its B-operand layout for k-step r (k = lane/16) - so the output of one block product feeds
// the next with no data movement at all.
//
// The triangular recurrences are folded into dense blocks once, on the host (mfma_plan_build):
//   forward   y_l  = F1_l qh_{l+1} + F2_l qh_l + F3_l y_{l-1} (+ c0 for l = 0)
//   backward  mu_l = B1_l y_l + B2_l mu_{l+1}
//   primal    z_t  = Z_t mu_t - Hd_t o (qh_t - [mu_{t-1}; 0])
// with  F1 = Bi' Dx, F2 = -Bi' AB Hd, F3 = -Bi' Alpha', B1 = Bi, B2 = -Bi Alpha, Z = -Hd AB',
// Bi_l = inverse of the upper-triangular Cholesky block Beta_l.  This re-associates the reference's
// sums (results agree to ~1e-13, tests hold 1e-10 = the reference's own tol_spcies) - the STREAM
// variant is the one that reproduces the reference bit for bit.
#pragma once
#include <cmath>

#include "common.hpp"

namespace spcies {

// [rtc-begin]  (regions between these markers are also compiled at run time: mfma4_rtc.hpp, gen_rtc_src.py)
typedef double d4 __attribute__((ext_vector_type(4)));
// [rtc-end]

// ---------------------------------------------------------------------------------------------
// Tile table layout (shared by the host packer and the kernel).  A "tile" is one MFMA A-operand:
// 64 doubles, tile[lane] = M[lane & 15][4*s + (lane >> 4)] for k-step s of a 16x16 padded matrix M.
// KX = k-steps that cover the x rows (ceil(n/4)), KS = k-steps that cover x and u rows
// (ceil((n+m)/4)), KU0 = first k-step that contains a u row (n/4).
// ---------------------------------------------------------------------------------------------
struct MfmaLayout {
    int N, KX, KS, KU0;
    bool terminal;
    __host__ __device__ constexpr int f2_cnt(int l) const { return l == 0 ? KS - KU0 : KS; }
    __host__ __device__ constexpr int f1_cnt(int l) const { return (terminal || l < N - 1) ? KX : 0; }
    __host__ __device__ constexpr int f3_cnt(int l) const { return l >= 1 ? KX : 0; }
    __host__ __device__ constexpr int fwd_blk(int l) const { return f2_cnt(l) + f1_cnt(l) + f3_cnt(l); }
    __host__ __device__ constexpr int fwd_off(int l) const {
        int o = 0;
        for (int i = 0; i < l; i++) o += fwd_blk(i);
        return o;
    }
    __host__ __device__ constexpr int f2(int l) const { return fwd_off(l); }
    __host__ __device__ constexpr int f1(int l) const { return fwd_off(l) + f2_cnt(l); }
    __host__ __device__ constexpr int f3(int l) const { return fwd_off(l) + f2_cnt(l) + f1_cnt(l); }
    __host__ __device__ constexpr int bwd_base() const { return fwd_off(N); }
    __host__ __device__ constexpr int b1(int l) const { return bwd_base() + l * 2 * KX; }
    __host__ __device__ constexpr int b2(int l) const { return bwd_base() + l * 2 * KX + KX; }
    __host__ __device__ constexpr int z0() const { return bwd_base() + N * 2 * KX; }
    __host__ __device__ constexpr int zmid() const { return z0() + KX; }
    __host__ __device__ constexpr int zN() const { return z0() + 2 * KX; }
    __host__ __device__ constexpr int s0() const { return z0() + 3 * KX; }   // setup: Bi_0' A       (x0 -> c0)
    __host__ __device__ constexpr int sT() const { return s0() + KX; }        // setup: T (negated)   (xr -> qT)
    __host__ __device__ constexpr int sN() const { return s0() + 2 * KX; }    // setup: -Bi_{N-1}'    (xr -> cN, equMPC)
    __host__ __device__ constexpr int n_tiles() const { return s0() + 3 * KX; }
    // row constants (16 doubles each) follow the tiles
    enum { RC_NEGHD_MID = 0, RC_NEGHD_0, RC_LB_MID, RC_UB_MID, RC_LB_0, RC_UB_0, RC_LB_N, RC_UB_N, RC_QR, RC_COUNT };
    __host__ __device__ constexpr int rc_off(int i) const { return n_tiles() * 64 + i * 16; }
    __host__ __device__ constexpr int total_doubles() const { return n_tiles() * 64 + RC_COUNT * 16; }
};

struct MfmaPlan {
    bool ok = false;
    std::string why = "not built";
    MfmaLayout lay{};
    int n = 0, m = 0;
    double *d_table = nullptr;  // tiles + row constants, device
    size_t table_bytes = 0;
    int num_cu = 256;
};

// ---------------------------------------------------------------------------------------------
// Host: fold the reference-style ingredients into the tile table.
// ---------------------------------------------------------------------------------------------
namespace hostla {
typedef std::vector<double> Mat;  // 16x16 row-major, zero padded
inline Mat zeros() { return Mat(256, 0.0); }
inline Mat mul(const Mat &A, const Mat &B) {
    Mat C = zeros();
    for (int i = 0; i < 16; i++)
        for (int k = 0; k < 16; k++) {
            double a = A[i * 16 + k];
            if (a == 0.0) continue;
            for (int j = 0; j < 16; j++) C[i * 16 + j] += a * B[k * 16 + j];
        }
    return C;
}
inline Mat transpose(const Mat &A) {
    Mat T = zeros();
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) T[j * 16 + i] = A[i * 16 + j];
    return T;
}
inline Mat neg(Mat A) {
    for (auto &x : A) x = -x;
    return A;
}
inline Mat scale_cols(Mat A, const std::vector<double> &d) {  // A * diag(d)
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) A[i * 16 + j] *= d[j];
    return A;
}
inline Mat scale_rows(Mat A, const std::vector<double> &d) {  // diag(d) * A
    for (int i = 0; i < 16; i++)
        for (int j = 0; j < 16; j++) A[i * 16 + j] *= d[i];
    return A;
}
// inverse of an n x n upper-triangular matrix held in the top-left corner of U (true diagonal)
inline Mat inv_upper(const Mat &U, int n) {
    Mat X = zeros();
    for (int j = 0; j < n; j++) {
        X[j * 16 + j] = 1.0 / U[j * 16 + j];
        for (int i = j - 1; i >= 0; i--) {
            double s = 0.0;
            for (int k = i + 1; k <= j; k++) s += U[i * 16 + k] * X[k * 16 + j];
            X[i * 16 + j] = -s / U[i * 16 + i];
        }
    }
    return X;
}
}  // namespace hostla

inline void mfma_plan_free(MfmaPlan &p) {
    if (p.d_table) hipFree(p.d_table);
    p.d_table = nullptr;
}

inline bool mfma_shape_instantiated(int N, int KX, int KS);

inline int mfma_plan_build(MfmaPlan &p, const AdmmHost &a) {
    using namespace hostla;
    const int n = a.n, m = a.m, N = a.N, nm = n + m;
    p.ok = false;
    p.n = n;
    p.m = m;
    if (nm > 16) { p.why = "n+m > 16 needs the multi-tile kernel (not built yet)"; return 0; }
    MfmaLayout L{N, (n + 3) / 4, (nm + 3) / 4, n / 4, a.terminal};
    if (L.KU0 != L.KS - 1) { p.why = "u rows must sit inside the last k-step (n % 4 + m <= 4)"; return 0; }
    if (!mfma_shape_instantiated(N, L.KX, L.KS)) {
        p.why = "MFMA kernel not instantiated for this (N, n, m)";
        return 0;
    }
    // the scalar-rho, time-invariant structure the shared Zmid tile relies on
    for (int l = 1; l < N - 1; l++)
        for (int j = 0; j < nm; j++)
            if (a.Hi[(size_t)l * nm + j] != a.Hi[j]) { p.why = "Hi differs between stages (vector rho?)"; return 0; }
    p.lay = L;
    std::vector<double> tab((size_t)L.total_doubles(), 0.0);
    auto put = [&](int tile0, const Mat &M, int s_begin, int s_end) {
        for (int s = s_begin; s < s_end; s++) {
            double *t = tab.data() + (size_t)(tile0 + (s - s_begin)) * 64;
            for (int lane = 0; lane < 64; lane++) t[lane] = M[(lane & 15) * 16 + 4 * s + (lane >> 4)];
        }
    };
    // padded ingredient matrices
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
    std::vector<Mat> Bi(N), Al(N > 1 ? N - 1 : 0);
    for (int l = 0; l < N; l++) {
        Mat U = zeros();
        for (int i = 0; i < n; i++)
            for (int j = i; j < n; j++) {
                double v = a.Beta[((size_t)l * n + i) * n + j];
                U[i * 16 + j] = (i == j) ? 1.0 / v : v;  // Beta stores the inverted diagonal
            }
        Bi[l] = inv_upper(U, n);
    }
    for (int l = 0; l < N - 1; l++) {
        Al[l] = zeros();
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) Al[l][i * 16 + j] = a.Alpha[((size_t)l * n + i) * n + j];
    }
    for (int l = 0; l < N; l++) {
        Mat BiT = transpose(Bi[l]);
        // F2_l = -Bi_l' AB diag(Hd_l)
        Mat F2 = neg(mul(BiT, scale_cols(AB, l == 0 ? hd_0 : hd_mid)));
        put(L.f2(l), F2, l == 0 ? L.KU0 : 0, L.KS);
        // F1_l = Bi_l' Dx_{l+1}   (dense Hi_N for the terminal stage)
        if (L.f1_cnt(l)) {
            Mat Dx = zeros();
            if (l + 1 == N) Dx = HiN;
            else for (int j = 0; j < n; j++) Dx[j * 16 + j] = hdx_mid[j];
            put(L.f1(l), mul(BiT, Dx), 0, L.KX);
        }
        if (l >= 1) put(L.f3(l), neg(mul(BiT, transpose(Al[l - 1]))), 0, L.KX);
        put(L.b1(l), Bi[l], 0, L.KX);
        if (l < N - 1) put(L.b2(l), neg(mul(Bi[l], Al[l])), 0, L.KX);
    }
    Mat ABt = transpose(AB);
    put(L.z0(), neg(scale_rows(ABt, hd_0)), 0, L.KX);
    put(L.zmid(), neg(scale_rows(ABt, hd_mid)), 0, L.KX);
    put(L.zN(), neg(HiN), 0, L.KX);
    put(L.s0(), mul(transpose(Bi[0]), A), 0, L.KX);
    put(L.sT(), Tm, 0, L.KX);
    put(L.sN(), neg(transpose(Bi[N - 1])), 0, L.KX);
    // row constants
    auto rc = [&](int i) { return tab.data() + L.rc_off(i); };
    for (int j = 0; j < 16; j++) {
        rc(MfmaLayout::RC_NEGHD_MID)[j] = -hd_mid[j];
        rc(MfmaLayout::RC_NEGHD_0)[j] = -hd_0[j];
    }
    for (int j = 0; j < nm; j++) {
        rc(MfmaLayout::RC_LB_MID)[j] = a.LB[j];
        rc(MfmaLayout::RC_UB_MID)[j] = a.UB[j];
    }
    for (int j = 0; j < m; j++) {
        rc(MfmaLayout::RC_LB_0)[n + j] = a.LB[n + j];
        rc(MfmaLayout::RC_UB_0)[n + j] = a.UB[n + j];
    }
    for (int j = 0; j < n; j++) {
        rc(MfmaLayout::RC_LB_N)[j] = a.LB[j];
        rc(MfmaLayout::RC_UB_N)[j] = a.UB[j];
        rc(MfmaLayout::RC_QR)[j] = a.Q[j];
    }
    for (int j = 0; j < m; j++) rc(MfmaLayout::RC_QR)[n + j] = a.R[j];
    for (double x : tab)
        if (!std::isfinite(x)) { p.why = "non-finite folded constant (singular Beta block?)"; return 0; }
    p.table_bytes = tab.size() * sizeof(double);
    if (p.table_bytes > 160 * 1024 - 512) { p.why = "tile table exceeds the 160 KB LDS"; return 0; }
    SPCIES_HIP_CHECK(hipMalloc((void **)&p.d_table, p.table_bytes + 64));  // + a dump word for masked-off stores
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

// ---------------------------------------------------------------------------------------------
// Device
// ---------------------------------------------------------------------------------------------
// [rtc-begin]
struct MfmaArgs {
    int n, m, k_max;
    double tol, rho, rho_i;
    long B;
    int ref_stride;
};

#define LAUNDER4(x) asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]))
// [rtc-end]
#define SPCIES_MFMA(acc, a, b) acc = __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (acc), 0, 0, 0)

template <int N, int KX, int KS, bool TERMINAL, bool WANT_SOL>
__global__ __launch_bounds__(256, 1) void admm_mfma_kernel(MfmaArgs p, const double *__restrict__ table_g,
                                                           const double *__restrict__ x0g,
                                                           const double *__restrict__ xrg,
                                                           const double *__restrict__ urg, double *__restrict__ u_out,
                                                           int *__restrict__ k_out, int *__restrict__ e_out,
                                                           double *__restrict__ z_out, double *__restrict__ v_out,
                                                           double *__restrict__ lam_out, double *__restrict__ dump) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
    const int n = p.n, m = p.m, nm = n + m;
    // Every shape instantiated in this library keeps the u rows inside the last k-step (KU0 == KS-1;
    // mfma_plan_build refuses anything else), so the whole tile layout is a compile-time constant.
    constexpr MfmaLayout LL{N, KX, KS, KS - 1, TERMINAL};
    // ---- stage the whole table into LDS once per workgroup (16-byte loads)
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

    // LDS reads of the tiles are loop-invariant; left alone, LICM hoists all of them out of the
    // iteration loop and spills.  `lo` / `go` are laundered through an empty asm once per iteration
    // so the addresses look iteration-dependent: reads stay inside the loop, free to be scheduled.
    int lo = lane, go = g;
    auto A = [&](int tile) -> double { return lds[tile * 64 + lo]; };
    auto RC = [&](int i) -> d4 {
        const double *r = lds + LL.rc_off(i);
        return d4{r[go], r[4 + go], r[8 + go], r[12 + go]};
    };

    for (long tile = (long)blockIdx.x * 4 + wave; tile < n_tiles; tile += (long)gridDim.x * 4) {
        const long inst = tile * 16 + c;
        const bool valid = inst < p.B;
        // ---- per-instance setup (code_laxMPC_ADMM_C.c:282-299)
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
        const d4 qm = RC(MfmaLayout::RC_QR) * xuv;  // [Q o xr; R o ur]  (negated weights)
        d4 c0 = {0, 0, 0, 0}, qT = {0, 0, 0, 0}, cN = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < KX; s++) {
            SPCIES_MFMA(c0, A(LL.s0() + s), x0v[s]);
            if constexpr (TERMINAL) SPCIES_MFMA(qT, A(LL.sT() + s), xrv[s]);
            else SPCIES_MFMA(cN, A(LL.sN() + s), xrv[s]);
        }

        // State kept between iterations: ONE vector per stage, w = z + lambda/rho (the argument of the
        // box projection).  v and lambda are functions of it - v = clamp(w), lambda = rho (w - v), which is
        // the reference's  lambda + rho (z - v)  (code_laxMPC_ADMM_C.c:542-568) written out - so they are
        // rebuilt on the fly instead of occupying 2 x 128 registers.  The cold start v = lambda = 0
        // (:58-66) is not representable as a w when 0 lies outside the box, hence the `first` flag.
        d4 w[N + 1], mu[N];
#pragma unroll
        for (int t = 0; t <= N; t++) w[t] = d4{0, 0, 0, 0};
        bool active = valid;
        int kk = 0;

        auto clampv = [&](const d4 &x, const d4 &lb, const d4 &ub) -> d4 {
            d4 r;
#pragma unroll
            for (int i = 0; i < 4; i++) r[i] = fmin(fmax(x[i], lb[i]), ub[i]);  // == the reference's two ?: for non-NaN
            return r;
        };
        // resident: row constants of the middle stages and the shared Zmid tiles
        d4 lbm = RC(MfmaLayout::RC_LB_MID), ubm = RC(MfmaLayout::RC_UB_MID), nhm = RC(MfmaLayout::RC_NEGHD_MID);
        
        double zm[KX];
#pragma unroll
        for (int s = 0; s < KX; s++) zm[s] = A(LL.zmid() + s);

        // Tile prefetch: the first PF A-operands of segment j+1 are read from LDS while segment j computes
        // (a segment's other tiles are read at its start and land behind its first MFMAs).
        // Segments: N forward blocks, N backward blocks (+ the terminal-stage tiles), one final (stage 0).
        constexpr int PF = 2;
        double pfc[PF], pfn[PF];
#pragma unroll
        for (int i = 0; i < PF; i++) pfc[i] = A(LL.fwd_off(0) + i);

        while (true) {
            kk += 1;
            // cold start: v = lambda = 0 in iteration 1 whatever clamp(0) is
            const double fz = (kk == 1) ? 0.0 : 1.0, rf = rho * fz;
            asm volatile("" : "+v"(lo), "+v"(go));
            long il = inst;  // laundered too: otherwise every output address is hoisted out of the loop and spills
            asm volatile("" : "+v"(il));
            double *zp = WANT_SOL ? z_out + il * dim + g : nullptr;  // row 4r+g of stage t lives at zp[off_t + 4r]
            // (the empty asm keeps the compiler from proving that the backward sweep recomputes what the forward
            //  sweep already had - it would otherwise keep 14 stages of clamp(w) / q_hat alive and spill)
            LAUNDER4(lbm); LAUNDER4(ubm);
            auto LBt = [&](int t) -> d4 { return t == 0 ? RC(MfmaLayout::RC_LB_0) : (t == N ? RC(MfmaLayout::RC_LB_N) : lbm); };
            auto UBt = [&](int t) -> d4 { return t == 0 ? RC(MfmaLayout::RC_UB_0) : (t == N ? RC(MfmaLayout::RC_UB_N) : ubm); };
            // q_hat_t = q_t + lambda_t - rho v_t = q_t + rho fz (w_t - 2 clamp(w_t))
            auto qhat = [&](int t, d4 &cw) -> d4 {
#if defined(SPCIES_ABLATE) && (SPCIES_ABLATE & 1)  // timing-only diagnostic build: no elementwise work
                cw = w[t];
                return w[t];
#else
                cw = clampv(w[t], LBt(t), UBt(t));
                return ((t == N) ? qT : qm) + rf * (w[t] - 2.0 * cw);
#endif
            };
            d4 cw;
            // Both sweeps are software-pipelined by one segment: the elementwise work a segment issues never
            // depends on that segment's own MFMAs, so VALU and matrix pipe overlap instead of alternating.
            // ============ forward sweep ============
            // carried: qh = q_hat_l, qn = q_hat_{l+1} (both ready before block l starts)
            d4 qh = qhat(0, cw);
            d4 qn = qhat(1, cw);
#pragma unroll
            for (int l = 0; l < N; l++) {
                const int nxt = (l + 1 < N) ? LL.fwd_off(l + 1) : LL.b1(N - 1);
#pragma unroll
                for (int i = 0; i < PF; i++) pfn[i] = A(nxt + i);
                auto T = [&](int i) -> double { return i < PF ? pfc[i] : A(LL.fwd_off(l) + i); };
                d4 acc = (l == 0) ? c0 : d4{0, 0, 0, 0};
                if constexpr (!TERMINAL) {
                    if (l == N - 1) acc = cN;
                }
                int ti = 0;
                // F2_l qh_l
                if (l == 0) {
                    SPCIES_MFMA(acc, T(ti), qh[KS - 1]);
                    ti++;
                } else {
#pragma unroll
                    for (int s = 0; s < KS; s++) {
                        SPCIES_MFMA(acc, T(ti), qh[s]);
                        ti++;
                    }
                }
                // q_hat_{l+2} for the next block: independent of everything in flight
                d4 qnn = qn;
                if (l + 2 < N || (l + 2 == N && TERMINAL)) qnn = qhat(l + 2, cw);
                // F1_l qh_{l+1}
                if (TERMINAL || l < N - 1) {
#pragma unroll
                    for (int s = 0; s < KX; s++) {
                        SPCIES_MFMA(acc, T(ti), qn[s]);
                        ti++;
                    }
                }
                // F3_l y_{l-1}
                if (l >= 1) {
#pragma unroll
                    for (int s = 0; s < KX; s++) {
                        SPCIES_MFMA(acc, T(ti), mu[l - 1][s]);
                        ti++;
                    }
                }
                mu[l] = acc;
                qh = qn;
                qn = qnn;
                __builtin_amdgcn_sched_barrier(0);  // a segment's tile reads must not drift into earlier segments
#pragma unroll
                for (int i = 0; i < PF; i++) pfc[i] = pfn[i];
            }
            // ============ backward sweep, primal update, projection, dual update ============
            asm volatile("" : "+v"(go));  // re-read (not keep alive) the stage-0 / stage-N row constants
            LAUNDER4(lbm); LAUNDER4(ubm);
            bool res = false;
            // stage t, first half: z_t = Z mu_t - Hd o (qh_t - [mu_{t-1}; 0])   (MFMA + the VALU that feeds it)
            auto stage_z = [&](int t, double z0t, double z1t, double z2t, double z3t, d4 &cwt) -> d4 {
                const double zt[4] = {z0t, z1t, z2t, z3t};
                d4 z;
                const d4 qq = qhat(t, cwt);
                if (t == N) {  // terminal stage, dense Hi_N: z_N = -Hi_N (qh_N - mu_{N-1})
                    const d4 wv = qq - mu[N - 1];
                    z = d4{0, 0, 0, 0};
#pragma unroll
                    for (int s = 0; s < KX; s++) SPCIES_MFMA(z, zt[s], wv[s]);
                } else if (t == 0) {
                    z = RC(MfmaLayout::RC_NEGHD_0) * qq;
#pragma unroll
                    for (int s = 0; s < KX; s++) SPCIES_MFMA(z, zt[s], mu[0][s]);
                } else {
                    z = nhm * (qq - mu[t - 1]);
#pragma unroll
                    for (int s = 0; s < KX; s++) SPCIES_MFMA(z, zt[s], mu[t][s]);
                }
                return z;
            };
            // stage t, second half (issued one segment later): w_t <- z_t + lambda_t / rho, residuals
            auto stage_w = [&](int t, const d4 &z, const d4 &cwt) {
#if defined(SPCIES_ABLATE) && (SPCIES_ABLATE & 1)
                const d4 wn = z * 1e-3;
                res |= (wn[0] < 1e300);
#else
                const d4 wn = z + fz * (w[t] - cwt);  // z + lambda/rho
                const d4 vn = clampv(wn, LBt(t), UBt(t));
                const d4 vo = fz * cwt;
#pragma unroll
                for (int r = 0; r < 4; r++) res |= (fabs(vo[r] - vn[r]) > tol) | (fabs(z[r] - vn[r]) > tol);  // no short-circuit: no branches
#endif
                w[t] = wn;  // finished instances keep iterating harmlessly: their results are already stored
                if constexpr (WANT_SOL) {
                    // Branch-free: lanes that must not write (finished instance, padding row) aim at a dump
                    // word instead - an `if` per stage here splits the iteration into dozens of blocks and
                    // the register allocator then spills hundreds of values.
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
            auto zmid = [&](int t, d4 &cwt) -> d4 {
                return stage_z(t, zm[0], KX > 1 ? zm[KX > 1 ? 1 : 0] : 0.0, KX > 2 ? zm[KX > 2 ? 2 : 0] : 0.0,
                               KX > 3 ? zm[KX > 3 ? 3 : 0] : 0.0, cwt);
            };
            // segment for block l:  (1) finish stage l+3 (VALU only, z from the previous segment)
            //                       (2) B1_l y_l   (3) z of stage l+2   (4) B2_l mu_{l+1}
            d4 zc = {0, 0, 0, 0}, cwc = {0, 0, 0, 0};  // z and clamp(w) of the stage awaiting its second half
#pragma unroll
            for (int l = N - 1; l >= 0; l--) {
                // this segment's tiles: b1(l) [KX], b2(l) [KX, l < N-1], then zN [KX] when its stage is t = N
                const int nb = (l < N - 1) ? 2 * KX : KX;
                auto T = [&](int i) -> double {
                    return i < PF ? pfc[i] : (i < nb ? A(LL.b1(l) + i) : A(LL.zN() + (i - nb)));
                };
                const int nxt = (l >= 1) ? LL.b1(l - 1) : LL.z0();
#pragma unroll
                for (int i = 0; i < PF; i++) pfn[i] = A(nxt + i);
                const int tp = l + 3;  // stage whose z was produced by the previous segment
                if (tp < N || (tp == N && TERMINAL)) stage_w(tp, zc, cwc);
                // one accumulator chain (B1 then B2, starting from C = 0), then the Z chain: every switch of
                // accumulator costs the matrix pipe ~50 cycles (profiles/r01_microbench_f64_v2.txt)
                d4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int s = 0; s < KX; s++) SPCIES_MFMA(acc, T(s), mu[l][s]);
                if (l < N - 1) {
#pragma unroll
                    for (int s = 0; s < KX; s++) SPCIES_MFMA(acc, T(KX + s), mu[l + 1][s]);
                }
                const int t = l + 2;
                if (t < N) zc = zmid(t, cwc);
                else if (t == N && TERMINAL)
                    zc = stage_z(N, T(nb), KX > 1 ? T(nb + 1) : 0.0, KX > 2 ? T(nb + 2) : 0.0, KX > 3 ? T(nb + 3) : 0.0, cwc);
                mu[l] = acc;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < PF; i++) pfc[i] = pfn[i];
            }
            // tail: stages 2 (second half), 1, 0; meanwhile fetch the head of forward block 0 for the next iteration
            {
#pragma unroll
                for (int i = 0; i < PF; i++) pfn[i] = A(LL.fwd_off(0) + i);
                asm volatile("" : "+v"(go));
                auto T = [&](int i) -> double { return i < PF ? pfc[i] : A(LL.z0() + i); };
                d4 cw1, cw0;
                const d4 z1 = zmid(1, cw1);
                const d4 z0 = stage_z(0, T(0), KX > 1 ? T(1) : 0.0, KX > 2 ? T(2) : 0.0, KX > 3 ? T(3) : 0.0, cw0);
                stage_w(2, zc, cwc);
                stage_w(1, z1, cw1);
                stage_w(0, z0, cw0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < PF; i++) pfc[i] = pfn[i];
            }
            // ============ exit test per instance (code_laxMPC_ADMM_C.c:572-631) ============
            unsigned long long bal = __ballot(res);
            bal |= bal >> 32;
            bal |= bal >> 16;
            const bool res_inst = (bal >> c) & 1ull;
            const bool done_now = active && (!res_inst || kk >= p.k_max);
            if (__any(done_now)) {
                // ---- results of the instances that stop at this iteration (code_laxMPC_ADMM_C.c:642-686)
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
}

// shapes compiled into this library: (N, KX, KS)
#define SPCIES_MFMA_SHAPES(X) X(10, 2, 2) X(15, 3, 4)

inline bool mfma_shape_instantiated(int N, int KX, int KS) {
#define X(NN, KKX, KKS) \
    if (N == NN && KX == KKX && KS == KKS) return true;
    SPCIES_MFMA_SHAPES(X)
#undef X
    return false;
}

#ifndef SPCIES_NO_BUILTIN_LAUNCHERS  // (admm_mfma4u.hip includes these headers for the host-side declarations only: no second copy of the kernels there)
template <int N, int KX, int KS>
static int launch_mfma_shape(MfmaPlan &pl, const AdmmHost &a, const MfmaArgs &args, const double *x0, const double *xr,
                             const double *ur, double *u, int *k, int *e, double *z, double *v, double *lam,
                             hipStream_t st) {
    const bool want_sol = (z || v || lam);
    if (want_sol && !(z && v && lam)) return fail(SPCIES_HIP_EINVAL, "MFMA variant: pass all of z, v, lambda or none");
    const long n_tiles = (args.B + 15) / 16;
    long wgs = (n_tiles + 3) / 4;
    if (wgs > pl.num_cu) wgs = pl.num_cu;  // one 160KB-LDS workgroup per CU; waves loop over tiles
    const size_t shmem = pl.table_bytes;
    dim3 grid((unsigned)wgs), block(256);
#define SPCIES_LAUNCH(TERM, SOL)                                                                                     \
    do {                                                                                                             \
        auto kern = admm_mfma_kernel<N, KX, KS, TERM, SOL>;                                                          \
        /* per device, so set before every launch (a handle may live on any GPU of the process) */                   \
        SPCIES_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                                 160 * 1024));                                                      \
        hipLaunchKernelGGL(kern, grid, block, shmem, st, args, pl.d_table, x0, xr, ur, u, k, e, z, v, lam,           \
                           pl.d_table + pl.table_bytes / sizeof(double));          \
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

inline int launch_mfma(MfmaPlan &pl, const AdmmHost &a, const double *x0, const double *xr, const double *ur,
                       int ref_stride, long B, double *u, int *k, int *e, double *z, double *v, double *lam,
                       hipStream_t st) {
    if (!pl.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA variant unavailable: %s", pl.why.c_str());
    MfmaArgs args{a.n, a.m, a.k_max, a.tol, a.rho, a.rho_i, B, ref_stride};
#define X(NN, KKX, KKS)                                                \
    if (pl.lay.N == NN && pl.lay.KX == KKX && pl.lay.KS == KKS)        \
        return launch_mfma_shape<NN, KKX, KKS>(pl, a, args, x0, xr, ur, u, k, e, z, v, lam, st);
    SPCIES_MFMA_SHAPES(X)
#undef X
    return fail(SPCIES_HIP_ENOSUP, "MFMA kernel not instantiated for N=%d KX=%d KS=%d", pl.lay.N, pl.lay.KX, pl.lay.KS);
}

#endif  // SPCIES_NO_BUILTIN_LAUNCHERS

}  // namespace spcies
