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
#include "common.hpp"

namespace spcies {
namespace tvr {

struct Args {
    int k_max, ref_stride;
    double rho, tol;
    long B, Bp;
    long RA;  // 0: the scratch is [row][Bp]; > 0: instance-major, RA rows per instance (tv_update_coop_kernel, FORM 1)
};
#define SPCIES_TVR_ARGS_DEFINED 1

// shapes instantiated at build time: (n, m, N); any other horizon of these (n, m) is specialised with hiprtc at create time
#define SPCIES_TVR_SHAPES(X) X(12, 2, 15) X(6, 2, 10)
inline bool shape_built(int n, int m, int N) {
#define X(nn, mm, NN) \
    if (n == nn && m == mm && N == NN) return true;
    SPCIES_TVR_SHAPES(X)
#undef X
    return false;
}

struct Plan {
    bool ok = false;
    std::string why = "not built";
    bool build_failed = false;  // the variant applies to this controller but its run-time specialisation failed: what SPCIES_HIP_STRICT reacts to
    int n = 0, m = 0, N = 0;
    bool terminal = true, builtin = false, fista = false;
    void *module = nullptr;            // hipModule_t of a run-time specialised kernel
    void *fn[3] = {nullptr, nullptr, nullptr};  // (unused), iteration without / with the record
    void *fn_ms = nullptr;                      // (run-time specialised path) tv_ms_kernel<n>: the update phase's rows -> the L D L' form's M, S
    void *fn_update = nullptr;                  // the update-phase kernel of an (n, m) without a build-time instantiation (tv_update_kernel.inc)
    bool update_builtin = false;                // (n, m) = (6, 2), (12, 2): admm_stream.hpp's instantiations, launched by the caller
    // plants past the register file (n + m > 16, or a horizon whose factors the registers do not hold): the same iteration with the instance's
    // factors in the LDS - admm_tvl_kernel.inc (ADMM and FISTA), n + m <= 32, N n^2 + (N - 1) n^2 + n (n + m) (+ n^2) doubles within 160 KB; always run-time
    // specialised, with the update phase (rolled past n = 16) and the explicit inverses (tv_bi_rolled_kernel) of the same module
    bool lds = false;
    int lds_per_cu = 1;                         // workgroups (= instances) a CU's LDS holds at once
    void *fn_bi = nullptr;
    void *fn_coop = nullptr;                    // tv_update_coop_kernel: the LDS form's update phase, several lanes per instance (the product path)
    bool coop = true;                           // (SPCIES_TVL_COOP=0: fn_update + fn_bi instead - one lane per instance, the same bits)
    // the register-resident form (not lds): coop_im = the cooperative update phase in its instance-major form (FORM 1: S_l / M_l rows, no L D L' transform
    // kernel); SPCIES_TVR_COOP=0: the one-lane update phase + tv_ms_kernel, [row][Bp] scratch
    bool coop_im = false;
    long rows_all = 0;                          // rows per instance of the scratch (Bi + N n^2)
};
// decides whether the variant applies (n + m <= 16 and the state within the register file; past it the LDS form, see Plan::lds) and, for a shape without build-time kernels, compiles
// them (hiprtc; code-object cache) - the update phase included, so that ANY plant size within those limits has a time-varying path
int plan_build(Plan &p, int n, int m, int N, bool terminal, bool fista = false);
// (admm_tvl.hip) the LDS form: the text of admm_tvl_kernel.inc for hiprtc, and the bytes of LDS one instance's images take
constexpr int coop_instances_per_wavefront(int n) { return 64 / (n <= 8 ? 8 : (n <= 16 ? 16 : (n <= 21 ? 21 : 32))); }  // (tvl_coop_lpi of admm_tvl_kernel.inc)
const char *tvl_source();
long tvl_lds_bytes(int n, int m, int N, bool terminal, bool fista);
int launch_coop_builtin(int n, int m, int N, bool terminal, bool fista, double c0, const double *Tc, const double *model, long model_stride, long B, long Bp,
                        double *TVS, hipStream_t st);
void plan_free(Plan &p);
// the solve of one chunk; pointers are device memory, TVS as admm_tv_update_kernel<n, m, TERMINAL, BI = true> left it (factors AND the explicit
// inverses Bi: tv_update_kernel.inc)
int launch(const Plan &p, bool want_sol, const Args &a, const double *TRI, const double *T, double *TVS, const double *x0, const double *xr,
           const double *ur, double *u, int *k, int *e, double *z, double *v, double *lam, int num_cu, hipStream_t st);
// the FISTA twin (code_laxMPC_FISTA_C.c, TIME_VARYING == 1): T, Ti = the negated diagonal terminal weight and -1 / it (controller constants);
// TVS as fista_tv_update_kernel left it; record z, lambda (= y)
int launch_fista(const Plan &p, bool want_sol, const Args &a, const double *T, const double *Ti, double *TVS, const double *x0, const double *xr,
                 const double *ur, double *u, int *k, int *e, double *z, double *lam, int num_cu, hipStream_t st);

// the update phase WITH the explicit inverses (admm_tv_update_kernel<n, m, TERMINAL, BI = true> / fista_tv_update_kernel<...>): the build-time shapes
// (instantiated in admm_tvr.hip) or the run-time specialised kernel of any other (n, m)
// (ADMM: c0 = rho, Tc = T_rho_i; FISTA: Tc = Ti, c0 unused)
int launch_update(const Plan &p, double c0, const double *Tc, const double *model, long model_stride, long B, long Bp, double *TVS, hipStream_t st);

}  // namespace tvr
}  // namespace spcies
