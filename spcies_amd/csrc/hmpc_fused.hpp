// Variant FUSED of the HMPC solvers: the dense contraction of an iteration (the reference's NON_SPARSE path,
// code_HMPC_ADMM_split_C.c:174-190; the non-split solver's z-update, code_HMPC_ADMM_C.c:123-157) AND everything
// between two contractions (box / proj_SOC3 projections, dual steps, residual flags, exit test) in ONE hand-written
// kernel on v_mfma_f64_4x4x4: four instances per wavefront, sixteen rows per register, all state and all accumulators
// in registers, the controller's matrix streamed L2 -> LDS by LDS-DMA (hmpc_fused_kernel.inc).  No library GEMM, no
// state in HBM.  Host side: the table in the kernel's issue order, build-time instantiations for the benchmark
// shapes, hiprtc specialisation for any other controller (Spcies prints one solver per controller; so does this).
#pragma once
#include "common.hpp"

namespace spcies {
namespace hfused {

struct SplitHost {  // what parse_hmpc collected (HMPC ADMM / SADMM split, blob arrays of cons_HMPC_ADMM_split_C.m:121-150)
    int n, m, N, dim, n_s, n_eq, n_soc, use_soc, symmetric, k_max;
    int coupled, n_y;                                          // COUPLED_CONSTRAINTS: n_y outputs, s = [N n_y box slacks ; cones]
    double tol_p, tol_d, rho, rho_i, sigma, sigma_i, alpha;
    const double *M1, *M2, *bh_nat;                            // [np][np], [np][n_eq + n_s], [n_eq + n_s]
    const double *A, *QQ, *Te, *Se, *LB, *UB, *LBy, *UBy;     // dense small matrices and bounds
};

struct NosplitHost {  // what parse_hmpc_dense collected (HMPC ADMM / SADMM without the splitting, cons_HMPC_ADMM_C.m:88-131)
    int n, m, N, dim, n_s, n_box, n_soc, use_soc, symmetric, k_max;
    double tol_p, tol_d, rho, rho_i, alpha;
    const double *M1, *M2;                                     // [dim][dim], [dim][n]
    const double *A, *QQ, *Te, *Se, *LB, *UB, *LBy, *UBy, *d;  // d: [n_s] or NULL (diamond mode)
    const double *C_val;                                       // CSR of C [n_s x dim]
    const int *C_row, *C_col;
};

struct Plan {
    bool ok = false;
    std::string why = "not built";
    bool build_failed = false;  // the variant applies to this controller but its run-time specialisation failed (hiprtc missing, compile error): what SPCIES_HIP_STRICT reacts to
    int n = 0, m = 0, N = 0, use_soc = 0, symmetric = 0, mode = 0, ny = 0;  // mode 0: split, 1: no splitting; ny > 0: coupled constraints
    int NR = 0, NK = 0, NCH = 0, CHB = 0;
    double *d_ME = nullptr, *d_PRO = nullptr, *d_C = nullptr;  // iteration table, prologue table (inputs | 1), constants
    int oQQ = 0, oTe = 0, oSe = 0, oLB = 0, oUB = 0, oD1 = 0, oD2 = 0, oZcol = 0, oZcoef = 0, oZd = 0;
    int num_cu = 256;
    void *module = nullptr;        // hipModule_t of a run-time specialised kernel (shapes not instantiated at build time)
    void *fn[2] = {nullptr, nullptr};  // WANT_SOL = false, true
    int builtin = -1;              // index into the build-time instantiations, or -1
    // HMPC without the splitting and WITH coupled constraints: the z record.  d_ZR [dim][nin + n_s + 1] = (M2 b + M1 q as a map of
    // [x0; xr; ur] | M1 C' | -rho M1 C' d), d_T [cap_T][n_s] the operand rho s + lambda of every instance's last product
    double *d_ZR = nullptr, *d_T = nullptr;
    long cap_T = 0;
    int z_dim = 0, z_ns = 0;
    double z_rho = 0;              // (the rho the constant column was folded with)
};

int plan_build_split(Plan &p, const SplitHost &h);
int plan_build_nosplit(Plan &p, const NosplitHost &h);
void plan_free(Plan &p);
// u, k, e, fields (split: z, s, z_hat, s_hat, lambda, mu; no splitting: z, s, lambda; NULL entries are skipped) are device
// pointers; k_max / tolerances as given (sigma is not used without the splitting)
int launch(Plan &p, int k_max, double tol_p, double tol_d, double rho, double rho_i, double sigma, double sigma_i, double alpha,
                 const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u, int *k, int *e,
                 double *const *f, hipStream_t st);

}  // namespace hfused
}  // namespace spcies
