// Variant FUSED of the MPCT ADMM solver on the extended state space ('cs'; code_MPCT_ADMM_cs_C.c:99-217): the whole
// iteration - q_hat, the CSR product, the L D L' solve of W, the two CSR products, clamp, dual step, residuals - as ONE
// dense contraction with the state in registers (cs_fused_kernel.inc): z = ME (w - 2 clamp(w)) + c in w-form, ME built
// on the host by pushing unit vectors through the reference's own sparse operator.  1e-10 against the oracle (sums in
// another order); controllers whose W is too ill-conditioned for that (checked at build time by a residual test of ME
// against the sparse operator) stay on TILE.
#pragma once
#include "common.hpp"

namespace spcies {
namespace csfused {

struct Host {  // what parse_mpct_cs collected (cons_MPCT_ADMM_cs_C.m:66-112)
    int n, m, N, dim, nrow, scalar_rho;
    double rho;
    const double *rho_v;                                   // [dim] (vector rho) or NULL
    const double *Tz, *Sz, *LB, *UB;                       // [n][n], [m][m], [dim], [dim]
    const double *L_val, *Dinv, *AHi_val, *HiA_val, *Hi_val;
    const int *L_col, *L_row, *AHi_col, *AHi_row, *HiA_col, *HiA_row, *Hi_col, *Hi_row;
};

struct Plan {
    bool ok = false;
    std::string why = "not built";
    bool build_failed = false;  // the variant applies to this controller but its run-time specialisation failed (hiprtc missing, compile error): what SPCIES_HIP_STRICT reacts to
    int n = 0, m = 0, N = 0, NR = 0, NCH = 0, CHB = 0;
    double *d_ME = nullptr, *d_PRO = nullptr, *d_C = nullptr;
    int oLB = 0, oUB = 0, oRho = 0;
    int num_cu = 256;
    void *module = nullptr;            // hipModule_t of a run-time specialised kernel
    void *fn[2] = {nullptr, nullptr};  // WANT_SOL = false, true
    int builtin = -1;
};

int plan_build(Plan &p, const Host &h);
void plan_free(Plan &p);
// u, k, e, z, v, lam: device pointers (z, v, lam may be NULL)
int launch(Plan &p, int k_max, double tol, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
           int *k, int *e, double *z, double *v, double *lam, hipStream_t st);

}  // namespace csfused
}  // namespace spcies
