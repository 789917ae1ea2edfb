// Variant MFMA4R of the MPCT EADMM solver (diagonal Q, R, and - round 4 - general Q, R): unrolled on the horizon, the whole iteration state (z3, lambda,
// y through an L2-resident scratch slot) resident in registers + LDS, eight instances per wavefront in the "H" lane layout,
// the controller's 4x4 blocks streamed L2 -> LDS by LDS-DMA (eadmm_r_kernel.inc has the design).  Specialised per controller:
// hiprtc at create time (Spcies prints one solver per controller; so does this), a build-time instantiation for BASELINE configs[3].
#pragma once
#include "common.hpp"

namespace spcies {
namespace er {

struct Host {  // what parse_banded collected for the MPCT EADMM solver (cons_MPCT_EADMM_C.m arrays)
    int n, m, N, k_max;
    double tol;
    const double *AB, *Alpha, *Beta, *T, *S;  // T, S negated as the reference stores them
    const double *rho, *rho0, *rhos, *LB, *UB, *LB0, *UB0, *LBs, *UBs, *H1i, *W2, *H3i;
    // general Q, R (IS_DIAG == 0, cons_MPCT_EADMM_C.m:102-107): dense inverses of the blocks of H3 and AB times them; H3i unused
    bool general = false;
    const double *Q_bi = nullptr, *Q_mi = nullptr, *R_bi = nullptr, *R_mi = nullptr, *AB_bi = nullptr, *AB_mi = nullptr;
};

struct Plan {
    bool ok = false;
    std::string why = "not built";
    bool build_failed = false;  // the variant applies to this controller but its run-time specialisation failed (hiprtc missing, compile error): what SPCIES_HIP_STRICT reacts to
    int n = 0, m = 0, N = 0, KX = 0, KS = 0, RX = 0, NLS = 0;  // NLS: stages whose z3 / lambda live in LDS
    bool midsame = false, general = false;                    // one table of row constants for the stages 1 .. N - 1; general Q, R
    double *d_table = nullptr;
    double *d_yscr = nullptr;  // per-wavefront scratch slots of the forward-substituted y
    size_t table_bytes = 0;
    int num_cu = 256;
    void *module = nullptr;            // hipModule_t of the run-time specialised kernel
    void *fn[2] = {nullptr, nullptr};  // WANT_SOL = false, true
    int builtin = -1;                  // index into the build-time instantiations, or -1
};

int plan_build(Plan &p, const Host &h);
void plan_free(Plan &p);
// device pointers; z1, z2, z3, lam: all four or none (lam in the reference's packed layout, code_MPCT_EADMM_C.c:495-513)
int launch(Plan &p, int k_max, double tol, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
           int *k, int *e, double *z1, double *z2, double *z3, double *lam, hipStream_t st);

}  // namespace er
}  // namespace spcies
