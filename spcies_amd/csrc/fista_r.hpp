// Variant MFMA4R of the banded-Cholesky FISTA solvers (laxMPC / equMPC): unrolled on the horizon, the whole iteration
// state (y, lambda, forward-substituted d) resident in registers + LDS, the controller's 4x4 blocks streamed L2 -> LDS by
// LDS-DMA (fista_r_kernel.inc has the design).  The kernel is specialised per controller: hiprtc at create time (Spcies
// prints one solver per controller; so does this), build-time instantiations for the benchmark shapes.
#pragma once
#include "common.hpp"

namespace spcies {
namespace fr {

struct Host {  // what parse_banded collected for a FISTA solver (cons_laxMPC_FISTA_C.m / cons_equMPC_FISTA_C.m arrays)
    int n, m, N, k_max;
    bool terminal;
    double tol;
    const double *AB, *Alpha, *Beta, *Q, *R, *QRi, *Td, *Ti, *LB, *UB;  // Td, Ti: terminal weight diagonal / its inverse (lax)
};

struct Plan {
    bool ok = false;
    std::string why = "not built";
    bool build_failed = false;  // the variant applies to this controller but its run-time specialisation failed (hiprtc missing, compile error): what SPCIES_HIP_STRICT reacts to
    int n = 0, m = 0, N = 0, KX = 0, KS = 0, NW = 0, NLDS = 0;
    bool terminal = false;
    double *d_table = nullptr;  // + a dump word for masked-off stores
    double *d_scr = nullptr;    // per-wavefront scratch slots of the forward-substituted d
    int PD = 3;
    size_t table_bytes = 0;
    int num_cu = 256;
    void *module = nullptr;            // hipModule_t of the run-time specialised kernel
    void *fn[2] = {nullptr, nullptr};  // WANT_SOL = false, true
    int builtin = -1;                  // index into the build-time instantiations, or -1
};

int plan_build(Plan &p, const Host &h);
void plan_free(Plan &p);
// device pointers; z, lam may be NULL (then neither is written)
int launch(Plan &p, int k_max, double tol, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
           int *k, int *e, double *z, double *lam, hipStream_t st);

}  // namespace fr
}  // namespace spcies
