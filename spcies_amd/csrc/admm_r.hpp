// Variant MFMA4R of the banded-Cholesky ADMM solvers (laxMPC / equMPC, scalar rho, constant bounds) for the shapes admm_mfma4.hpp cannot
// hold: (N + 1) ceil((n + m) / 4) + N ceil(n / 4) > 112 slab registers, n + m > 16, or a block table beyond the LDS.  Unrolled on the
// horizon, w on the chip (registers + LDS), y through an L2-resident scratch slot, the controller's 4x4 blocks streamed L2 -> LDS by
// LDS-DMA (admm_r_kernel.inc has the design); specialised per controller with hiprtc at create time (on-disk code-object cache).
#pragma once
#include "common.hpp"

namespace spcies {
namespace ar {

struct Plan {
    bool ok = false;
    std::string why = "not built";
    bool build_failed = false;  // the variant applies to this controller but its run-time specialisation failed: what SPCIES_HIP_STRICT reacts to
    int n = 0, m = 0, N = 0, KX = 0, KS = 0, NW = 0, NLDS = 0, PD = 3;
    bool terminal = false;
    bool gen = false;   // vector rho / stage-wise bounds (row constants of the middle stages in the chunk stream)
    bool unit = false;  // unit-box coordinates (admm_r_kernel.inc, UNIT): every real row has a finite box around 0
    double rho = 0;
    double *d_table = nullptr;  // + a dump word for masked-off stores
    double *d_scr = nullptr;    // per-wavefront scratch slots of the forward-substituted y
    size_t table_bytes = 0;
    int num_cu = 256;
    void *module = nullptr;            // hipModule_t of the run-time specialised kernels
    void *fn[2] = {nullptr, nullptr};  // WANT_SOL = false, true
};

int plan_build(Plan &p, const AdmmHost &a);
void plan_free(Plan &p);
// device pointers; z, v, lam: all three or none
int launch(Plan &p, int k_max, double tol, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
           int *k, int *e, double *z, double *v, double *lam, hipStream_t st);

}  // namespace ar
}  // namespace spcies
