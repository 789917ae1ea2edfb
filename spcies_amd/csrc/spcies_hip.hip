// libspcies_hip.so - C-ABI implementation (include/spcies_hip.h).  gfx950 only; no CPU fallback.
#include <chrono>
#include <memory>
#include <mutex>

#include "admm_mfma.hpp"
#include "admm_mfma4.hpp"
#include "mfma4_rtc.hpp"
#include "admm_stream.hpp"
#include "admm_tvr.hpp"
#include "fista_stream.hpp"
#include "fista_mfma4g.hpp"
#include "admm_mfma4g.hpp"
#include "eadmm_stream.hpp"
#include "eadmm_mfma4g.hpp"
#include "soc_stream.hpp"
#include "hmpc_stream.hpp"
#include "sparse_tile.hpp"
#include "hmpc_gemm.hpp"
#include "hmpc_dense.hpp"
#include "mpct_cs.hpp"
#include "soc_bsp.hpp"
#include "ellip_bsp.hpp"
#include "hmpc_fused.hpp"
#include "cs_fused.hpp"
#include "fista_r.hpp"
#include "eadmm_r.hpp"
#include "admm_r.hpp"
#include "common.hpp"

// Sources of the run-time specialised STREAM kernel (any plant size: ensure_stream_rtc below)
static const char *const kAdmmDevSrc =
#include "admm_dev_src.inc"
    ;
static const char *const kTvUpdateSrc =
#include "tv_update_src.inc"
    ;
static const char *const kAdmmStreamSrc =
#include "admm_stream_src.inc"
    ;
static const char *const kFistaStreamSrc =
#include "fista_stream_src.inc"
    ;
static const char *const kEadmmStreamSrc =
#include "eadmm_stream_src.inc"
    ;

namespace spcies {

thread_local std::string g_last_error;

struct Solver {
    int device = 0;
    bool device_bound = false;  // hipSetDevice(device) succeeded: device resources may exist
    int formulation = 0, method = 0, submethod = 0;
    int variant = SPCIES_VARIANT_AUTO;
    AdmmHost host;
    // device constants (one allocation) + pointers
    double *d_consts = nullptr;
    AdmmDev dev{};
    FistaDev fdev{};
    std::vector<double> QRi, Td, Ti;  // FISTA-only ingredients
    bool eng = false;                 // in_engineering: arguments scaled on the way in, u on the way out
    std::vector<double> eng_v;        // scaling_x [n] | OpPoint_x [n] | scaling_u [m] | OpPoint_u [m] | scaling_i_u [m]
    double *d_eng = nullptr, *d_eng_in = nullptr;
    size_t eng_in_bytes = 0;
    bool tv = false;                  // time-varying lax/equ ADMM: model arrives with every call (extra)
    int tv_model_size() const { return host.n * host.n + host.n * host.m + host.n + host.m + 2 * (host.n + host.m); }
    SocDev sdev{};
    HmpcDev hdev{};
    std::vector<double> soc_f64;   // ellipMPC-soc: all FP64 constants, concatenated
    std::vector<int> soc_i32;      //               all index arrays, concatenated
    int *d_idx = nullptr;
    EadmmDev edev{};
    std::vector<double> e_rho, e_rho0, e_rhos, e_LB0, e_UB0, e_LBs, e_UBs, e_S, e_H1i, e_W2, e_H3i;  // EADMM-only
    bool e_general = false;                                    // EADMM with general Q, R (IS_DIAG == 0): these instead of e_H3i
    std::vector<double> e_Qbi, e_Qmi, e_Rbi, e_Rmi, e_ABbi, e_ABmi;
    bool is_soc() const { return (formulation == SPCIES_ELLIPMPC && submethod == 1) || is_hmpc(); }  // 6-field (z, s, ...) record
    bool is_hmpc() const { return formulation == SPCIES_HMPC && submethod == 2; }    // split (code_HMPC_ADMM_split_C.c)
    bool is_cs() const { return formulation == SPCIES_MPCT && method == SPCIES_ADMM && submethod == 3; }  // code_MPCT_ADMM_cs_C.c
    bool is_hdense() const { return formulation == SPCIES_HMPC && submethod == 0; }  // no splitting (code_HMPC_ADMM_C.c)
    int soc_dim() const { return is_hmpc() ? hdev.dim : sdev.dim; }
    int soc_ns() const { return is_hmpc() ? hdev.n_s : sdev.n_s; }
    int lam_dim() const {
        if (is_cs()) return cdev.dim;
        if (is_hdense()) return hd_host.n_s;
        if (method == SPCIES_FISTA) return host.N * host.n;
        if (method == SPCIES_EADMM) return (host.N + 3) * (host.n + host.m);
        return host.dim();
    }
    // solution record of the generated solver, in the reference's field order
    //   ADMM  : z, v, lambda            (header_laxMPC_ADMM_C.h:14-22)
    //   FISTA : z, lambda               (header_laxMPC_FISTA_C.h:14-21)
    //   EADMM : z1, z2, z3, lambda      (header_MPCT_EADMM_C.h:14-23)
    //   soc   : z, s, z_hat, s_hat, lambda, mu   (header_ellipMPC_ADMM_soc_C.h:14-24)
    //   HMPC (no splitting): z, s, lambda        (header_HMPC_ADMM_C.h:14-22)
    int n_fields() const { return is_soc() ? 6 : (method == SPCIES_FISTA ? 2 : (method == SPCIES_EADMM ? 4 : 3)); }
    int field_dim(int i) const {
        const int nm = host.n + host.m;
        if (is_soc()) return (i % 2 == 0) ? soc_dim() : soc_ns();
        if (is_hdense()) return i == 0 ? hd_host.dim : hd_host.n_s;
        if (is_cs()) return cdev.dim;
        if (method == SPCIES_FISTA) return i == 0 ? host.dim() : lam_dim();
        if (method == SPCIES_EADMM) return i == 1 ? nm : (i == 3 ? lam_dim() : (host.N + 1) * nm);
        return host.dim();
    }
    const char *field_name(int i) const {
        static const char *admm[] = {"z", "v", "lambda"}, *fista[] = {"z", "lambda"}, *eadmm[] = {"z1", "z2", "z3", "lambda"};
        static const char *soc[] = {"z", "s", "z_hat", "s_hat", "lambda", "mu"};
        static const char *hdn[] = {"z", "s", "lambda"};
        if (is_soc()) return soc[i];
        if (is_hdense()) return hdn[i];
        return method == SPCIES_FISTA ? fista[i] : (method == SPCIES_EADMM ? eadmm[i] : admm[i]);
    }
    // MFMA-variant packing
    MfmaPlan mfma;
    Mfma4Plan mfma4;
    rtc::Mfma4Module mfma4_rtc;  // run-time compiled MFMA4 kernel when the shape was not instantiated at build time
    g4::Plan g4plan;  // MFMA4G (FISTA, EADMM)
    hgemm::Plan hgemm;             // GEMM (HMPC split, NON_SPARSE path)
    unsigned long long *d_hist = nullptr;  // spcies_hip_k_histogram_device's bins and counters
    int num_cu = 256;                      // compute units of the handle's device (hipGetDeviceProperties once, at create time)
    tvr::Plan tvrp;                        // time-varying ADMM, MFMA4R: one wavefront per instance, factors in registers (admm_tvr.hpp)
    std::string build_failures;    // the subset of `notes` that are failed builds (SPCIES_HIP_STRICT)
    std::string notes;             // which faster (run-time specialised) variants AUTO could not use, and why (spcies_hip_get_notes)
    fr::Plan frplan;               // MFMA4R (FISTA with the iteration state in registers + LDS, run-time specialised)
    er::Plan erplan;               // MFMA4R (MPCT EADMM, diagonal or general Q, R: the whole iteration state on the chip, run-time specialised)
    ar::Plan arplan;               // MFMA4R (lax / equ ADMM past MFMA4's register file / LDS: w on the chip, blocks streamed, run-time specialised)
    struct StreamRtc {             // STREAM for an (n, m) without a build-time kernel: the same text, specialised with hiprtc on first use
        bool ok = false, tried = false;
        std::string why;
        hipModule_t mod = nullptr;
        hipFunction_t fn = nullptr;
        hipFunction_t fn_update = nullptr;  // time-varying solvers: the update phase of the same (n, m)
    } srtc;
    hfused::Plan hfused;           // FUSED (HMPC split NON_SPARSE path: product + projections in one MFMA kernel)
    std::vector<double> h_M1, h_M2, h_bh_nat;
    bsp::Plan bsp;                 // BSP (ellipMPC soc): block-sparse MFMA program, generated per controller
    CsDev cdev{};                  // MPCT ADMM on the extended state space (STREAM, TILE)
    csfused::Plan csf;             // ... its FUSED variant (dense operator, state in registers)
    hdense::Host hd_host;          // HMPC without the splitting: blob contents, and its GEMM plan
    hdense::Plan hd_plan;
    tile::TileDev tdev{};          // TILE (soc, HMPC): step streams of the sparse operations
    std::vector<tile::Rec> tile_recs;
    int4 *d_recs = nullptr;
    // scratch of the STREAM variant (grown on demand)
    double *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    // staging buffers of the host entry point
    double *d_io = nullptr;
    size_t io_bytes = 0;
    // record fields the caller did not ask for, for the variants whose kernels write the whole record or nothing
    double *d_part = nullptr;
    size_t part_bytes = 0;
    hipStream_t stream = nullptr;  // owned; used by the host entry point
    std::mutex mu;
};

static const double *find_array(const uint8_t *blob, size_t bytes, const spcies_blob_header &h, uint32_t id,
                                uint64_t expect) {
    for (uint32_t i = 0; i < h.n_arrays; i++) {
        spcies_blob_entry e;
        memcpy(&e, blob + h.header_bytes + (size_t)i * sizeof(e), sizeof(e));
        if (e.id != id) continue;
        if (e.dtype != 0 || e.count != expect || e.offset % 8 != 0 || e.offset + e.count * 8 > bytes) return nullptr;
        return reinterpret_cast<const double *>(blob + e.offset);
    }
    return nullptr;
}

static const int *find_iarray(const uint8_t *blob, size_t bytes, const spcies_blob_header &h, uint32_t id, uint64_t *count) {
    for (uint32_t i = 0; i < h.n_arrays; i++) {
        spcies_blob_entry e;
        memcpy(&e, blob + h.header_bytes + (size_t)i * sizeof(e), sizeof(e));
        if (e.id != id) continue;
        if (e.dtype != 1 || e.offset % 4 != 0 || e.offset + e.count * 4 > bytes) return nullptr;
        *count = e.count;
        return reinterpret_cast<const int *>(blob + e.offset);
    }
    return nullptr;
}

static const double *find_farray_any(const uint8_t *blob, size_t bytes, const spcies_blob_header &h, uint32_t id, uint64_t *count) {
    for (uint32_t i = 0; i < h.n_arrays; i++) {
        spcies_blob_entry e;
        memcpy(&e, blob + h.header_bytes + (size_t)i * sizeof(e), sizeof(e));
        if (e.id != id) continue;
        if (e.dtype != 0 || e.offset % 8 != 0 || e.offset + e.count * 8 > bytes) return nullptr;
        *count = e.count;
        return reinterpret_cast<const double *>(blob + e.offset);
    }
    return nullptr;
}

// ellipMPC ADMM soc (cons_ellipMPC_ADMM_soc_C.m:66-117): dense Q, R, T, A, PhiP, bounds, and the sparse
// factors (CSC of L - I, Dinv, three CSR matrices).  Everything is validated here because the kernel
// follows these indices without further checks.
// option in_engineering (header flags bit3) of every generated solver: x0, xr, ur arrive in engineering units and are scaled
// on the way in, u is un-scaled on the way out (code_laxMPC_ADMM_C.c:83-100, 642-646; code_ellipMPC_ADMM_soc_C.c:63-72, 292-297;
// code_HMPC_ADMM_split_C.c:78-86, 356-360; code_HMPC_ADMM_C.c:64-72, 265-269; code_MPCT_ADMM_cs_C.c:56-64, 226-230)
static int parse_eng(const uint8_t *blob, size_t bytes, const spcies_blob_header &h, Solver &s) {
    s.eng = (h.flags & 8u) != 0;
    if (!s.eng) return 0;
    const uint32_t ids[5] = {SPCIES_A_SCALING_X, SPCIES_A_OPPOINT_X, SPCIES_A_SCALING_U, SPCIES_A_OPPOINT_U, SPCIES_A_SCALING_I_U};
    const uint64_t cnt[5] = {h.n, h.n, h.m, h.m, h.m};
    s.eng_v.clear();
    for (int i = 0; i < 5; i++) {
        const double *p = find_array(blob, bytes, h, ids[i], cnt[i]);
        if (!p) return fail(SPCIES_HIP_EINVAL, "in_engineering: blob array id %u missing or mis-sized", ids[i]);
        s.eng_v.insert(s.eng_v.end(), p, p + cnt[i]);
    }
    return 0;
}

static int parse_soc(const uint8_t *blob, size_t bytes, const spcies_blob_header &h, Solver &s) {
    s.formulation = (int)h.formulation; s.method = (int)h.method; s.submethod = (int)h.submethod;
    AdmmHost &a = s.host;
    a.n = (int)h.n; a.m = (int)h.m; a.N = (int)h.N; a.k_max = (int)h.k_max; a.terminal = true;
    a.tol = h.tol; a.rho = h.rho; a.rho_i = h.rho_i;
    if (h.n == 0 || h.m == 0 || h.N < 2 || h.n > 4096 || h.m > 4096 || h.N > 100000 || a.k_max <= 0 || !(a.rho > 0) || !(h.reserved[0] > 0))
        return fail(SPCIES_HIP_EINVAL, "bad n/m/N/k_max/rho/sigma");
    const int n = a.n, m = a.m, N = a.N, nm = n + m;
    SocDev &d = s.sdev;
    d.n = n; d.m = m; d.N = N; d.dim = N * nm + 1; d.n_s = n + 1; d.n_eq = N * n + 1; d.k_max = a.k_max;
    d.tol_p = h.tol; d.tol_d = h.reserved[2]; d.rho = h.rho; d.rho_i = h.rho_i; d.sigma = h.reserved[0]; d.sigma_i = h.reserved[1];
    const int np = d.dim + d.n_s, nr = d.n_eq + d.n_s;
    struct F { uint32_t id; uint64_t want; int *off; };  // want == 0: any length
    F fs[] = {{SPCIES_A_A, (uint64_t)n * n, &d.A}, {SPCIES_A_Q, (uint64_t)n * n, &d.Q}, {SPCIES_A_R, (uint64_t)m * m, &d.R},
              {SPCIES_A_T, (uint64_t)n * n, &d.T}, {SPCIES_A_LB, (uint64_t)(d.dim - n - 1), &d.LB},
              {SPCIES_A_UB, (uint64_t)(d.dim - n - 1), &d.UB}, {SPCIES_A_PHIP, (uint64_t)n * n, &d.PhiP},
              {SPCIES_A_L_VAL, 0, &d.L_val}, {SPCIES_A_DINV, (uint64_t)nr, &d.Dinv}, {SPCIES_A_GHHHI_VAL, 0, &d.GhHhi_val},
              {SPCIES_A_HHIGH_VAL, 0, &d.HhiGh_val}, {SPCIES_A_HHI_VAL, 0, &d.Hhi_val}};
    uint64_t nnz[4] = {0, 0, 0, 0};  // L, GhHhi, HhiGh, Hhi
    int vi = 0;
    for (auto &f : fs) {
        uint64_t cnt = 0;
        const double *p = find_farray_any(blob, bytes, h, f.id, &cnt);
        if (!p || (f.want && cnt != f.want)) return fail(SPCIES_HIP_EINVAL, "blob array id %u missing or mis-sized", f.id);
        *f.off = (int)s.soc_f64.size();
        s.soc_f64.insert(s.soc_f64.end(), p, p + cnt);
        while (s.soc_f64.size() % 8) s.soc_f64.push_back(0.0);
        if (f.want == 0) nnz[vi++] = cnt;
    }
    struct G { uint32_t id; uint64_t want; int *off; int maxval; bool is_ptr; };  // is_ptr: row/column pointer array
    G gs[] = {{SPCIES_A_L_COL, (uint64_t)nr + 1, &d.L_col, (int)nnz[0], true},
              {SPCIES_A_L_ROW, nnz[0], &d.L_row, nr - 1, false},
              {SPCIES_A_GHHHI_COL, nnz[1], &d.GhHhi_col, np - 1, false},
              {SPCIES_A_GHHHI_ROW, (uint64_t)nr + 1, &d.GhHhi_row, (int)nnz[1], true},
              {SPCIES_A_HHIGH_COL, nnz[2], &d.HhiGh_col, nr - 1, false},
              {SPCIES_A_HHIGH_ROW, (uint64_t)np + 1, &d.HhiGh_row, (int)nnz[2], true},
              {SPCIES_A_HHI_COL, nnz[3], &d.Hhi_col, np - 1, false},
              {SPCIES_A_HHI_ROW, (uint64_t)np + 1, &d.Hhi_row, (int)nnz[3], true}};
    for (auto &g : gs) {
        uint64_t cnt = 0;
        const int *p = find_iarray(blob, bytes, h, g.id, &cnt);
        if (!p || cnt != g.want) return fail(SPCIES_HIP_EINVAL, "blob index array id %u missing or mis-sized", g.id);
        const bool is_ptr = g.is_ptr;
        for (uint64_t i = 0; i < cnt; i++) {
            if (p[i] < 0 || p[i] > g.maxval) return fail(SPCIES_HIP_EINVAL, "blob index array id %u: value out of range", g.id);
            if (is_ptr && i > 0 && p[i] < p[i - 1]) return fail(SPCIES_HIP_EINVAL, "blob pointer array id %u not monotone", g.id);
        }
        if (is_ptr && (p[0] != 0 || p[cnt - 1] != g.maxval)) return fail(SPCIES_HIP_EINVAL, "blob pointer array id %u: bad ends", g.id);
        *g.off = (int)s.soc_i32.size();
        s.soc_i32.insert(s.soc_i32.end(), p, p + cnt);
    }
    // CSC of L - I must be strictly lower triangular (the forward sweep relies on it)
    {
        const int *Lc = s.soc_i32.data() + d.L_col, *Lr = s.soc_i32.data() + d.L_row;
        for (int i = 0; i < nr; i++)
            for (int j = Lc[i]; j < Lc[i + 1]; j++)
                if (Lr[j] <= i) return fail(SPCIES_HIP_EINVAL, "L - I is not strictly lower triangular");
    }
    {  // TILE variant: step streams (sparse_tile.hpp)
        const int *I = s.soc_i32.data();
        const double *F = s.soc_f64.data();
        const int lpi = tile::pick_lpi((long)nr + 2L * np, I + d.L_col, nr);
        s.tdev = tile::TileDev{};
        s.tdev.lpi = lpi;
        if (lpi) {
            s.tdev.lds_bytes = (size_t)(nr + 2 * np) * (64 / lpi) * sizeof(double);
            tile::build_ldl_streams(nr, I + d.L_col, I + d.L_row, F + d.L_val, lpi, s.tile_recs, s.tdev.fwd, s.tdev.bwd);
            tile::build_spmv_stream(nr, I + d.GhHhi_row, I + d.GhHhi_col, F + d.GhHhi_val, nr, nullptr, nullptr, nullptr, 0, lpi,
                                    s.tile_recs, s.tdev.rhs);
            tile::build_spmv_stream(np, I + d.Hhi_row, I + d.Hhi_col, F + d.Hhi_val, nr, I + d.HhiGh_row, I + d.HhiGh_col,
                                    F + d.HhiGh_val, 0, lpi, s.tile_recs, s.tdev.prim);
        }
    }
    return bsp::build_soc(s.bsp, d, s.soc_f64.data(), s.soc_i32.data());  // BSP variant: generate the program (host only)
}

// HMPC ADMM / SADMM split, sparse KKT path, box constraints (cons_HMPC_ADMM_split_C.m:88-181)
static int parse_hmpc(const uint8_t *blob, size_t bytes, const spcies_blob_header &h, Solver &s) {
    s.formulation = (int)h.formulation; s.method = (int)h.method; s.submethod = (int)h.submethod;
    AdmmHost &a = s.host;
    a.n = (int)h.n; a.m = (int)h.m; a.N = (int)h.N; a.k_max = (int)h.k_max; a.terminal = true;
    a.tol = h.tol; a.rho = h.rho; a.rho_i = h.rho_i;
    if (h.n == 0 || h.m == 0 || h.N < 3 || h.n > 4096 || h.m > 4096 || h.N > 100000 || a.k_max <= 0 || !(a.rho > 0) || !(h.reserved[0] > 0))
        return fail(SPCIES_HIP_EINVAL, "bad n/m/N/k_max/rho/sigma");
    const int n = a.n, m = a.m, N = a.N, nm = n + m;
    HmpcDev &d = s.hdev;
    d.n = n; d.m = m; d.N = N; d.use_soc = (h.flags & 2u) ? 1 : 0; d.symmetric = (h.method == SPCIES_SADMM);
    // header flags bit6: coupled output constraints (COUPLED_CONSTRAINTS); n_y = rows of E / F = entries of LBy
    d.coupled = (h.flags & 64u) ? 1 : 0;
    d.n_y = nm;
    if (d.coupled) {
        uint64_t cnt_lby = 0;
        if (!find_farray_any(blob, bytes, h, SPCIES_A_LBY, &cnt_lby) || cnt_lby == 0 || cnt_lby > 4096)
            return fail(SPCIES_HIP_EINVAL, "HMPC coupled constraints: LBy missing or mis-sized");
        d.n_y = (int)cnt_lby;
    }
    d.dim = (N - 1) * nm + m + 3 * nm; d.n_eq = (N + 3) * n; d.n_soc = d.use_soc ? 2 * d.n_y : d.n_y;
    d.n_s = 3 * d.n_soc + (d.coupled ? N * d.n_y : 0);
    d.nrow_M = d.dim + d.n_s + d.n_eq + d.n_s; d.k_max = a.k_max;
    d.tol_p = h.tol; d.tol_d = h.reserved[2]; d.rho = h.rho; d.rho_i = h.rho_i; d.sigma = h.reserved[0];
    d.sigma_i = h.reserved[1]; d.alpha = h.reserved[3];
    const int nc = d.n_eq + d.n_s;
    struct F { uint32_t id; uint64_t want; int *off; };
    F fs[] = {{SPCIES_A_A, (uint64_t)n * n, &d.A}, {SPCIES_A_Q, (uint64_t)n * n, &d.QQ}, {SPCIES_A_TE, (uint64_t)n * n, &d.Te},
              {SPCIES_A_SE, (uint64_t)m * m, &d.Se}, {SPCIES_A_LB, (uint64_t)(d.dim - 3 * nm), &d.LB},
              {SPCIES_A_UB, (uint64_t)(d.dim - 3 * nm), &d.UB}, {SPCIES_A_LBY, (uint64_t)d.n_y, &d.LBy},
              {SPCIES_A_UBY, (uint64_t)d.n_y, &d.UBy}, {SPCIES_A_L_VAL, 0, &d.L_val}, {SPCIES_A_DINV, (uint64_t)d.nrow_M, &d.Dinv},
              {SPCIES_A_BH, (uint64_t)nc, &d.bh}};
    uint64_t nnz = 0;
    for (auto &f : fs) {
        uint64_t cnt = 0;
        const double *p = find_farray_any(blob, bytes, h, f.id, &cnt);
        if (!p || (f.want && cnt != f.want)) return fail(SPCIES_HIP_EINVAL, "blob array id %u missing or mis-sized", f.id);
        *f.off = (int)s.soc_f64.size();
        s.soc_f64.insert(s.soc_f64.end(), p, p + cnt);
        while (s.soc_f64.size() % 8) s.soc_f64.push_back(0.0);
        if (f.want == 0) nnz = cnt;
    }
    struct G { uint32_t id; uint64_t want; int *off; int maxval; bool is_ptr; };
    G gs[] = {{SPCIES_A_L_COL, (uint64_t)d.nrow_M + 1, &d.L_col, (int)nnz, true},
              {SPCIES_A_L_ROW, nnz, &d.L_row, d.nrow_M - 1, false},
              {SPCIES_A_IDX_X0, (uint64_t)n, &d.idx_x0, nc - 1, false}};
    for (auto &g : gs) {
        uint64_t cnt = 0;
        const int *p = find_iarray(blob, bytes, h, g.id, &cnt);
        if (!p || cnt != g.want) return fail(SPCIES_HIP_EINVAL, "blob index array id %u missing or mis-sized", g.id);
        for (uint64_t i = 0; i < cnt; i++) {
            if (p[i] < 0 || p[i] > g.maxval) return fail(SPCIES_HIP_EINVAL, "blob index array id %u: value out of range", g.id);
            if (g.is_ptr && i > 0 && p[i] < p[i - 1]) return fail(SPCIES_HIP_EINVAL, "blob pointer array id %u not monotone", g.id);
        }
        if (g.is_ptr && (p[0] != 0 || p[cnt - 1] != g.maxval)) return fail(SPCIES_HIP_EINVAL, "blob pointer array id %u: bad ends", g.id);
        *g.off = (int)s.soc_i32.size();
        s.soc_i32.insert(s.soc_i32.end(), p, p + cnt);
    }
    const int *Lc = s.soc_i32.data() + d.L_col, *Lr = s.soc_i32.data() + d.L_row;
    for (int i = 0; i < d.nrow_M; i++)
        for (int j = Lc[i]; j < Lc[i + 1]; j++)
            if (Lr[j] <= i) return fail(SPCIES_HIP_EINVAL, "L - I is not strictly lower triangular");
    {  // optional dense matrices of the NON_SPARSE path (GEMM variant)
        const int np = d.dim + d.n_s;
        const double *pM1 = find_array(blob, bytes, h, SPCIES_A_M1, (uint64_t)np * np);
        const double *pM2 = find_array(blob, bytes, h, SPCIES_A_M2, (uint64_t)np * nc);
        const double *pbh = find_array(blob, bytes, h, SPCIES_A_BH_NAT, (uint64_t)nc);
        if (pM1 && pM2 && pbh) {
            s.h_M1.assign(pM1, pM1 + (size_t)np * np);
            s.h_M2.assign(pM2, pM2 + (size_t)np * nc);
            s.h_bh_nat.assign(pbh, pbh + nc);
        }
    }
    {  // TILE variant: step streams (sparse_tile.hpp)
        const int lpi = tile::pick_lpi((long)d.nrow_M, Lc, d.nrow_M);
        s.tdev = tile::TileDev{};
        s.tdev.lpi = lpi;
        if (lpi) {
            s.tdev.lds_bytes = (size_t)d.nrow_M * (64 / lpi) * sizeof(double);
            tile::build_ldl_streams(d.nrow_M, Lc, Lr, s.soc_f64.data() + d.L_val, lpi, s.tile_recs, s.tdev.fwd, s.tdev.bwd);
        }
    }
    return 0;
}

// MPCT ADMM on the extended state space (cons_MPCT_ADMM_cs_C.m:66-112)
static int parse_mpct_cs(const uint8_t *blob, size_t bytes, const spcies_blob_header &h, Solver &s) {
    s.formulation = (int)h.formulation; s.method = (int)h.method; s.submethod = (int)h.submethod;
    AdmmHost &a = s.host;
    a.n = (int)h.n; a.m = (int)h.m; a.N = (int)h.N; a.k_max = (int)h.k_max; a.terminal = true;
    a.tol = h.tol; a.rho = h.rho; a.rho_i = h.rho_i;
    if (h.n == 0 || h.m == 0 || h.N < 2 || h.n > 4096 || h.m > 4096 || h.N > 100000 || a.k_max <= 0 || !(a.rho > 0))
        return fail(SPCIES_HIP_EINVAL, "bad n/m/N/k_max/rho");
    const int n = a.n, m = a.m, N = a.N, dnm = 2 * (n + m);
    CsDev &d = s.cdev;
    d.n = n; d.m = m; d.N = N; d.dim = N * dnm; d.nrow = 2 * n + (2 * n + m) * (N - 1) + n; d.k_max = a.k_max;
    d.scalar_rho = (h.flags & 1u) ? 1 : 0;
    d.tol = h.tol; d.rho = h.rho; d.rho_i = h.rho_i;
    const int dim = d.dim, nr = d.nrow;
    struct F { uint32_t id; uint64_t want; int *off; bool optional; };  // want == 0: any length
    F fs[] = {{SPCIES_A_TZ, (uint64_t)n * n, &d.Tz, false}, {SPCIES_A_SZ, (uint64_t)m * m, &d.Sz, false},
              {SPCIES_A_LB, (uint64_t)dim, &d.LB, false}, {SPCIES_A_UB, (uint64_t)dim, &d.UB, false},
              {SPCIES_A_L_VAL, 0, &d.L_val, false}, {SPCIES_A_DINV, (uint64_t)nr, &d.Dinv, false},
              {SPCIES_A_AHI_VAL, 0, &d.AHi_val, false}, {SPCIES_A_HIA_VAL, 0, &d.HiA_val, false}, {SPCIES_A_HI_VAL, 0, &d.Hi_val, false},
              {SPCIES_A_RHO_CS, (uint64_t)dim, &d.rho_v, d.scalar_rho != 0}, {SPCIES_A_RHO_I_CS, (uint64_t)dim, &d.rho_i_v, d.scalar_rho != 0}};
    uint64_t nnz[4] = {0, 0, 0, 0};  // L, AHi, HiA, Hi
    int vi = 0;
    for (auto &f : fs) {
        uint64_t cnt = 0;
        const double *p = find_farray_any(blob, bytes, h, f.id, &cnt);
        *f.off = 0;
        if (f.optional) continue;
        if (!p || (f.want && cnt != f.want)) return fail(SPCIES_HIP_EINVAL, "blob array id %u missing or mis-sized", f.id);
        *f.off = (int)s.soc_f64.size();
        s.soc_f64.insert(s.soc_f64.end(), p, p + cnt);
        while (s.soc_f64.size() % 8) s.soc_f64.push_back(0.0);
        if (f.want == 0) nnz[vi++] = cnt;
    }
    if (!d.scalar_rho)
        for (int j = 0; j < dim; j++)
            if (!(s.soc_f64[d.rho_v + j] > 0)) return fail(SPCIES_HIP_EINVAL, "bad rho");
    struct G { uint32_t id; uint64_t want; int *off; int maxval; bool is_ptr; };
    G gs[] = {{SPCIES_A_L_COL, (uint64_t)nr + 1, &d.L_col, (int)nnz[0], true},
              {SPCIES_A_L_ROW, nnz[0], &d.L_row, nr - 1, false},
              {SPCIES_A_AHI_COL, nnz[1], &d.AHi_col, dim - 1, false},
              {SPCIES_A_AHI_ROW, (uint64_t)nr + 1, &d.AHi_row, (int)nnz[1], true},
              {SPCIES_A_HIA_COL, nnz[2], &d.HiA_col, nr - 1, false},
              {SPCIES_A_HIA_ROW, (uint64_t)dim + 1, &d.HiA_row, (int)nnz[2], true},
              {SPCIES_A_HI_COL, nnz[3], &d.Hi_col, dim - 1, false},
              {SPCIES_A_HI_ROW, (uint64_t)dim + 1, &d.Hi_row, (int)nnz[3], true}};
    for (auto &g : gs) {
        uint64_t cnt = 0;
        const int *p = find_iarray(blob, bytes, h, g.id, &cnt);
        if (!p || cnt != g.want) return fail(SPCIES_HIP_EINVAL, "blob index array id %u missing or mis-sized", g.id);
        for (uint64_t i = 0; i < cnt; i++) {
            if (p[i] < 0 || p[i] > g.maxval) return fail(SPCIES_HIP_EINVAL, "blob index array id %u: value out of range", g.id);
            if (g.is_ptr && i > 0 && p[i] < p[i - 1]) return fail(SPCIES_HIP_EINVAL, "blob pointer array id %u not monotone", g.id);
        }
        if (g.is_ptr && (p[0] != 0 || p[cnt - 1] != g.maxval)) return fail(SPCIES_HIP_EINVAL, "blob pointer array id %u: bad ends", g.id);
        *g.off = (int)s.soc_i32.size();
        s.soc_i32.insert(s.soc_i32.end(), p, p + cnt);
    }
    {
        const int *Lc = s.soc_i32.data() + d.L_col, *Lr = s.soc_i32.data() + d.L_row;
        for (int i = 0; i < nr; i++)
            for (int j = Lc[i]; j < Lc[i + 1]; j++)
                if (Lr[j] <= i) return fail(SPCIES_HIP_EINVAL, "L - I is not strictly lower triangular");
    }
    {  // TILE variant: step streams (sparse_tile.hpp); LDS rows RH (nr) | QH (dim) | PL (dim)
        const int *I = s.soc_i32.data();
        const double *F = s.soc_f64.data();
        const int lpi = tile::pick_lpi((long)nr + 2L * dim, I + d.L_col, nr);
        s.tdev = tile::TileDev{};
        s.tdev.lpi = lpi;
        if (lpi) {
            s.tdev.lds_bytes = (size_t)(nr + 2 * dim) * (64 / lpi) * sizeof(double);
            tile::build_ldl_streams(nr, I + d.L_col, I + d.L_row, F + d.L_val, lpi, s.tile_recs, s.tdev.fwd, s.tdev.bwd);
            tile::build_spmv_stream(nr, I + d.AHi_row, I + d.AHi_col, F + d.AHi_val, nr, nullptr, nullptr, nullptr, 0, lpi,
                                    s.tile_recs, s.tdev.rhs);
            tile::build_spmv_stream(dim, I + d.Hi_row, I + d.Hi_col, F + d.Hi_val, nr, I + d.HiA_row, I + d.HiA_col,
                                    F + d.HiA_val, 0, lpi, s.tile_recs, s.tdev.prim);
        }
    }
    return 0;
}

// HMPC ADMM / SADMM without the splitting, box constraints (cons_HMPC_ADMM_C.m:88-131)
static int parse_hmpc_dense(const uint8_t *blob, size_t bytes, const spcies_blob_header &h, Solver &s) {
    s.formulation = (int)h.formulation; s.method = (int)h.method; s.submethod = (int)h.submethod;
    AdmmHost &a = s.host;
    a.n = (int)h.n; a.m = (int)h.m; a.N = (int)h.N; a.k_max = (int)h.k_max; a.terminal = true;
    a.tol = h.tol; a.rho = h.rho; a.rho_i = h.rho_i;
    if (h.n == 0 || h.m == 0 || h.N < 3 || h.n > 4096 || h.m > 4096 || h.N > 100000 || a.k_max <= 0 || !(a.rho > 0))
        return fail(SPCIES_HIP_EINVAL, "bad n/m/N/k_max/rho");
    const int n = a.n, m = a.m, N = a.N, nm = n + m;
    hdense::Host &d = s.hd_host;
    d.n = n; d.m = m; d.N = N; d.use_soc = (h.flags & 2u) ? 1 : 0; d.symmetric = (h.method == SPCIES_SADMM);
    // box constraints: n_y = n + m outputs, n_box = dim - 3(n+m) slacks of the decision variables; coupled output constraints
    // (compute_HMPC_ADMM_ingredients.m:155-180): n_y = rows of E / F, n_box = N n_y - both read off the array sizes
    uint64_t cnt_lby = 0, cnt_lb = 0;
    if (!find_farray_any(blob, bytes, h, SPCIES_A_LBY, &cnt_lby) || !find_farray_any(blob, bytes, h, SPCIES_A_LB, &cnt_lb) || cnt_lby == 0 ||
        cnt_lby > 4096 || cnt_lb > (uint64_t)(1 << 24))
        return fail(SPCIES_HIP_EINVAL, "HMPC: LBy / LB missing or mis-sized");
    const int n_y = (int)cnt_lby;
    d.dim = (N - 1) * nm + m + 3 * nm; d.n_eq = (N + 3) * n; d.n_soc = d.use_soc ? 2 * n_y : n_y; d.n_box = (int)cnt_lb;
    if (d.n_box != d.dim - 3 * nm && d.n_box != N * n_y) return fail(SPCIES_HIP_EINVAL, "HMPC: LB has neither dim - 3(n+m) nor N n_y entries");
    d.n_s = d.n_box + 3 * d.n_soc; d.k_max = a.k_max;
    if ((long)d.dim * d.dim > (1L << 28)) return fail(SPCIES_HIP_EINVAL, "HMPC: dim too large for the dense M1");
    d.tol_p = h.tol; d.tol_d = h.reserved[2]; d.rho = h.rho; d.rho_i = h.rho_i;
    d.alpha = d.symmetric ? h.reserved[3] : 1.0;
    struct F { uint32_t id; uint64_t want; std::vector<double> *dst; bool optional; };
    F fs[] = {{SPCIES_A_A, (uint64_t)n * n, &d.A, false}, {SPCIES_A_Q, (uint64_t)n * n, &d.QQ, false},
              {SPCIES_A_TE, (uint64_t)n * n, &d.Te, false}, {SPCIES_A_SE, (uint64_t)m * m, &d.Se, false},
              {SPCIES_A_LB, (uint64_t)d.n_box, &d.LB, false}, {SPCIES_A_UB, (uint64_t)d.n_box, &d.UB, false},
              {SPCIES_A_LBY, (uint64_t)n_y, &d.LBy, false}, {SPCIES_A_UBY, (uint64_t)n_y, &d.UBy, false},
              {SPCIES_A_D, (uint64_t)d.n_s, &d.d, !d.use_soc}, {SPCIES_A_C_VAL, 0, &d.C_val, false},
              {SPCIES_A_CT_VAL, 0, &d.Ct_val, false}, {SPCIES_A_M1, (uint64_t)d.dim * d.dim, &d.M1, false},
              {SPCIES_A_M2, (uint64_t)d.dim * n, &d.M2, false}};
    for (auto &f : fs) {
        uint64_t cnt = 0;
        const double *p = find_farray_any(blob, bytes, h, f.id, &cnt);
        if (!p && f.optional) continue;
        if (!p || (f.want && cnt != f.want)) return fail(SPCIES_HIP_EINVAL, "blob array id %u missing or mis-sized", f.id);
        f.dst->assign(p, p + cnt);
    }
    if (d.C_val.size() != d.Ct_val.size() || d.C_val.size() > (size_t)d.n_s * d.dim)
        return fail(SPCIES_HIP_EINVAL, "C and C' hold different numbers of non-zeros");
    const int nnz = (int)d.C_val.size();
    struct G { uint32_t id; uint64_t want; std::vector<int> *dst; int maxval; bool is_ptr; };
    G gs[] = {{SPCIES_A_C_ROW, (uint64_t)d.n_s + 1, &d.C_row, nnz, true}, {SPCIES_A_C_COL, (uint64_t)nnz, &d.C_col, d.dim - 1, false},
              {SPCIES_A_CT_ROW, (uint64_t)d.dim + 1, &d.Ct_row, nnz, true}, {SPCIES_A_CT_COL, (uint64_t)nnz, &d.Ct_col, d.n_s - 1, false}};
    for (auto &g : gs) {
        uint64_t cnt = 0;
        const int *p = find_iarray(blob, bytes, h, g.id, &cnt);
        if (!p || cnt != g.want) return fail(SPCIES_HIP_EINVAL, "blob index array id %u missing or mis-sized", g.id);
        for (uint64_t i = 0; i < cnt; i++) {
            if (p[i] < 0 || p[i] > g.maxval) return fail(SPCIES_HIP_EINVAL, "blob index array id %u: value out of range", g.id);
            if (g.is_ptr && i > 0 && p[i] < p[i - 1]) return fail(SPCIES_HIP_EINVAL, "blob pointer array id %u not monotone", g.id);
        }
        if (g.is_ptr && (p[0] != 0 || p[cnt - 1] != g.maxval)) return fail(SPCIES_HIP_EINVAL, "blob pointer array id %u: bad ends", g.id);
        g.dst->assign(p, p + cnt);
    }
    return 0;
}

static int parse_blob(const void *blobv, size_t bytes, Solver &s) {
    const uint8_t *blob = static_cast<const uint8_t *>(blobv);
    if (!blob || bytes < sizeof(spcies_blob_header)) return fail(SPCIES_HIP_EINVAL, "blob too small");
    spcies_blob_header h;
    memcpy(&h, blob, sizeof(h));
    if (memcmp(h.magic, SPCIES_BLOB_MAGIC, 8) != 0) return fail(SPCIES_HIP_EINVAL, "bad blob magic");
    if (h.version != SPCIES_BLOB_VERSION || h.header_bytes != sizeof(h) || h.total_bytes != bytes)
        return fail(SPCIES_HIP_EINVAL, "blob version/size mismatch");
    if ((size_t)h.header_bytes + (size_t)h.n_arrays * sizeof(spcies_blob_entry) > bytes)
        return fail(SPCIES_HIP_EINVAL, "blob directory out of range");
    const bool ellip_admm = (h.method == SPCIES_ADMM && h.formulation == SPCIES_ELLIPMPC && h.submethod == 0);
    const bool banded = ((h.method == SPCIES_ADMM || h.method == SPCIES_FISTA) &&
                         (h.formulation == SPCIES_LAXMPC || h.formulation == SPCIES_EQUMPC)) || ellip_admm;
    const bool mpct = (h.method == SPCIES_EADMM && h.formulation == SPCIES_MPCT);
    const bool soc = (h.method == SPCIES_ADMM && h.formulation == SPCIES_ELLIPMPC && h.submethod == 1);
    if (soc || h.formulation == SPCIES_HMPC || (h.formulation == SPCIES_MPCT && h.method == SPCIES_ADMM)) {
        int rc = parse_eng(blob, bytes, h, s);
        if (rc) return rc;
    }
    if (soc) return parse_soc(blob, bytes, h, s);
    if (h.formulation == SPCIES_HMPC && (h.method == SPCIES_ADMM || h.method == SPCIES_SADMM) && h.submethod == 2)
        return parse_hmpc(blob, bytes, h, s);
    if (h.formulation == SPCIES_HMPC && (h.method == SPCIES_ADMM || h.method == SPCIES_SADMM) && h.submethod == 0)
        return parse_hmpc_dense(blob, bytes, h, s);
    if (h.formulation == SPCIES_MPCT && h.method == SPCIES_ADMM) {
        if (h.submethod != 3) return fail(SPCIES_HIP_ENOSUP, "MPCT ADMM: the 'cs' submethod is built (not 'semiband')");
        return parse_mpct_cs(blob, bytes, h, s);
    }
    if (!banded && !mpct)
        return fail(SPCIES_HIP_ENOSUP, "formulation %u / method %u not built in this library", h.formulation, h.method);
    const bool vec_rho = (h.method == SPCIES_ADMM && !(h.flags & 1u)), var_b = (h.flags & 16u) != 0;
    if ((vec_rho || var_b) && !(h.method == SPCIES_ADMM && (h.formulation == SPCIES_LAXMPC || h.formulation == SPCIES_EQUMPC)) &&
        !(ellip_admm && !var_b))  // ellipMPC ADMM: vector rho (cons_ellipMPC_ADMM_C.m:111-117); its bounds are stage-wise already
        return fail(SPCIES_HIP_ENOSUP, "vector rho / stage-wise bounds are built for the lax/equ/ellip MPC ADMM solvers only");
    if ((vec_rho || var_b) && (h.flags & 4u)) return fail(SPCIES_HIP_EINVAL, "time-varying solvers take a scalar rho and constant bounds");
    if (h.n == 0 || h.m == 0 || h.N < 2 || h.n > 4096 || h.m > 4096 || h.N > 100000) return fail(SPCIES_HIP_EINVAL, "bad n/m/N");
    s.formulation = (int)h.formulation;
    s.method = (int)h.method;
    s.submethod = (int)h.submethod;
    AdmmHost &a = s.host;
    a.n = (int)h.n; a.m = (int)h.m; a.N = (int)h.N; a.k_max = (int)h.k_max;
    a.terminal = (h.formulation != SPCIES_EQUMPC);
    {
        int rc = parse_eng(blob, bytes, h, s);
        if (rc) return rc;
    }
    a.tol = h.tol; a.rho = h.rho; a.rho_i = h.rho_i;
    if (a.k_max <= 0 || (h.method == SPCIES_ADMM && !vec_rho && !(a.rho > 0))) return fail(SPCIES_HIP_EINVAL, "bad rho / k_max");
    const uint64_t n = h.n, m = h.m, N = h.N, nm = n + m;
    struct Want { uint32_t id; uint64_t count; std::vector<double> *dst; };
    std::vector<Want> want = {{SPCIES_A_AB, n * nm, &a.AB},       {SPCIES_A_ALPHA, (N - 1) * n * n, &a.Alpha},
                              {SPCIES_A_BETA, N * n * n, &a.Beta}, {SPCIES_A_LB, nm, &a.LB},
                              {SPCIES_A_UB, nm, &a.UB}};
    a.ellip = ellip_admm;
    if (ellip_admm) {  // no LB / UB: stage-wise bounds and the ellipsoid instead (cons_ellipMPC_ADMM_C.m:74-110)
        want = {{SPCIES_A_AB, n * nm, &a.AB}, {SPCIES_A_ALPHA, (N - 1) * n * n, &a.Alpha}, {SPCIES_A_BETA, N * n * n, &a.Beta},
                {SPCIES_A_P, n * n, &a.P}, {SPCIES_A_P_HALF, n * n, &a.P_half}, {SPCIES_A_PINV_HALF, n * n, &a.Pinv_half},
                {SPCIES_A_C_ELL, n, &a.c_ell}, {SPCIES_A_LBZ, (N - 1) * nm, &a.LBz}, {SPCIES_A_UBZ, (N - 1) * nm, &a.UBz},
                {SPCIES_A_LBU0, m, &a.LBu0}, {SPCIES_A_UBU0, m, &a.UBu0}};
        a.r_ell = h.reserved[4];
        if (!(a.r_ell >= 0)) return fail(SPCIES_HIP_EINVAL, "ellipMPC: bad ellipsoid radius");
    }
    s.tv = (h.flags & 4u) != 0;
    if (s.tv) {
        if ((h.method != SPCIES_ADMM && h.method != SPCIES_FISTA) || !banded || ellip_admm)
            return fail(SPCIES_HIP_ENOSUP, "time-varying: built for the laxMPC / equMPC ADMM and FISTA solvers");
        if (h.method == SPCIES_ADMM)
            want = {{SPCIES_A_T, n * n, &a.T}, {SPCIES_A_T_RHO_I, n * n, &a.Hi_N}};  // Hi_N = T_rho_i (code_laxMPC_ADMM_C.c:123)
        else  // FISTA: the diagonal terminal weight and its inverse are the only constants (cons_laxMPC_FISTA_C.m:94-108)
            want = {{SPCIES_A_TDIAG, n, &s.Td}, {SPCIES_A_TI, n, &s.Ti}};
    }
    if (h.method != SPCIES_EADMM && !s.tv) {
        want.push_back({SPCIES_A_Q, n, &a.Q});
        want.push_back({SPCIES_A_R, m, &a.R});
    }
    if (s.tv) {
    } else if (h.method == SPCIES_EADMM) {
        want.push_back({SPCIES_A_T, n * n, &a.T});
        want.push_back({SPCIES_A_S, m * m, &s.e_S});
        want.push_back({SPCIES_A_RHO_MAT, (N + 1) * nm, &s.e_rho});
        want.push_back({SPCIES_A_RHO_0, nm, &s.e_rho0});
        want.push_back({SPCIES_A_RHO_S, nm, &s.e_rhos});
        want.push_back({SPCIES_A_LB_0, nm, &s.e_LB0});
        want.push_back({SPCIES_A_UB_0, nm, &s.e_UB0});
        want.push_back({SPCIES_A_LB_S, nm, &s.e_LBs});
        want.push_back({SPCIES_A_UB_S, nm, &s.e_UBs});
        want.push_back({SPCIES_A_H1I, (N + 1) * nm, &s.e_H1i});
        want.push_back({SPCIES_A_W2, nm * nm, &s.e_W2});
        s.e_general = (h.flags & 32u) != 0;
        if (!s.e_general) {
            want.push_back({SPCIES_A_H3I, (N + 1) * nm, &s.e_H3i});
        } else {
            want.push_back({SPCIES_A_Q_BI, n * n, &s.e_Qbi});
            want.push_back({SPCIES_A_Q_MI, n * n, &s.e_Qmi});
            want.push_back({SPCIES_A_R_BI, m * m, &s.e_Rbi});
            want.push_back({SPCIES_A_R_MI, m * m, &s.e_Rmi});
            want.push_back({SPCIES_A_AB_BI, n * nm, &s.e_ABbi});
            want.push_back({SPCIES_A_AB_MI, n * nm, &s.e_ABmi});
        }
    } else if (h.method == SPCIES_ADMM) {
        want.push_back({SPCIES_A_HI, (N - 1) * nm, &a.Hi});
        want.push_back({SPCIES_A_HI_0, m, &a.Hi_0});
        want.push_back({SPCIES_A_HI_N, n * n, &a.Hi_N});
        want.push_back({SPCIES_A_T, n * n, &a.T});
    } else {
        want.push_back({SPCIES_A_QRI, nm, &s.QRi});
        want.push_back({SPCIES_A_TDIAG, n, &s.Td});
        want.push_back({SPCIES_A_TI, n, &s.Ti});
    }
    if (var_b) {  // LB / UB are [N-1][n+m] here
        for (auto &w : want)
            if (w.id == SPCIES_A_LB || w.id == SPCIES_A_UB) w.count = (N - 1) * nm;
        want.push_back({SPCIES_A_LB_0, m, &a.LBu0});
        want.push_back({SPCIES_A_UB_0, m, &a.UBu0});
        if (a.terminal) {
            want.push_back({SPCIES_A_LBN, n, &a.LBN});
            want.push_back({SPCIES_A_UBN, n, &a.UBN});
        }
    }
    if (vec_rho) {
        want.push_back({SPCIES_A_RHO_0, m, &a.rho_0});
        want.push_back({SPCIES_A_RHO_V, (N - 1) * nm, &a.rho_v});
        want.push_back({SPCIES_A_RHO_I_0, m, &a.rho_i_0});
        want.push_back({SPCIES_A_RHO_I_V, (N - 1) * nm, &a.rho_i_v});
        if (a.terminal) {
            want.push_back({SPCIES_A_RHO_N, n, &a.rho_N});
            want.push_back({SPCIES_A_RHO_I_N, n, &a.rho_i_N});
        }
    }
    for (auto &w : want) {
        const double *p = find_array(blob, bytes, h, w.id, w.count);
        if (!p) return fail(SPCIES_HIP_EINVAL, "blob array id %u missing or mis-sized", w.id);
        w.dst->assign(p, p + w.count);
    }
    a.gen = vec_rho || var_b;
    if (a.gen) {  // keep the stage-wise form of both switches
        if (a.ellip) {
            a.LBN.assign(n, 0.0); a.UBN.assign(n, 0.0);  // (no terminal box: the ellipsoid)
        } else if (var_b) {
            a.LBz = a.LB; a.UBz = a.UB;
        } else {
            a.LBu0.assign(a.LB.begin() + n, a.LB.end()); a.UBu0.assign(a.UB.begin() + n, a.UB.end());
            a.LBN.assign(a.LB.begin(), a.LB.begin() + n); a.UBN.assign(a.UB.begin(), a.UB.begin() + n);
            a.LBz.clear(); a.UBz.clear();
            for (uint64_t l = 0; l + 1 < N; l++) {
                a.LBz.insert(a.LBz.end(), a.LB.begin(), a.LB.end());
                a.UBz.insert(a.UBz.end(), a.UB.begin(), a.UB.end());
            }
        }
        if (!a.terminal) { a.LBN.assign(n, 0.0); a.UBN.assign(n, 0.0); }
        if (!vec_rho) {
            a.rho_0.assign(m, a.rho); a.rho_v.assign((N - 1) * nm, a.rho); a.rho_N.assign(n, a.rho);
            a.rho_i_0.assign(m, a.rho_i); a.rho_i_v.assign((N - 1) * nm, a.rho_i); a.rho_i_N.assign(n, a.rho_i);
        } else if (!a.terminal) {
            a.rho_N.assign(n, 1.0); a.rho_i_N.assign(n, 1.0);
        }
        for (double r : a.rho_0) if (!(r > 0)) return fail(SPCIES_HIP_EINVAL, "bad rho");
        for (double r : a.rho_v) if (!(r > 0)) return fail(SPCIES_HIP_EINVAL, "bad rho");
        for (double r : a.rho_N) if (!(r > 0)) return fail(SPCIES_HIP_EINVAL, "bad rho");
    }
    if (a.ellip) return bsp::build_ellip(s.bsp, a);  // BSP variant: generate the controller's block program (host only)
    return 0;
}

static int upload_consts(Solver &s) {
    AdmmHost &a = s.host;
    if (s.is_hdense()) return hdense::plan_build(s.hd_plan, s.hd_host);
    if (s.is_soc() || s.is_cs()) {
        SPCIES_HIP_CHECK(hipMalloc((void **)&s.d_consts, s.soc_f64.size() * sizeof(double)));
        SPCIES_HIP_CHECK(hipMemcpy(s.d_consts, s.soc_f64.data(), s.soc_f64.size() * sizeof(double), hipMemcpyHostToDevice));
        SPCIES_HIP_CHECK(hipMalloc((void **)&s.d_idx, s.soc_i32.size() * sizeof(int)));
        SPCIES_HIP_CHECK(hipMemcpy(s.d_idx, s.soc_i32.data(), s.soc_i32.size() * sizeof(int), hipMemcpyHostToDevice));
        if (!s.tile_recs.empty()) {
            SPCIES_HIP_CHECK(hipMalloc((void **)&s.d_recs, s.tile_recs.size() * sizeof(tile::Rec)));
            SPCIES_HIP_CHECK(hipMemcpy(s.d_recs, s.tile_recs.data(), s.tile_recs.size() * sizeof(tile::Rec), hipMemcpyHostToDevice));
        }
        return 0;
    }
    std::vector<const std::vector<double> *> arrs = {&a.AB, &a.Alpha, &a.Beta, &a.Hi, &a.Hi_0, &a.Hi_N,
                                                     &a.Q,  &a.R,     &a.T,    &a.LB, &a.UB};
    if (a.ellip)
        for (auto *v : {&a.P, &a.P_half, &a.Pinv_half, &a.c_ell, &a.LBz, &a.UBz, &a.LBu0, &a.UBu0}) arrs.push_back(v);
    if (a.gen)
        for (auto *v : {&a.LBz, &a.UBz, &a.LBu0, &a.UBu0, &a.LBN, &a.UBN, &a.rho_0, &a.rho_v, &a.rho_N, &a.rho_i_0, &a.rho_i_v,
                        &a.rho_i_N})
            arrs.push_back(v);
    if (s.method == SPCIES_FISTA) arrs = {&a.AB, &a.Alpha, &a.Beta, &a.Q, &a.R, &s.QRi, &s.Td, &s.Ti, &a.LB, &a.UB};
    if (s.method == SPCIES_EADMM)
        arrs = {&s.e_rho, &s.e_rho0, &s.e_rhos, &a.LB,    &a.UB,   &s.e_LB0, &s.e_UB0, &s.e_LBs, &s.e_UBs,
                &a.AB,    &a.T,      &s.e_S,    &a.Alpha, &a.Beta, &s.e_H1i, &s.e_W2,  &s.e_H3i,
                &s.e_Qbi, &s.e_Qmi,  &s.e_Rbi,  &s.e_Rmi, &s.e_ABbi, &s.e_ABmi};  // (general Q, R: the last six; empty otherwise)
    std::vector<double> flat;
    std::vector<size_t> offs;
    for (auto *v : arrs) {
        offs.push_back(flat.size());
        flat.insert(flat.end(), v->begin(), v->end());
        while (flat.size() % 8) flat.push_back(0.0);  // 64-byte aligned sub-arrays
    }
    SPCIES_HIP_CHECK(hipMalloc((void **)&s.d_consts, flat.size() * sizeof(double)));
    SPCIES_HIP_CHECK(hipMemcpy(s.d_consts, flat.data(), flat.size() * sizeof(double), hipMemcpyHostToDevice));
    if (s.method == SPCIES_EADMM) {
        s.edev = EadmmDev{(int)offs[0],  (int)offs[1],  (int)offs[2],  (int)offs[3],  (int)offs[4],  (int)offs[5],
                          (int)offs[6],  (int)offs[7],  (int)offs[8],  (int)offs[9],  (int)offs[10], (int)offs[11],
                          (int)offs[12], (int)offs[13], (int)offs[14], (int)offs[15], (int)offs[16], a.N, a.k_max, a.tol};
        s.edev.Q_bi = (int)offs[17]; s.edev.Q_mi = (int)offs[18]; s.edev.R_bi = (int)offs[19]; s.edev.R_mi = (int)offs[20];
        s.edev.AB_bi = (int)offs[21]; s.edev.AB_mi = (int)offs[22];
        return 0;
    }
    if (s.method == SPCIES_FISTA) {
        s.fdev = FistaDev{(int)offs[0], (int)offs[1], (int)offs[2], (int)offs[3], (int)offs[4], (int)offs[5],
                          (int)offs[6], (int)offs[7], (int)offs[8], (int)offs[9], a.N, a.k_max, a.tol};
        return 0;
    }
    s.dev = AdmmDev{(int)offs[0], (int)offs[1], (int)offs[2], (int)offs[3], (int)offs[4], (int)offs[5],
                    (int)offs[6], (int)offs[7], (int)offs[8], (int)offs[9], (int)offs[10],
                    a.N,          a.k_max,      a.tol,        a.rho,        a.rho_i};
    if (a.ellip) {
        s.dev.P = (int)offs[11]; s.dev.P_half = (int)offs[12]; s.dev.Pinv_half = (int)offs[13]; s.dev.c_ell = (int)offs[14];
        s.dev.LBz = (int)offs[15]; s.dev.UBz = (int)offs[16]; s.dev.LBu0 = (int)offs[17]; s.dev.UBu0 = (int)offs[18];
        s.dev.r_ell = a.r_ell;
    }
    if (a.gen) {
        const size_t o = a.ellip ? 19 : 11;  // (ellipMPC with a vector rho: after the ellipsoid's arrays)
        s.dev.LBz = (int)offs[o]; s.dev.UBz = (int)offs[o + 1]; s.dev.LBu0 = (int)offs[o + 2]; s.dev.UBu0 = (int)offs[o + 3];
        s.dev.LBN = (int)offs[o + 4]; s.dev.UBN = (int)offs[o + 5]; s.dev.rho_0 = (int)offs[o + 6]; s.dev.rho_v = (int)offs[o + 7];
        s.dev.rho_N = (int)offs[o + 8]; s.dev.rho_i_0 = (int)offs[o + 9]; s.dev.rho_i_v = (int)offs[o + 10];
        s.dev.rho_i_N = (int)offs[o + 11];
    }
    return 0;
}

static bool stream_shape_built(int n, int m) {
    return (m == 2 && (n == 6 || n == 12 || n == 20)) || (n == 4 && m == 1) || (n == 8 && m == 2) ||
           (n == 2 && m == 1);
}

// compile the MFMA4 kernel for this controller's shape (tens of seconds, once)
static int ensure_mfma4_rtc(Solver &s) {
    if (s.mfma4.ok) return 0;
    if (!s.mfma4.needs_rtc) return fail(SPCIES_HIP_ENOSUP, "MFMA4 variant not available for this shape: %s", s.mfma4.why.c_str());
    const Mfma4Layout &L = s.mfma4.lay;
    int rc = rtc::compile_mfma4(s.mfma4_rtc, L.N, L.KX, L.KS, L.terminal, s.mfma4.unit);
    if (rc) return rc;
    s.mfma4.ok = true;
    s.mfma4.why.clear();
    return 0;
}

// Time-varying lax/equ ADMM, variant MFMA4R (admm_tvr.hpp: one wavefront per instance, the instance's factors in its registers): the
// plan is built at create time (hiprtc for horizons without a build-time kernel); SPCIES_HIP_TVR=0 switches the variant off
static bool tvr_ok(const Solver &s) { return s.tv && (s.method == SPCIES_ADMM || s.method == SPCIES_FISTA) && s.tvrp.ok; }

static int resolve_variant(const Solver &s) {
    if (s.variant != SPCIES_VARIANT_AUTO) return s.variant;
    if (s.host.ellip) return s.bsp.ok ? SPCIES_VARIANT_BSP : SPCIES_VARIANT_STREAM;
    // (time-varying ADMM / FISTA: MFMA4R - factors in registers, admm_tvr.hpp - where its kernel is built.  The north star's sketch - one
    // wavefront per instance, factors in LDS, the reference's recurrences - was built in round 3, measured at 0.157 M solves/s against STREAM's
    // 0.28 M and removed in round 5: DESIGN.md 4.2f keeps the numbers.)
    if (s.tv) return tvr_ok(s) ? SPCIES_VARIANT_MFMA4R : SPCIES_VARIANT_STREAM;
    // (HMPC, split or not: the hand-written FUSED kernel, else the reference-order kernels of the library itself - TILE / STREAM.  The
    // rocBLAS variant GEMM is a cross-check that runs only when asked for by name: AUTO never hands a product to a library, never
    // loads librocblas and stays asynchronous / graph-capturable on every path.)
    if (s.is_hdense()) return s.hfused.ok ? SPCIES_VARIANT_FUSED : SPCIES_VARIANT_STREAM;
    if (s.is_cs()) return s.csf.ok ? SPCIES_VARIANT_FUSED : (s.tdev.lpi ? SPCIES_VARIANT_TILE : SPCIES_VARIANT_STREAM);
    if (s.is_hmpc() && s.hfused.ok) return SPCIES_VARIANT_FUSED;
    if (s.is_soc() && !s.is_hmpc() && s.bsp.ok) return SPCIES_VARIANT_BSP;
    if (s.is_soc()) return s.tdev.lpi ? SPCIES_VARIANT_TILE : SPCIES_VARIANT_STREAM;
    if (s.method == SPCIES_FISTA && s.frplan.ok) return SPCIES_VARIANT_MFMA4R;
    if (s.method == SPCIES_EADMM && s.erplan.ok) return SPCIES_VARIANT_MFMA4R;
    if (s.method == SPCIES_FISTA || s.method == SPCIES_EADMM) return s.g4plan.ok ? SPCIES_VARIANT_MFMA4G : SPCIES_VARIANT_STREAM;
    if (s.host.gen && s.bsp.ok && (s.formulation == SPCIES_LAXMPC || s.formulation == SPCIES_EQUMPC)) return SPCIES_VARIANT_BSP;
    if (s.mfma4.ok) return SPCIES_VARIANT_MFMA4;
    if (s.mfma.ok) return SPCIES_VARIANT_MFMA;
    if (s.arplan.ok) return SPCIES_VARIANT_MFMA4R;  // (shapes MFMA4 cannot hold: admm_r.hpp)
    if (s.g4plan.ok) return SPCIES_VARIANT_MFMA4G;
    return SPCIES_VARIANT_STREAM;
}

static size_t stream_scratch_bytes(const Solver &s, long B, bool want_sol) {
    long Bp = (B + 63) / 64 * 64;
    size_t rows = 2 * (size_t)s.host.dim() + (size_t)s.host.N * s.host.n + (want_sol ? (size_t)s.host.dim() : 0);
    if (s.method == SPCIES_FISTA) rows = 3 * (size_t)s.host.N * s.host.n + (want_sol ? (size_t)s.host.dim() : 0);
    if (s.method == SPCIES_EADMM)
        rows = (size_t)(3 * s.host.N + 5) * (s.host.n + s.host.m) + (size_t)s.host.N * s.host.n;
    if (s.is_soc()) rows = 4 * (size_t)(s.sdev.dim + s.sdev.n_s) + 2 * (size_t)(s.sdev.n_eq + s.sdev.n_s) + (size_t)s.sdev.dim;
    if (s.is_cs()) rows = 4 * (size_t)s.cdev.dim + (size_t)s.cdev.nrow + 2 * (size_t)(s.cdev.n + s.cdev.m);
    if (s.is_hmpc())
        rows = 2 * (size_t)(s.hdev.dim + s.hdev.n_s) + (size_t)s.hdev.nrow_M + (size_t)(s.hdev.n_eq + s.hdev.n_s) + (size_t)s.hdev.dim;
    return rows * (size_t)Bp * sizeof(double);
}

static int ensure_scratch(Solver &s, size_t need) {
    if (need <= s.scratch_bytes) return 0;
    if (s.d_scratch) SPCIES_HIP_CHECK(hipFree(s.d_scratch));
    s.d_scratch = nullptr;
    s.scratch_bytes = 0;
    SPCIES_HIP_CHECK(hipMalloc((void **)&s.d_scratch, need));
    s.scratch_bytes = need;
    // Scratch is never zeroed: a kernel that reads a row before it wrote it gets whatever the allocation held.  SPCIES_HIP_POISON=1
    // (test runs) fills it with NaNs so that such a read shows in the results instead of depending on the memory's history.
    if (const char *ev = getenv("SPCIES_HIP_POISON"))
        if (ev[0] == '1') {
            SPCIES_HIP_CHECK(hipMemset(s.d_scratch, 0xFF, need));
            SPCIES_HIP_CHECK(hipDeviceSynchronize());
        }
    return 0;
}

template <int n, int m, bool ELLIP = false, bool GEN = false>
static int launch_stream_nm(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B,
                            double *u, int *k, int *e, double *z, double *v, double *lam, hipStream_t st) {
    const bool want_sol = (z || v || lam);
    const long Bp = (B + 63) / 64 * 64;
    const size_t dim = (size_t)s.host.dim();
    double *V = s.d_scratch;
    double *LAM = V + dim * Bp;
    double *Y = LAM + dim * Bp;
    double *ZS = want_sol ? Y + (size_t)s.host.N * n * Bp : nullptr;
    dim3 grid((unsigned)(Bp / 64)), block(64);
    if constexpr (ELLIP) {
        if (s.host.gen)  // vector rho
            hipLaunchKernelGGL((admm_stream_kernel<n, m, true, true, false, true, true>), grid, block, 0, st, s.dev, s.d_consts, x0, xr,
                               ur, ref_stride, B, Bp, V, LAM, Y, ZS, u, k, e, nullptr);
        else
            hipLaunchKernelGGL((admm_stream_kernel<n, m, true, true, false, true>), grid, block, 0, st, s.dev, s.d_consts, x0, xr, ur,
                               ref_stride, B, Bp, V, LAM, Y, ZS, u, k, e, nullptr);
    } else if constexpr (GEN) {
        if (s.host.terminal)
            hipLaunchKernelGGL((admm_stream_kernel<n, m, true, true, false, false, true>), grid, block, 0, st, s.dev, s.d_consts, x0,
                               xr, ur, ref_stride, B, Bp, V, LAM, Y, ZS, u, k, e, nullptr);
        else
            hipLaunchKernelGGL((admm_stream_kernel<n, m, false, true, false, false, true>), grid, block, 0, st, s.dev, s.d_consts, x0,
                               xr, ur, ref_stride, B, Bp, V, LAM, Y, ZS, u, k, e, nullptr);
    } else if (s.host.terminal)
        hipLaunchKernelGGL((admm_stream_kernel<n, m, true, true>), grid, block, 0, st, s.dev, s.d_consts, x0, xr, ur,
                           ref_stride, B, Bp, V, LAM, Y, ZS, u, k, e);
    else
        hipLaunchKernelGGL((admm_stream_kernel<n, m, false, true>), grid, block, 0, st, s.dev, s.d_consts, x0, xr, ur,
                           ref_stride, B, Bp, V, LAM, Y, ZS, u, k, e);
    SPCIES_HIP_CHECK(hipGetLastError());
    if (want_sol) {
        dim3 tg((unsigned)(Bp / 64), (unsigned)((dim + 63) / 64));
        if (z) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, ZS, Bp, B, (int)dim, z);
        if (v) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, V, Bp, B, (int)dim, v);
        if (lam) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, LAM, Bp, B, (int)dim, lam);
        SPCIES_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

template <int n, int m>
static int launch_fista_nm(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B,
                           double *u, int *k, int *e, double *z, double *lam, hipStream_t st) {
    const bool want_sol = (z || lam);
    const long Bp = (B + 63) / 64 * 64;
    const size_t Nn = (size_t)s.host.N * n, dim = (size_t)s.host.dim();
    double *Y = s.d_scratch, *LAM = Y + Nn * Bp, *DL = LAM + Nn * Bp;
    double *ZS = want_sol ? DL + Nn * Bp : nullptr;
    dim3 grid((unsigned)(Bp / 64)), block(64);
    if (s.host.terminal)
        hipLaunchKernelGGL((fista_stream_kernel<n, m, true, true>), grid, block, 0, st, s.fdev, s.d_consts, x0, xr, ur,
                           ref_stride, B, Bp, Y, LAM, DL, ZS, u, k, e);
    else
        hipLaunchKernelGGL((fista_stream_kernel<n, m, false, true>), grid, block, 0, st, s.fdev, s.d_consts, x0, xr, ur,
                           ref_stride, B, Bp, Y, LAM, DL, ZS, u, k, e);
    SPCIES_HIP_CHECK(hipGetLastError());
    if (z) {
        dim3 tg((unsigned)(Bp / 64), (unsigned)((dim + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, ZS, Bp, B, (int)dim, z);
    }
    if (lam) {  // the reference returns y as sol.lambda (code_laxMPC_FISTA_C.c:439-445)
        dim3 tg((unsigned)(Bp / 64), (unsigned)((Nn + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, Y, Bp, B, (int)Nn, lam);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

// Time-varying lax/equ FISTA: update phase (per-instance banded Cholesky, fista_tv_update_kernel) + the STREAM iteration reading
// the instance's own constants.  Large batches are split so that one launch's constants stay below the 4 GB a buffer
// resource can address.
template <int n, int m>
static int launch_fista_tv_nm(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, const double *model,
                              int model_stride, long B, double *u, int *k, int *e, double *z, double *lam, hipStream_t st) {
    const int N = s.host.N;
    const FistaTvLayout tl = fista_tv_layout(n, m, N);
    const bool want_sol = (z || lam);
    const size_t Nn = (size_t)N * n, dim = (size_t)s.host.dim();
    const bool regs = resolve_variant(s) == SPCIES_VARIANT_MFMA4R;  // one wavefront per instance, factors in registers (admm_tvr_kernel.inc)
    const size_t rows_stream = regs ? 0 : 3 * Nn + (want_sol ? dim : 0);
    const size_t rows_tv = (size_t)tl.rows + (regs ? (size_t)N * n * n : 0);  // (MFMA4R: the explicit inverses behind the factors)
    long chunk = (long)((3900ull << 20) / (rows_tv * 8)) / 64 * 64;
    if (chunk > B) chunk = (B + 63) / 64 * 64;
    int rc = ensure_scratch(s, (rows_stream + rows_tv) * (size_t)chunk * sizeof(double));
    if (rc) return rc;
    const int num_cu = s.num_cu;  // (queried once at create time)
    for (long b0 = 0; b0 < B; b0 += chunk) {
        const long Bc = std::min(chunk, B - b0), Bp = (Bc + 63) / 64 * 64;
        double *Y = s.d_scratch, *LAM = Y + Nn * Bp, *DL = LAM + Nn * Bp;
        double *ZS = want_sol ? DL + Nn * Bp : nullptr;
        double *TVS = s.d_scratch + rows_stream * Bp;
        const double *xrc = ref_stride ? xr + b0 * n : xr, *urc = ref_stride ? ur + b0 * m : ur;
        const double *mc = model_stride ? model + b0 * (long)model_stride : model;
        dim3 grid((unsigned)(Bp / 64)), block(64);
        if (regs) {  // (the update phase that writes the explicit inverses too: instantiated in admm_tvr.hip)
            rc = tvr::launch_update(s.tvrp, 0.0, s.d_consts + s.fdev.Ti, mc, (long)model_stride, Bc, Bp, TVS, st);
            if (rc) return rc;
            tvr::Args ta{s.host.k_max, ref_stride, 0.0, s.host.tol, Bc, Bp};
            rc = tvr::launch_fista(s.tvrp, want_sol, ta, s.d_consts + s.fdev.T, s.d_consts + s.fdev.Ti, TVS, x0 + b0 * n, xrc, urc, u + b0 * m, k + b0, e + b0,
                                   z ? z + b0 * dim : nullptr, lam ? lam + b0 * Nn : nullptr, num_cu, st);
            if (rc) return rc;
            continue;
        }
        if (s.host.terminal) {
            hipLaunchKernelGGL((fista_tv_update_kernel<n, m, true>), grid, block, 0, st, N, s.d_consts + s.fdev.Ti, mc, (long)model_stride,
                               Bc, Bp, TVS);
            hipLaunchKernelGGL((fista_stream_kernel<n, m, true, true, true>), grid, block, 0, st, s.fdev, s.d_consts, x0 + b0 * n, xrc,
                               urc, ref_stride, Bc, Bp, Y, LAM, DL, ZS, u + b0 * m, k + b0, e + b0, TVS);
        } else {
            hipLaunchKernelGGL((fista_tv_update_kernel<n, m, false>), grid, block, 0, st, N, s.d_consts + s.fdev.Ti, mc, (long)model_stride,
                               Bc, Bp, TVS);
            hipLaunchKernelGGL((fista_stream_kernel<n, m, false, true, true>), grid, block, 0, st, s.fdev, s.d_consts, x0 + b0 * n, xrc,
                               urc, ref_stride, Bc, Bp, Y, LAM, DL, ZS, u + b0 * m, k + b0, e + b0, TVS);
        }
        SPCIES_HIP_CHECK(hipGetLastError());
        if (z) {
            dim3 tg((unsigned)(Bp / 64), (unsigned)((dim + 63) / 64));
            hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, ZS, Bp, Bc, (int)dim, z + b0 * dim);
        }
        if (lam) {
            dim3 tg((unsigned)(Bp / 64), (unsigned)((Nn + 63) / 64));
            hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, Y, Bp, Bc, (int)Nn, lam + b0 * Nn);
        }
        SPCIES_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

static int launch_fista(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B,
                        double *u, int *k, int *e, double *z, double *lam, hipStream_t st) {
    const int n = s.host.n, m = s.host.m;
#define SPCIES_CASE(NN, MM) \
    if (n == NN && m == MM) return launch_fista_nm<NN, MM>(s, x0, xr, ur, ref_stride, B, u, k, e, z, lam, st);
    SPCIES_CASE(6, 2)
    SPCIES_CASE(12, 2)
    SPCIES_CASE(20, 2)
    SPCIES_CASE(8, 2)
    SPCIES_CASE(4, 1)
    SPCIES_CASE(2, 1)
#undef SPCIES_CASE
    return fail(SPCIES_HIP_ENOSUP, "FISTA STREAM variant not instantiated for n=%d m=%d", n, m);
}

template <int n, int m>
static int launch_eadmm_nm(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B,
                           double *u, int *k, int *e, double *z1, double *z2, double *z3, double *lam, hipStream_t st) {
    const long Bp = (B + 63) / 64 * 64;
    const int nm = n + m, N = s.host.N;
    const size_t dz = (size_t)(N + 1) * nm;
    double *Z1 = s.d_scratch, *Z3 = Z1 + dz * Bp, *LAM = Z3 + dz * Bp, *MU = LAM + (size_t)(N + 3) * nm * Bp;
    dim3 grid((unsigned)(Bp / 64)), block(64);
    hipLaunchKernelGGL((eadmm_stream_kernel<n, m>), grid, block, 0, st, s.edev, s.d_consts, x0, xr, ur, ref_stride, B, Bp,
                       Z1, Z3, LAM, MU, z2, u, k, e);
    SPCIES_HIP_CHECK(hipGetLastError());
    dim3 tg((unsigned)(Bp / 64), (unsigned)((dz + 63) / 64));
    if (z1) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, Z1, Bp, B, (int)dz, z1);
    if (z3) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, Z3, Bp, B, (int)dz, z3);
    if (lam) {
        const long tot = B * (long)(N + 3) * nm;
        hipLaunchKernelGGL(eadmm_pack_lambda_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, LAM, Bp, B, N,
                           n, nm, lam);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

static bool eadmm_stream_shape_built(int n, int m) { return m == 2 && (n == 6 || n == 12 || n == 20); }

static int launch_eadmm(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B,
                        double *u, int *k, int *e, double *z1, double *z2, double *z3, double *lam, hipStream_t st) {
    const int n = s.host.n, m = s.host.m;
#define SPCIES_CASE(NN, MM) \
    if (n == NN && m == MM) return launch_eadmm_nm<NN, MM>(s, x0, xr, ur, ref_stride, B, u, k, e, z1, z2, z3, lam, st);
    SPCIES_CASE(6, 2)
    SPCIES_CASE(12, 2)
    SPCIES_CASE(20, 2)
#undef SPCIES_CASE
    return fail(SPCIES_HIP_ENOSUP, "EADMM STREAM variant not instantiated for n=%d m=%d", n, m);
}

// Time-varying lax/equ ADMM: update phase (per-instance banded Cholesky) + the STREAM iteration reading the
// instance's own constants.  Large batches are split so that one launch's constants stay below the 4 GB a
// buffer resource can address.
template <int n, int m>
static int launch_tv_nm(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, const double *model,
                        int model_stride, long B, double *u, int *k, int *e, double *z, double *v, double *lam,
                        hipStream_t st) {
    const int N = s.host.N;
    const TvLayout tl = tv_layout(n, m, N);
    const bool want_sol = (z || v || lam);
    const size_t dim = (size_t)s.host.dim();
    const bool regs = resolve_variant(s) == SPCIES_VARIANT_MFMA4R;               // MFMA4R: one wavefront per instance, factors in registers (admm_tvr.hpp)
    const size_t rows_stream = regs ? 0 : 2 * dim + (size_t)N * n + (want_sol ? dim : 0);
    const size_t rows_tv = regs ? (size_t)tl.rows_all : (size_t)tl.rows;  // (MFMA4R: the explicit inverses behind the factors)
    long chunk = (long)((3900ull << 20) / (rows_tv * 8)) / 64 * 64;
    if (chunk > B) chunk = (B + 63) / 64 * 64;
    int rc = ensure_scratch(s, (rows_stream + rows_tv) * (size_t)chunk * sizeof(double));
    if (rc) return rc;
    const int num_cu = s.num_cu;  // (queried once at create time)
    for (long b0 = 0; b0 < B; b0 += chunk) {
        const long Bc = std::min(chunk, B - b0), Bp = (Bc + 63) / 64 * 64;
        double *V = s.d_scratch, *LAM = V + dim * Bp, *Y = LAM + dim * Bp;
        double *ZS = want_sol ? Y + (size_t)N * n * Bp : nullptr;
        double *TVS = s.d_scratch + rows_stream * Bp;
        const double *xrc = ref_stride ? xr + b0 * n : xr, *urc = ref_stride ? ur + b0 * m : ur;
        const double *mc = model_stride ? model + b0 * (long)model_stride : model;
        dim3 grid((unsigned)(Bp / 64)), block(64);
        if (regs) {  // update phase (the reference's factorisation, one lane per instance, and the explicit inverses: admm_tvr.hip), then one wavefront per instance
            rc = tvr::launch_update(s.tvrp, s.host.rho, s.d_consts + s.dev.Hi_N, mc, (long)model_stride, Bc, Bp, TVS, st);
            if (rc) return rc;
            tvr::Args ta{s.host.k_max, ref_stride, s.host.rho, s.host.tol, Bc, Bp};
            rc = tvr::launch(s.tvrp, want_sol, ta, s.d_consts + s.dev.Hi_N, s.d_consts + s.dev.T, TVS, x0 + b0 * n, xrc, urc, u + b0 * m,
                             k + b0, e + b0, z ? z + b0 * dim : nullptr, v ? v + b0 * dim : nullptr, lam ? lam + b0 * dim : nullptr, num_cu, st);
            if (rc) return rc;
            continue;
        }
        if (s.host.terminal) {
            hipLaunchKernelGGL((admm_tv_update_kernel<n, m, true>), grid, block, 0, st, N, s.host.rho, s.d_consts + s.dev.Hi_N, mc,
                               (long)model_stride, Bc, Bp, TVS);
            hipLaunchKernelGGL((admm_stream_kernel<n, m, true, true, true>), grid, block, 0, st, s.dev, s.d_consts, x0 + b0 * n,
                               xrc, urc, ref_stride, Bc, Bp, V, LAM, Y, ZS, u + b0 * m, k + b0, e + b0, TVS);
        } else {
            hipLaunchKernelGGL((admm_tv_update_kernel<n, m, false>), grid, block, 0, st, N, s.host.rho, s.d_consts + s.dev.Hi_N,
                               mc, (long)model_stride, Bc, Bp, TVS);
            hipLaunchKernelGGL((admm_stream_kernel<n, m, false, true, true>), grid, block, 0, st, s.dev, s.d_consts, x0 + b0 * n,
                               xrc, urc, ref_stride, Bc, Bp, V, LAM, Y, ZS, u + b0 * m, k + b0, e + b0, TVS);
        }
        SPCIES_HIP_CHECK(hipGetLastError());
        if (want_sol) {
            dim3 tg((unsigned)(Bp / 64), (unsigned)((dim + 63) / 64));
            if (z) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, ZS, Bp, Bc, (int)dim, z + b0 * dim);
            if (v) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, V, Bp, Bc, (int)dim, v + b0 * dim);
            if (lam) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, LAM, Bp, Bc, (int)dim, lam + b0 * dim);
            SPCIES_HIP_CHECK(hipGetLastError());
        }
    }
    return 0;
}

// The plain banded solvers (lax / equ ADMM with scalar rho and constant bounds, lax / equ FISTA, MPCT EADMM with diagonal Q, R; not time-varying):
// their STREAM kernel - the bit-exact variant - exists for EVERY plant size: the build-time instantiations for the benchmark shapes, hiprtc for
// any other (admm_stream_kernel.inc / fista_stream_kernel.inc / eadmm_stream_kernel.inc are the text of both).
static bool stream_rtc_applies(const Solver &s) {
    if (s.is_soc() || s.is_cs() || s.is_hmpc() || s.is_hdense()) return false;
    // (round 5: the template's other switches too - time-varying model with its update phase, vector rho / stage-wise bounds, the ellipMPC
    // terminal block: every option of the banded ADMM / FISTA solvers has its bit-exact kernel at any plant size)
    if (s.method == SPCIES_ADMM) return s.formulation == SPCIES_LAXMPC || s.formulation == SPCIES_EQUMPC || s.host.ellip;
    if (s.method == SPCIES_FISTA) return !s.host.gen && !s.host.ellip && (s.formulation == SPCIES_LAXMPC || s.formulation == SPCIES_EQUMPC);
    if (s.tv || s.host.gen || s.host.ellip) return false;
    if (s.method == SPCIES_EADMM) return true;  // (general Q, R: that branch's STREAM kernel is always the run-time specialised one)
    return false;
}
static int ensure_stream_rtc(Solver &s) {
    if (s.srtc.ok) return 0;
    if (s.srtc.tried) return fail(SPCIES_HIP_ENOSUP, "STREAM variant (run-time specialised for n=%d m=%d) unavailable: %s", s.host.n, s.host.m, s.srtc.why.c_str());
    s.srtc.tried = true;
    if (const char *ev = getenv("SPCIES_HIP_RTC"))
        if (ev[0] == '0') {
            s.srtc.why = "no build-time kernel for this (n, m) and SPCIES_HIP_RTC=0";
            return fail(SPCIES_HIP_ENOSUP, "STREAM variant unavailable: %s", s.srtc.why.c_str());
        }
    char name[200], uname[200] = "";
    std::string src = std::string(kAdmmDevSrc) + "\n" + kTvUpdateSrc + "\n" + kAdmmStreamSrc;
    const char *fname = "spcies_admm_stream_rtc.hip";
    const char *term = s.host.terminal ? "true" : "false", *tvs = s.tv ? "true" : "false";
    if (s.method == SPCIES_FISTA) {
        snprintf(name, sizeof(name), "spcies::fista_stream_kernel<%d, %d, %s, true, %s>", s.host.n, s.host.m, term, tvs);
        if (s.tv) snprintf(uname, sizeof(uname), "spcies::fista_tv_update_kernel<%d, %d, %s>", s.host.n, s.host.m, term);
        src += std::string("\n") + kFistaStreamSrc;
        fname = "spcies_fista_stream_rtc.hip";
    } else if (s.method == SPCIES_ADMM && (s.tv || s.host.gen || s.host.ellip)) {
        // <n, m, TERMINAL, EXACT, TV, ELLIP, GEN>: the instantiations launch_stream_nm / launch_tv_nm pick at build time
        snprintf(name, sizeof(name), "spcies::admm_stream_kernel<%d, %d, %s, true, %s, %s, %s>", s.host.n, s.host.m, s.host.ellip ? "true" : term, tvs,
                 s.host.ellip ? "true" : "false", s.host.gen ? "true" : "false");
        if (s.tv) snprintf(uname, sizeof(uname), "spcies::admm_tv_update_kernel<%d, %d, %s>", s.host.n, s.host.m, term);
    } else if (s.method == SPCIES_EADMM) {
        snprintf(name, sizeof(name), "spcies::eadmm_stream_kernel<%d, %d, %s>", s.host.n, s.host.m, s.e_general ? "true" : "false");
        src += std::string("\n") + kEadmmStreamSrc;
        fname = "spcies_eadmm_stream_rtc.hip";
    } else {
        snprintf(name, sizeof(name), "spcies::admm_stream_kernel<%d, %d, %s, true>", s.host.n, s.host.m, s.host.terminal ? "true" : "false");
    }
    hipModule_t mod = nullptr;
    hipFunction_t fns[2] = {nullptr, nullptr};
    std::vector<std::string> names = {std::string(name)};
    if (uname[0]) names.push_back(uname);
    int rc = rtc::compile_module(src.c_str(), fname, names, {}, &mod, fns);
    if (rc) {
        s.srtc.why = spcies_hip_last_error();
        return rc;
    }
    s.srtc.mod = mod;
    s.srtc.fn = fns[0];
    s.srtc.fn_update = fns[1];
    s.srtc.ok = true;
    return 0;
}
// ... FISTA (launch_fista_nm's buffers) and EADMM (launch_eadmm_nm's) through the run-time specialised kernel
static int launch_fista_rtc(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u, int *k, int *e,
                            double *z, double *lam, hipStream_t st) {
    int rc = ensure_stream_rtc(s);
    if (rc) return rc;
    const bool want_sol = (z || lam);
    long Bp = (B + 63) / 64 * 64;
    const size_t Nn = (size_t)s.host.N * s.host.n, dim = (size_t)s.host.dim();
    double *Y = s.d_scratch, *LAM = Y + Nn * Bp, *DL = LAM + Nn * Bp;
    double *ZS = want_sol ? DL + Nn * Bp : nullptr;
    const double *C = s.d_consts, *TVS = nullptr;
    FistaDev dev = s.fdev;
    void *params[] = {&dev, &C, &x0, &xr, &ur, &ref_stride, &B, &Bp, &Y, &LAM, &DL, &ZS, &u, &k, &e, &TVS};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel(s.srtc.fn, (unsigned)(Bp / 64), 1, 1, 64, 1, 1, 0, st, params, nullptr));
    if (z) {
        dim3 tg((unsigned)(Bp / 64), (unsigned)((dim + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, ZS, Bp, B, (int)dim, z);
    }
    if (lam) {  // the reference returns y as sol.lambda (code_laxMPC_FISTA_C.c:439-445)
        dim3 tg((unsigned)(Bp / 64), (unsigned)((Nn + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, Y, Bp, B, (int)Nn, lam);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}
static int launch_eadmm_rtc(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u, int *k, int *e,
                            double *z1, double *z2, double *z3, double *lam, hipStream_t st) {
    int rc = ensure_stream_rtc(s);
    if (rc) return rc;
    long Bp = (B + 63) / 64 * 64;
    const int n = s.host.n, nm = s.host.n + s.host.m, N = s.host.N;
    const size_t dz = (size_t)(N + 1) * nm;
    double *Z1 = s.d_scratch, *Z3 = Z1 + dz * Bp, *LAM = Z3 + dz * Bp, *MU = LAM + (size_t)(N + 3) * nm * Bp;
    const double *C = s.d_consts;
    EadmmDev dev = s.edev;
    void *params[] = {&dev, &C, &x0, &xr, &ur, &ref_stride, &B, &Bp, &Z1, &Z3, &LAM, &MU, &z2, &u, &k, &e};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel(s.srtc.fn, (unsigned)(Bp / 64), 1, 1, 64, 1, 1, 0, st, params, nullptr));
    dim3 tg((unsigned)(Bp / 64), (unsigned)((dz + 63) / 64));
    if (z1) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, Z1, Bp, B, (int)dz, z1);
    if (z3) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, Z3, Bp, B, (int)dz, z3);
    if (lam) {
        const long tot = B * (long)(N + 3) * nm;
        hipLaunchKernelGGL(eadmm_pack_lambda_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, LAM, Bp, B, N, n, nm, lam);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}
static int launch_stream_rtc(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u, int *k, int *e,
                             double *z, double *v, double *lam, hipStream_t st) {
    int rc = ensure_stream_rtc(s);
    if (rc) return rc;
    const bool want_sol = (z || v || lam);
    long Bp = (B + 63) / 64 * 64;
    const size_t dim = (size_t)s.host.dim();
    double *V = s.d_scratch, *LAM = V + dim * Bp, *Y = LAM + dim * Bp;
    double *ZS = want_sol ? Y + (size_t)s.host.N * s.host.n * Bp : nullptr;
    const double *C = s.d_consts, *TVS = nullptr;
    AdmmDev dev = s.dev;
    void *params[] = {&dev, &C, &x0, &xr, &ur, &ref_stride, &B, &Bp, &V, &LAM, &Y, &ZS, &u, &k, &e, &TVS};
    SPCIES_HIP_CHECK(hipModuleLaunchKernel(s.srtc.fn, (unsigned)(Bp / 64), 1, 1, 64, 1, 1, 0, st, params, nullptr));
    if (want_sol) {
        dim3 tg((unsigned)(Bp / 64), (unsigned)((dim + 63) / 64));
        if (z) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, ZS, Bp, B, (int)dim, z);
        if (v) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, V, Bp, B, (int)dim, v);
        if (lam) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, LAM, Bp, B, (int)dim, lam);
        SPCIES_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

static int launch_stream(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B,
                         double *u, int *k, int *e, double *z, double *v, double *lam, hipStream_t st) {
    const int n = s.host.n, m = s.host.m;
#define SPCIES_CASE(NN, MM) \
    if (n == NN && m == MM) return launch_stream_nm<NN, MM>(s, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
    SPCIES_CASE(6, 2)
    SPCIES_CASE(12, 2)
    SPCIES_CASE(20, 2)
    SPCIES_CASE(8, 2)
    SPCIES_CASE(4, 1)
    SPCIES_CASE(2, 1)
#undef SPCIES_CASE
    if (stream_rtc_applies(s)) return launch_stream_rtc(s, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
    return fail(SPCIES_HIP_ENOSUP, "STREAM variant not instantiated for n=%d m=%d", n, m);
}

// Time-varying lax / equ ADMM and FISTA at an (n, m) without build-time kernels (STREAM and the update phase are instantiated for (6, 2) and
// (12, 2)): the MFMA4R path with EVERYTHING run-time specialised - update phase, inverses, solve (admm_tvr.hpp; tv_update_kernel.inc is
// the same text the build-time instantiations compile).  z, v, lam: ADMM's record; FISTA passes z and lam (v = NULL).
static int launch_tv_rtc(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, const double *model, int model_stride,
                         long B, double *u, int *k, int *e, double *z, double *v, double *lam, hipStream_t st) {
    const int n = s.host.n, m = s.host.m, N = s.host.N;
    const bool fista = s.method == SPCIES_FISTA;
    const bool want_sol = (z || v || lam);
    const size_t dim = (size_t)s.host.dim(), Nn = (size_t)N * n;
    const size_t rows_tv = fista ? (size_t)fista_tv_layout(n, m, N).rows + (size_t)N * n * n : (size_t)tv_layout(n, m, N).rows_all;
    long chunk = (long)((3900ull << 20) / (rows_tv * 8)) / 64 * 64;  // one launch's rows stay below the 4 GB a buffer resource addresses
    if (chunk > B) chunk = (B + 63) / 64 * 64;
    int rc = ensure_scratch(s, rows_tv * (size_t)chunk * sizeof(double));
    if (rc) return rc;
    const int num_cu = s.num_cu;  // (queried once at create time)
    for (long b0 = 0; b0 < B; b0 += chunk) {
        const long Bc = std::min(chunk, B - b0), Bp = (Bc + 63) / 64 * 64;
        double *TVS = s.d_scratch;
        const double *xrc = ref_stride ? xr + b0 * n : xr, *urc = ref_stride ? ur + b0 * m : ur;
        const double *mc = model_stride ? model + b0 * (long)model_stride : model;
        if (fista) {
            rc = tvr::launch_update(s.tvrp, 0.0, s.d_consts + s.fdev.Ti, mc, (long)model_stride, Bc, Bp, TVS, st);
            if (rc) return rc;
            tvr::Args ta{s.host.k_max, ref_stride, 0.0, s.host.tol, Bc, Bp};
            rc = tvr::launch_fista(s.tvrp, want_sol, ta, s.d_consts + s.fdev.T, s.d_consts + s.fdev.Ti, TVS, x0 + b0 * n, xrc, urc, u + b0 * m, k + b0, e + b0,
                                   z ? z + b0 * dim : nullptr, lam ? lam + b0 * Nn : nullptr, num_cu, st);
        } else {
            rc = tvr::launch_update(s.tvrp, s.host.rho, s.d_consts + s.dev.Hi_N, mc, (long)model_stride, Bc, Bp, TVS, st);
            if (rc) return rc;
            tvr::Args ta{s.host.k_max, ref_stride, s.host.rho, s.host.tol, Bc, Bp};
            rc = tvr::launch(s.tvrp, want_sol, ta, s.d_consts + s.dev.Hi_N, s.d_consts + s.dev.T, TVS, x0 + b0 * n, xrc, urc, u + b0 * m, k + b0, e + b0,
                             z ? z + b0 * dim : nullptr, v ? v + b0 * dim : nullptr, lam ? lam + b0 * dim : nullptr, num_cu, st);
        }
        if (rc) return rc;
    }
    return 0;
}

// Time-varying lax / equ ADMM and FISTA on STREAM at an (n, m) without build-time kernels: launch_tv_nm / launch_fista_tv_nm's buffers, the
// update phase and the iteration of ensure_stream_rtc's module (bit-exact like the build-time pair; plants past n = 16 take the rolled update
// phase of tv_update_kernel.inc).  AUTO lands here when the register-resident solver does not hold the controller (n + m > 16, or a horizon
// past the register file); STREAM by name lands here for any plant size.  z, v, lam: ADMM's record; FISTA passes z and lam (v = NULL).
static int launch_tv_stream_rtc(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, const double *model, int model_stride,
                                long B, double *u, int *k, int *e, double *z, double *v, double *lam, hipStream_t st) {
    int rc = ensure_stream_rtc(s);
    if (rc) return rc;
    const int n = s.host.n, m = s.host.m, N = s.host.N;
    const bool fista = s.method == SPCIES_FISTA;
    const bool want_sol = (z || v || lam);
    const size_t dim = (size_t)s.host.dim(), Nn = (size_t)N * n;
    const size_t rows_stream = fista ? 3 * Nn + (want_sol ? dim : 0) : 2 * dim + Nn + (want_sol ? dim : 0);
    const size_t rows_tv = fista ? (size_t)fista_tv_layout(n, m, N).rows : (size_t)tv_layout(n, m, N).rows;
    long chunk = (long)((3900ull << 20) / (rows_tv * 8)) / 64 * 64;  // one launch's rows stay below the 4 GB a buffer resource addresses
    if (chunk < 64) return fail(SPCIES_HIP_ENOSUP, "time-varying STREAM: one wavefront's factors exceed a buffer resource (n=%d N=%d)", n, N);
    if (chunk > B) chunk = (B + 63) / 64 * 64;
    rc = ensure_scratch(s, (rows_stream + rows_tv) * (size_t)chunk * sizeof(double));
    if (rc) return rc;
    for (long b0 = 0; b0 < B; b0 += chunk) {
        long Bc = std::min(chunk, B - b0), Bp = (Bc + 63) / 64 * 64;
        double *TVS = s.d_scratch + rows_stream * Bp;
        const double *x0c = x0 + b0 * n, *xrc = ref_stride ? xr + b0 * n : xr, *urc = ref_stride ? ur + b0 * m : ur;
        const double *mc = model_stride ? model + b0 * (long)model_stride : model;
        long mstride = model_stride;
        double *uc = u + b0 * m;
        int *kc = k + b0, *ec = e + b0;
        const double *C = s.d_consts;
        int Nv = N;
        const unsigned grid = (unsigned)(Bp / 64);
        if (fista) {
            const double *Ti = s.d_consts + s.fdev.Ti;
            void *up[] = {&Nv, &Ti, &mc, &mstride, &Bc, &Bp, &TVS};
            SPCIES_HIP_CHECK(hipModuleLaunchKernel(s.srtc.fn_update, grid, 1, 1, 64, 1, 1, 0, st, up, nullptr));
            double *Y = s.d_scratch, *LAM = Y + Nn * Bp, *DL = LAM + Nn * Bp;
            double *ZS = want_sol ? DL + Nn * Bp : nullptr;
            const double *TVSc = TVS;
            FistaDev dev = s.fdev;
            void *params[] = {&dev, &C, &x0c, &xrc, &urc, &ref_stride, &Bc, &Bp, &Y, &LAM, &DL, &ZS, &uc, &kc, &ec, &TVSc};
            SPCIES_HIP_CHECK(hipModuleLaunchKernel(s.srtc.fn, grid, 1, 1, 64, 1, 1, 0, st, params, nullptr));
            if (z) {
                dim3 tg((unsigned)(Bp / 64), (unsigned)((dim + 63) / 64));
                hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, ZS, Bp, Bc, (int)dim, z + b0 * dim);
            }
            if (lam) {
                dim3 tg((unsigned)(Bp / 64), (unsigned)((Nn + 63) / 64));
                hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, Y, Bp, Bc, (int)Nn, lam + b0 * Nn);
            }
        } else {
            double rho = s.host.rho;
            const double *HiN = s.d_consts + s.dev.Hi_N;
            void *up[] = {&Nv, &rho, &HiN, &mc, &mstride, &Bc, &Bp, &TVS};
            SPCIES_HIP_CHECK(hipModuleLaunchKernel(s.srtc.fn_update, grid, 1, 1, 64, 1, 1, 0, st, up, nullptr));
            double *V = s.d_scratch, *LAM = V + dim * Bp, *Y = LAM + dim * Bp;
            double *ZS = want_sol ? Y + Nn * Bp : nullptr;
            const double *TVSc = TVS;
            AdmmDev dev = s.dev;
            void *params[] = {&dev, &C, &x0c, &xrc, &urc, &ref_stride, &Bc, &Bp, &V, &LAM, &Y, &ZS, &uc, &kc, &ec, &TVSc};
            SPCIES_HIP_CHECK(hipModuleLaunchKernel(s.srtc.fn, grid, 1, 1, 64, 1, 1, 0, st, params, nullptr));
            if (want_sol) {
                dim3 tg((unsigned)(Bp / 64), (unsigned)((dim + 63) / 64));
                if (z) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, ZS, Bp, Bc, (int)dim, z + b0 * dim);
                if (v) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, V, Bp, Bc, (int)dim, v + b0 * dim);
                if (lam) hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, LAM, Bp, Bc, (int)dim, lam + b0 * dim);
            }
        }
        SPCIES_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// f[] = the solver's record fields in reference order (Solver::field_name), NULL entries are not produced
static int launch_soc(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, const double *r,
                      int r_stride, long B, double *u, int *k, int *e, double *const *f, hipStream_t st) {
    const long Bp = (B + 63) / 64 * 64;
    const SocDev &d = s.sdev;
    const long np = d.dim + d.n_s;
    double *S = s.d_scratch;
    hipLaunchKernelGGL(soc_stream_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, st, d, s.d_consts, s.d_idx, x0, xr, ur,
                       ref_stride, r, r_stride, B, Bp, S, u, k, e);
    SPCIES_HIP_CHECK(hipGetLastError());
    // record fields z, s, z_hat, s_hat, lambda, mu = row slices of PR, PH, DU
    const double *base[3] = {S, S + np * Bp, S + 2 * np * Bp};
    for (int i = 0; i < 6; i++) {
        if (!f[i]) continue;
        const int rows = (i % 2 == 0) ? d.dim : d.n_s;
        const double *src = base[i / 2] + ((i % 2 == 0) ? 0 : (long)d.dim * Bp);
        dim3 tg((unsigned)(Bp / 64), (unsigned)((rows + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, src, Bp, B, rows, f[i]);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

// MPCT ADMM cs: STREAM (record z, v, lambda = row slices Z, V, LAM) and TILE (V | LAM | Z per tile)
static int launch_cs_stream(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
                            int *k, int *e, double *const *f, hipStream_t st) {
    const long Bp = (B + 63) / 64 * 64;
    const CsDev &d = s.cdev;
    double *S = s.d_scratch;
    hipLaunchKernelGGL(cs_stream_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, st, d, s.d_consts, s.d_idx, x0, xr, ur, ref_stride,
                       B, Bp, S, u, k, e);
    SPCIES_HIP_CHECK(hipGetLastError());
    for (int i = 0; i < 3; i++) {
        if (!f[i]) continue;
        dim3 tg((unsigned)(Bp / 64), (unsigned)((d.dim + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, S + (long)i * d.dim * Bp, Bp, B, d.dim, f[i]);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

static int launch_cs_tile(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
                          int *k, int *e, double *const *f, hipStream_t st) {
    const int lpi = s.tdev.lpi, T = 64 / lpi;
    const CsDev &d = s.cdev;
    const long tiles = ((B + T - 1) / T + tile::WAVES - 1) / tile::WAVES * tile::WAVES;
    const long rows = 3L * d.dim + d.n + 2 * (d.n + d.m);
    double *S = s.d_scratch;
    const size_t shmem = s.tdev.lds_bytes * tile::WAVES;
    dim3 grid((unsigned)(tiles / tile::WAVES)), block(64 * tile::WAVES);
#define SPCIES_CS_LAUNCH(LPI)                                                                                              \
    do {                                                                                                                   \
        auto kern = tile::cs_tile_kernel<LPI>;                                                                             \
        if (shmem > 48 * 1024)                                                                                             \
            SPCIES_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
        hipLaunchKernelGGL(kern, grid, block, shmem, st, d, s.tdev, s.d_consts, s.d_recs, x0, xr, ur, ref_stride, B, S, k, e); \
    } while (0)
    if (lpi == 4) SPCIES_CS_LAUNCH(4);
    else if (lpi == 8) SPCIES_CS_LAUNCH(8);
    else if (lpi == 16) SPCIES_CS_LAUNCH(16);
    else if (lpi == 32) SPCIES_CS_LAUNCH(32);
    else SPCIES_CS_LAUNCH(64);
#undef SPCIES_CS_LAUNCH
    SPCIES_HIP_CHECK(hipGetLastError());
    auto gather = [&](long row0, int nrows, double *out) {
        const long total = B * (long)nrows;
        hipLaunchKernelGGL(tile::tile_rows_to_aos_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, S, rows, T,
                           (int)row0, nrows, B, out);
    };
    gather(2 * d.n, d.m, u);  // u = v[2n .. 2n+m) (:229-231)
    const long base[3] = {2L * d.dim, 0, d.dim};  // z, v, lambda = Z | V | LAM
    for (int i = 0; i < 3; i++)
        if (f[i]) gather(base[i], d.dim, f[i]);
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

static int launch_hmpc(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B, double *u,
                       int *k, int *e, double *const *f, hipStream_t st) {
    const long Bp = (B + 63) / 64 * 64;
    const HmpcDev &d = s.hdev;
    const long np = d.dim + d.n_s;
    double *S = s.d_scratch;
    hipLaunchKernelGGL(hmpc_stream_kernel, dim3((unsigned)(Bp / 64)), dim3(64), 0, st, d, s.d_consts, s.d_idx, x0, xr, ur,
                       ref_stride, B, Bp, S, u, k, e);
    SPCIES_HIP_CHECK(hipGetLastError());
    // record fields z, s, z_hat, s_hat, lambda, mu: row slices of PR, RH (the solved rhs holds z_hat, s_hat), DU
    const double *base[3] = {S, S + 2 * np * Bp, S + np * Bp};
    for (int i = 0; i < 6; i++) {
        if (!f[i]) continue;
        const int rows = (i % 2 == 0) ? d.dim : d.n_s;
        const double *src = base[i / 2] + ((i % 2 == 0) ? 0 : (long)d.dim * Bp);
        dim3 tg((unsigned)(Bp / 64), (unsigned)((rows + 63) / 64));
        hipLaunchKernelGGL(soa_to_aos_kernel, tg, dim3(256), 0, st, src, Bp, B, rows, f[i]);
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

static size_t tile_scratch_bytes(const Solver &s, long B) {
    if (!s.tdev.lpi) return 0;
    const long T = 64 / s.tdev.lpi, tiles = ((B + T - 1) / T + tile::WAVES - 1) / tile::WAVES * tile::WAVES;
    if (s.is_cs()) return (size_t)tiles * (3L * s.cdev.dim + s.cdev.n + 2 * (s.cdev.n + s.cdev.m)) * T * sizeof(double);
    const long np = s.soc_dim() + s.soc_ns();
    const long rows = s.is_hmpc() ? 3 * np + (s.hdev.n_eq + s.hdev.n_s) + s.hdev.dim : 3 * np + (s.sdev.n_eq + s.sdev.n_s) + s.sdev.dim;
    return (size_t)tiles * rows * T * sizeof(double);
}

// TILE variant of the two sparse-KKT solvers; record fields and u are row slices of PR / PH / DU
static int launch_tile(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, const double *r,
                       int r_stride, long B, double *u, int *k, int *e, double *const *f, hipStream_t st) {
    const int lpi = s.tdev.lpi, T = 64 / lpi;
    const long tiles = ((B + T - 1) / T + tile::WAVES - 1) / tile::WAVES * tile::WAVES;
    const long np = s.soc_dim() + s.soc_ns();
    const long rows = s.is_hmpc() ? 3 * np + (s.hdev.n_eq + s.hdev.n_s) + s.hdev.dim : 3 * np + (s.sdev.n_eq + s.sdev.n_s) + s.sdev.dim;
    double *S = s.d_scratch;
    const size_t shmem = s.tdev.lds_bytes * tile::WAVES;
    dim3 grid((unsigned)(tiles / tile::WAVES)), block(64 * tile::WAVES);
#define SPCIES_TILE_LAUNCH(LPI)                                                                                            \
    do {                                                                                                                   \
        if (s.is_hmpc()) {                                                                                                 \
            auto kern = tile::hmpc_tile_kernel<LPI>;                                                                       \
            if (shmem > 48 * 1024)                                                                                         \
                SPCIES_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
            hipLaunchKernelGGL(kern, grid, block, shmem, st, s.hdev, s.tdev, s.d_consts, s.d_idx, s.d_recs, x0, xr, ur,    \
                               ref_stride, B, S, k, e);                                                                    \
        } else {                                                                                                           \
            auto kern = tile::soc_tile_kernel<LPI>;                                                                        \
            if (shmem > 48 * 1024)                                                                                         \
                SPCIES_HIP_CHECK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
            hipLaunchKernelGGL(kern, grid, block, shmem, st, s.sdev, s.tdev, s.d_consts, s.d_recs, x0, xr, ur, ref_stride, r, \
                               r_stride, B, S, k, e);                                                                      \
        }                                                                                                                  \
    } while (0)
    if (lpi == 4) SPCIES_TILE_LAUNCH(4);
    else if (lpi == 8) SPCIES_TILE_LAUNCH(8);
    else if (lpi == 16) SPCIES_TILE_LAUNCH(16);
    else if (lpi == 32) SPCIES_TILE_LAUNCH(32);
    else SPCIES_TILE_LAUNCH(64);
#undef SPCIES_TILE_LAUNCH
    SPCIES_HIP_CHECK(hipGetLastError());
    const int dim = s.soc_dim(), n_s = s.soc_ns();
    auto gather = [&](int row0, int nrows, double *out) {
        const long total = B * (long)nrows;
        hipLaunchKernelGGL(tile::tile_rows_to_aos_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, S, rows, T,
                           row0, nrows, B, out);
    };
    gather(0, s.host.m, u);  // u = first m entries of z (:283-285)
    // z, s | z_hat, s_hat | lambda, mu  =  PR | PH | DU
    const long base[3] = {0, 2 * np, np};
    for (int i = 0; i < 6; i++)
        if (f[i]) gather((int)(base[i / 2] + ((i % 2) ? dim : 0)), (i % 2) ? n_s : dim, f[i]);
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

static int solve_device_scaled(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B,
                               double *u, int *k, int *e, double *const *f, const double *extra, int extra_stride,
                               hipStream_t st);

#pragma clang fp contract(off)  // the reference's mul-then-add, not an FMA
// out[i][j] = scale[j] * (in[i][j] - op[j])  (code_laxMPC_ADMM_C.c:84-91)
__global__ __launch_bounds__(256) void eng_in_kernel(const double *__restrict__ in, long rows, int w, const double *__restrict__ scale,
                                                     const double *__restrict__ op, double *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * w) return;
    const int j = (int)(i % w);
    out[i] = scale[j] * (in[i] - op[j]);
}
// u[i][j] = u[i][j] * scaling_i_u[j] + OpPoint_u[j]  (:642-646)
__global__ __launch_bounds__(256) void eng_out_kernel(double *__restrict__ u, long rows, int w, const double *__restrict__ sc,
                                                      const double *__restrict__ op) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * w) return;
    const int j = (int)(i % w);
    u[i] = u[i] * sc[j] + op[j];
}

// time-varying solvers: the model's LB / UB columns arrive in engineering units too (code_laxMPC_ADMM_C.c:91-100); A, B, Q, R pass as they are
__global__ __launch_bounds__(256) void eng_tv_model_kernel(const double *__restrict__ in, long rows, int w, int off_lb, int n, int nm,
                                                           const double *__restrict__ scx, const double *__restrict__ opx,
                                                           const double *__restrict__ scu, const double *__restrict__ opu, double *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * w) return;
    const int j = (int)(i % w);
    double x = in[i];
    if (j >= off_lb) {
        const int r = (j - off_lb) % nm;
        x = (r < n) ? scx[r] * (x - opx[r]) : scu[r - n] * (x - opu[r - n]);
    }
    out[i] = x;
}

// Option in_engineering wraps every solver: scale the arguments, solve, un-scale the control action.
static int solve_device(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B,
                        double *u, int *k, int *e, double *const *f, const double *extra, int extra_stride,
                        hipStream_t st) {
    if (B <= 0) return 0;
    if (!s.eng) return solve_device_scaled(s, x0, xr, ur, ref_stride, B, u, k, e, f, extra, extra_stride, st);
    const long n = s.host.n, m = s.host.m, nref = ref_stride ? B : 1;
    const long msz = s.tv ? s.tv_model_size() : 0, mrows = (s.tv && extra) ? (extra_stride ? B : 1) : 0;
    const size_t need = (size_t)(B * n + nref * (n + m) + mrows * msz) * sizeof(double);
    if (need > s.eng_in_bytes) {
        if (s.d_eng_in) SPCIES_HIP_CHECK(hipFree(s.d_eng_in));
        s.d_eng_in = nullptr; s.eng_in_bytes = 0;
        SPCIES_HIP_CHECK(hipMalloc((void **)&s.d_eng_in, need));
        s.eng_in_bytes = need;
    }
    double *sx0 = s.d_eng_in, *sxr = sx0 + B * n, *sur = sxr + nref * n;
    const double *scx = s.d_eng, *opx = scx + n, *scu = opx + n, *opu = scu + m, *sciu = opu + m;
    auto blocks = [](long cnt) { return dim3((unsigned)((cnt + 255) / 256)); };
    hipLaunchKernelGGL(eng_in_kernel, blocks(B * n), dim3(256), 0, st, x0, B, (int)n, scx, opx, sx0);
    hipLaunchKernelGGL(eng_in_kernel, blocks(nref * n), dim3(256), 0, st, xr, nref, (int)n, scx, opx, sxr);
    hipLaunchKernelGGL(eng_in_kernel, blocks(nref * m), dim3(256), 0, st, ur, nref, (int)m, scu, opu, sur);
    if (mrows) {  // time-varying: the bounds of every instance's model in scaled units (the model layout: A, B, Q, R, LB, UB)
        double *sm = sur + nref * m;
        hipLaunchKernelGGL(eng_tv_model_kernel, blocks(mrows * msz), dim3(256), 0, st, extra, mrows, (int)msz, (int)(n * n + n * m + n + m), (int)n,
                           (int)(n + m), scx, opx, scu, opu, sm);
        extra = sm;
    }
    SPCIES_HIP_CHECK(hipGetLastError());
    int rc = solve_device_scaled(s, sx0, sxr, sur, ref_stride, B, u, k, e, f, extra, extra_stride, st);
    if (rc) return rc;
    hipLaunchKernelGGL(eng_out_kernel, blocks(B * m), dim3(256), 0, st, u, B, (int)m, sciu, opu);
    SPCIES_HIP_CHECK(hipGetLastError());
    return 0;
}

static int solve_device_scaled(Solver &s, const double *x0, const double *xr, const double *ur, int ref_stride, long B,
                               double *u, int *k, int *e, double *const *f_in, const double *extra, int extra_stride,
                               hipStream_t st) {
    if (B <= 0) return 0;
    // "a NULL entry skips that output" (include/spcies_hip.h) holds for every variant: the register-resident kernels (MFMA,
    // MFMA4, BSP, and every MFMA4R form: lax / equ ADMM, FISTA, MPCT EADMM and the time-varying pair) write the whole record
    // or nothing, so the fields the caller left out land in handle-owned scratch.
    double *f[6] = {f_in[0], f_in[1], f_in[2], f_in[3], f_in[4], f_in[5]};
    {
        const int nf = s.n_fields();
        bool any = false, all = true;
        for (int i = 0; i < nf; i++) { any |= f[i] != nullptr; all &= f[i] != nullptr; }
        const int var = resolve_variant(s);
        if (any && !all && (var == SPCIES_VARIANT_MFMA || var == SPCIES_VARIANT_MFMA4 || var == SPCIES_VARIANT_BSP || var == SPCIES_VARIANT_MFMA4R)) {
            size_t need = 0;
            for (int i = 0; i < nf; i++)
                if (!f[i]) need += (size_t)B * s.field_dim(i) * sizeof(double);
            if (need > s.part_bytes) {
                if (s.d_part) SPCIES_HIP_CHECK(hipFree(s.d_part));
                s.d_part = nullptr; s.part_bytes = 0;
                SPCIES_HIP_CHECK(hipMalloc((void **)&s.d_part, need));
                s.part_bytes = need;
            }
            double *p = s.d_part;
            for (int i = 0; i < nf; i++)
                if (!f[i]) { f[i] = p; p += (size_t)B * s.field_dim(i); }
        }
    }
    if (s.is_soc() && !s.is_hmpc() && !extra)
        return fail(SPCIES_HIP_EINVAL, "ellipMPC soc solvers take a 4th input r (extra): Spcies:ellipMPC:nrhs:r");
    if (s.is_cs()) {
        s.cdev.k_max = s.host.k_max; s.cdev.tol = s.host.tol;  // set_exit overrides
        if (resolve_variant(s) == SPCIES_VARIANT_FUSED)
            return csfused::launch(s.csf, s.cdev.k_max, s.cdev.tol, x0, xr, ur, ref_stride, B, u, k, e, f[0], f[1], f[2], st);
        if (resolve_variant(s) == SPCIES_VARIANT_TILE) {
            if (!s.tdev.lpi) return fail(SPCIES_HIP_ENOSUP, "TILE variant not available: the LDL right-hand side does not fit the LDS");
            int rc = ensure_scratch(s, tile_scratch_bytes(s, B));
            if (rc) return rc;
            return launch_cs_tile(s, x0, xr, ur, ref_stride, B, u, k, e, f, st);
        }
        if (resolve_variant(s) != SPCIES_VARIANT_STREAM) return fail(SPCIES_HIP_ENOSUP, "MPCT ADMM cs: variants STREAM, TILE and FUSED are built");
        int rc = ensure_scratch(s, stream_scratch_bytes(s, B, true));
        if (rc) return rc;
        return launch_cs_stream(s, x0, xr, ur, ref_stride, B, u, k, e, f, st);
    }
    if (s.is_hdense()) {
        // (coupled constraints: the FUSED kernel carries u and the operand of its last product; the z record follows from that operand in a
        // small kernel of hmpc_fused.hip - since round 4 no call of this solver is handed to the library GEMM unless GEMM is asked for by name)
        if (resolve_variant(s) == SPCIES_VARIANT_FUSED) {
            const hdense::Host &hh = s.hd_host;  // k_max / tolerances: set_exit overrides land here
            return hfused::launch(s.hfused, hh.k_max, hh.tol_p, hh.tol_d, hh.rho, hh.rho_i, 0.0, 0.0, hh.alpha, x0, xr, ur, ref_stride, B, u, k,
                                  e, f, st);
        }
        if (s.variant != SPCIES_VARIANT_AUTO && s.variant != SPCIES_VARIANT_GEMM && s.variant != SPCIES_VARIANT_STREAM)
            return fail(SPCIES_HIP_ENOSUP, "HMPC without the splitting: variants FUSED, GEMM and STREAM are built");
        hdense::Dev &hd = s.hd_plan.dev;
        hd.k_max = s.hd_host.k_max; hd.tol_p = s.hd_host.tol_p; hd.tol_d = s.hd_host.tol_d;  // set_exit overrides
        const bool stream = resolve_variant(s) == SPCIES_VARIANT_STREAM;
        int rc = ensure_scratch(s, stream ? hdense::stream_scratch_bytes(hd, B) : hdense::scratch_bytes(hd, B));
        if (rc) return rc;
        if (stream) return hdense::launch_stream(s.hd_plan, x0, xr, ur, ref_stride, B, s.d_scratch, u, k, e, f, st);
        return hdense::launch(s.hd_plan, x0, xr, ur, ref_stride, B, s.d_scratch, u, k, e, f, st);
    }
    if (s.is_hmpc() && resolve_variant(s) == SPCIES_VARIANT_FUSED) {
        const HmpcDev &hd = s.hdev;  // k_max / tolerances: set_exit overrides land here
        return hfused::launch(s.hfused, hd.k_max, hd.tol_p, hd.tol_d, hd.rho, hd.rho_i, hd.sigma, hd.sigma_i, hd.alpha, x0, xr, ur,
                                    ref_stride, B, u, k, e, f, st);
    }
    if (s.is_hmpc() && resolve_variant(s) == SPCIES_VARIANT_GEMM) {
        if (!s.hgemm.ok) return fail(SPCIES_HIP_ENOSUP, "GEMM variant not available: %s", s.hgemm.why.c_str());
        int rc = ensure_scratch(s, hgemm::scratch_bytes(s.hgemm.dev, B));
        if (rc) return rc;
        s.hgemm.dev.k_max = s.hdev.k_max; s.hgemm.dev.tol_p = s.hdev.tol_p; s.hgemm.dev.tol_d = s.hdev.tol_d;  // set_exit overrides
        return hgemm::launch(s.hgemm, x0, xr, ur, ref_stride, B, s.d_scratch, u, k, e, f, st);
    }
    if (s.is_soc() && !s.is_hmpc() && resolve_variant(s) == SPCIES_VARIANT_BSP)
        return bsp::launch_soc(s.bsp, s.sdev, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, f, st);
    if (s.is_soc() && resolve_variant(s) == SPCIES_VARIANT_TILE) {
        if (!s.tdev.lpi) return fail(SPCIES_HIP_ENOSUP, "TILE variant not available: the LDL right-hand side does not fit the LDS");
        int rc = ensure_scratch(s, tile_scratch_bytes(s, B));
        if (rc) return rc;
        return launch_tile(s, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, f, st);
    }
    if (s.is_hmpc()) {
        if (s.variant != SPCIES_VARIANT_AUTO && s.variant != SPCIES_VARIANT_STREAM)
            return fail(SPCIES_HIP_ENOSUP, "HMPC: variants STREAM and TILE are built");
        int rc = ensure_scratch(s, stream_scratch_bytes(s, B, true));
        if (rc) return rc;
        return launch_hmpc(s, x0, xr, ur, ref_stride, B, u, k, e, f, st);
    }
    if (s.is_soc()) {
        if (!extra) return fail(SPCIES_HIP_EINVAL, "ellipMPC soc solvers take a 4th input r (extra): Spcies:ellipMPC:nrhs:r");
        if (s.variant != SPCIES_VARIANT_AUTO && s.variant != SPCIES_VARIANT_STREAM)
            return fail(SPCIES_HIP_ENOSUP, "ellipMPC soc: variants STREAM and TILE are built");
        int rc = ensure_scratch(s, stream_scratch_bytes(s, B, true));
        if (rc) return rc;
        return launch_soc(s, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, f, st);
    }
    if (s.method == SPCIES_EADMM) {
        const int ev = resolve_variant(s);
        if (ev == SPCIES_VARIANT_MFMA4R) {
            if (!s.erplan.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4R variant not available: %s", s.erplan.why.c_str());
            return er::launch(s.erplan, s.host.k_max, s.host.tol, x0, xr, ur, ref_stride, B, u, k, e, f[0], f[1], f[2], f[3], st);
        }
        if (ev == SPCIES_VARIANT_MFMA4G) {
            if (!s.g4plan.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4G variant not available: %s", s.g4plan.why.c_str());
            int rc = ensure_scratch(s, g4::eadmm_state_bytes(s.g4plan, s.host, B));
            if (rc) return rc;
            return g4::launch_eadmm_g(s.g4plan, s.host, x0, xr, ur, ref_stride, B, s.d_scratch, u, k, e, f[0], f[1], f[2], f[3], st);
        }
        if (ev != SPCIES_VARIANT_STREAM) return fail(SPCIES_HIP_ENOSUP, "EADMM: variants STREAM, MFMA4G and MFMA4R are built");
        if (!eadmm_stream_shape_built(s.host.n, s.host.m) && !stream_rtc_applies(s))
            return fail(SPCIES_HIP_ENOSUP, "EADMM STREAM variant not instantiated for n=%d m=%d", s.host.n, s.host.m);
        int rc = ensure_scratch(s, stream_scratch_bytes(s, B, true));
        if (rc) return rc;
        if (s.e_general || !eadmm_stream_shape_built(s.host.n, s.host.m)) return launch_eadmm_rtc(s, x0, xr, ur, ref_stride, B, u, k, e, f[0], f[1], f[2], f[3], st);  // any plant size
        return launch_eadmm(s, x0, xr, ur, ref_stride, B, u, k, e, f[0], f[1], f[2], f[3], st);
    }
    double *z = f[0], *v = (s.method == SPCIES_FISTA) ? nullptr : f[1], *lam = (s.method == SPCIES_FISTA) ? f[1] : f[2];
    if (s.method == SPCIES_FISTA && s.tv) {
        if (!extra) return fail(SPCIES_HIP_EINVAL, "time-varying solvers take A, B, Q, R, LB, UB with every call (extra): Spcies:laxMPC:nrhs:number");
        if (extra_stride != 0 && extra_stride != s.tv_model_size())
            return fail(SPCIES_HIP_EINVAL, "time-varying: extra_stride must be 0 (shared model) or %d", s.tv_model_size());
        if (s.variant != SPCIES_VARIANT_AUTO && s.variant != SPCIES_VARIANT_STREAM && !(s.variant == SPCIES_VARIANT_MFMA4R && tvr_ok(s)))
            return fail(SPCIES_HIP_ENOSUP, "time-varying FISTA: variants STREAM and MFMA4R (factors in registers, admm_tvr_kernel.inc) are built");
        if (!s.tvrp.update_builtin && tvr_ok(s) && resolve_variant(s) == SPCIES_VARIANT_MFMA4R)  // any other (n, m): everything run-time specialised
            return launch_tv_rtc(s, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, z, nullptr, lam, st);
        if (s.host.n == 6 && s.host.m == 2) return launch_fista_tv_nm<6, 2>(s, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, z, lam, st);
        if (s.host.n == 12 && s.host.m == 2) return launch_fista_tv_nm<12, 2>(s, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, z, lam, st);
        return launch_tv_stream_rtc(s, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, z, nullptr, lam, st);  // STREAM at any other plant size
    }
    if (s.method == SPCIES_FISTA) {
        const int fv = resolve_variant(s);
        if (fv == SPCIES_VARIANT_MFMA4R) {
            if (!s.frplan.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4R variant not available: %s", s.frplan.why.c_str());
            if (!z != !lam) {  // the kernel writes both record fields or none: the missing one goes to handle-owned scratch
                const size_t need = (size_t)B * (z ? (size_t)s.host.N * s.host.n : (size_t)s.host.dim()) * sizeof(double);
                int rc = ensure_scratch(s, need);
                if (rc) return rc;
                (z ? lam : z) = s.d_scratch;
            }
            return fr::launch(s.frplan, s.host.k_max, s.host.tol, x0, xr, ur, ref_stride, B, u, k, e, z, lam, st);
        }
        if (fv == SPCIES_VARIANT_MFMA4G) {
            if (!s.g4plan.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4G variant not available: %s", s.g4plan.why.c_str());
            int rc = ensure_scratch(s, g4::fista_state_bytes(s.g4plan, s.host, B));
            if (rc) return rc;
            return g4::launch_fista_g(s.g4plan, s.host, x0, xr, ur, ref_stride, B, s.d_scratch, u, k, e, z, lam, st);
        }
        if (fv != SPCIES_VARIANT_STREAM) return fail(SPCIES_HIP_ENOSUP, "FISTA: variants STREAM, MFMA4G and MFMA4R are built");
        if (!stream_shape_built(s.host.n, s.host.m) && !stream_rtc_applies(s))
            return fail(SPCIES_HIP_ENOSUP, "STREAM variant not instantiated for n=%d m=%d", s.host.n, s.host.m);
        int rc = ensure_scratch(s, stream_scratch_bytes(s, B, z || lam));
        if (rc) return rc;
        if (!stream_shape_built(s.host.n, s.host.m)) return launch_fista_rtc(s, x0, xr, ur, ref_stride, B, u, k, e, z, lam, st);  // any plant size
        return launch_fista(s, x0, xr, ur, ref_stride, B, u, k, e, z, lam, st);
    }
    if (s.tv) {
        if (!extra) return fail(SPCIES_HIP_EINVAL, "time-varying solvers take A, B, Q, R, LB, UB with every call (extra): Spcies:laxMPC:nrhs:number");
        if (extra_stride != 0 && extra_stride != s.tv_model_size())
            return fail(SPCIES_HIP_EINVAL, "time-varying: extra_stride must be 0 (shared model) or %d", s.tv_model_size());
        if (s.variant != SPCIES_VARIANT_AUTO && s.variant != SPCIES_VARIANT_STREAM && !(s.variant == SPCIES_VARIANT_MFMA4R && tvr_ok(s)))
            return fail(SPCIES_HIP_ENOSUP, "time-varying ADMM: variants STREAM and MFMA4R (factors in registers: admm_tvr.hpp) are built");
        if (!s.tvrp.update_builtin && tvr_ok(s) && resolve_variant(s) == SPCIES_VARIANT_MFMA4R)  // any other (n, m): everything run-time specialised
            return launch_tv_rtc(s, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, z, v, lam, st);
        if (s.host.n == 6 && s.host.m == 2) return launch_tv_nm<6, 2>(s, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, z, v, lam, st);
        if (s.host.n == 12 && s.host.m == 2) return launch_tv_nm<12, 2>(s, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, z, v, lam, st);
        return launch_tv_stream_rtc(s, x0, xr, ur, ref_stride, extra, extra_stride, B, u, k, e, z, v, lam, st);  // STREAM at any other plant size
    }
    if (!s.is_soc() && s.bsp.ok && resolve_variant(s) == SPCIES_VARIANT_BSP)
        return bsp::launch_ellip(s.bsp, s.host, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
    if (s.host.ellip) {
        if (s.variant != SPCIES_VARIANT_AUTO && s.variant != SPCIES_VARIANT_STREAM)
            return fail(SPCIES_HIP_ENOSUP, "ellipMPC ADMM: variants BSP and STREAM are built");
        int rc = ensure_scratch(s, stream_scratch_bytes(s, B, z || v || lam));
        if (rc) return rc;
        if (s.host.n == 6 && s.host.m == 2) return launch_stream_nm<6, 2, true>(s, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
        if (s.host.n == 12 && s.host.m == 2) return launch_stream_nm<12, 2, true>(s, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
        return launch_stream_rtc(s, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);  // any other plant size: the ELLIP instantiation through hiprtc
    }
    const int variant = resolve_variant(s);
    if (variant == SPCIES_VARIANT_MFMA4) {
        if (!s.mfma4.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4 variant not available for this shape: %s", s.mfma4.why.c_str());
        if (s.mfma4_rtc.ok) return rtc::launch_mfma4(s.mfma4_rtc, s.mfma4, s.host, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
        return launch_mfma4(s.mfma4, s.host, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
    }
    if (variant == SPCIES_VARIANT_MFMA4R) {
        if (!s.arplan.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4R variant not available: %s", s.arplan.why.c_str());
        if ((z || v || lam) && !(z && v && lam)) {  // the kernel writes all three record fields or none: the missing ones go to handle-owned scratch
            const size_t one = (size_t)B * (size_t)s.host.dim();
            int rc = ensure_scratch(s, 3 * one * sizeof(double));
            if (rc) return rc;
            if (!z) z = s.d_scratch;
            if (!v) v = s.d_scratch + one;
            if (!lam) lam = s.d_scratch + 2 * one;
        }
        return ar::launch(s.arplan, s.host.k_max, s.host.tol, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
    }
    if (variant == SPCIES_VARIANT_MFMA4G) {
        if (!s.g4plan.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4G variant not available: %s", s.g4plan.why.c_str());
        int rc = ensure_scratch(s, g4::admm_state_bytes(s.g4plan, s.host, B));
        if (rc) return rc;
        return g4::launch_admm_g(s.g4plan, s.host, x0, xr, ur, ref_stride, B, s.d_scratch, u, k, e, z, v, lam, st);
    }
    if (variant == SPCIES_VARIANT_MFMA) {
        if (!s.mfma.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA variant not available for this shape: %s", s.mfma.why.c_str());
        return launch_mfma(s.mfma, s.host, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
    }
    if (s.host.gen) {
        int rc = ensure_scratch(s, stream_scratch_bytes(s, B, z || v || lam));
        if (rc) return rc;
        if (s.host.n == 6 && s.host.m == 2) return launch_stream_nm<6, 2, false, true>(s, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
        if (s.host.n == 12 && s.host.m == 2) return launch_stream_nm<12, 2, false, true>(s, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
        return launch_stream_rtc(s, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);  // any other plant size: the GEN instantiation through hiprtc
    }
    if (!stream_shape_built(s.host.n, s.host.m) && !stream_rtc_applies(s))
        return fail(SPCIES_HIP_ENOSUP, "STREAM variant not instantiated for n=%d m=%d", s.host.n, s.host.m);
    int rc = ensure_scratch(s, stream_scratch_bytes(s, B, z || v || lam));
    if (rc) return rc;
    return launch_stream(s, x0, xr, ur, ref_stride, B, u, k, e, z, v, lam, st);
}

#pragma clang fp contract(off)
// x+ = A x + B u in the operation order of examples/cl_in_C/main_cl_in_C.c:106-116, one instance per thread
__global__ __launch_bounds__(256) void plant_step_kernel(const double *__restrict__ AB, int n, int m, const double *__restrict__ x,
                                                         const double *__restrict__ u, long B, double *__restrict__ xn) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= B) return;
    const int nm = n + m;
    for (int i = 0; i < n; i++) {
        double acc = 0.0;
        for (int j = 0; j < n; j++) acc += AB[i * nm + j] * x[t * n + j];
        for (int j = 0; j < m; j++) acc += AB[i * nm + n + j] * u[t * m + j];
        xn[t * n + i] = acc;
    }
}

}  // namespace spcies

using namespace spcies;

// batch histogram of the iteration counts (what the dense MATLAB solvers' `genHist` gives a user per solve,
// spcies_laxMPC_ADMM_solver.m:253-261, for a batch): bin b counts the instances with (b) k_max / n_bins < k <= (b + 1) k_max / n_bins;
// counts[0..2] = instances with e_flag > 0 (converged), == -1 (k_max reached), anything else; counts[3] = sum of k
__global__ void k_histogram_kernel(const int *__restrict__ k, const int *__restrict__ e, long B, int k_max, int n_bins,
                                   unsigned long long *__restrict__ hist, unsigned long long *__restrict__ counts) {
    extern __shared__ unsigned long long s_h[];
    for (int i = threadIdx.x; i < n_bins + 4; i += blockDim.x) s_h[i] = 0ull;
    __syncthreads();
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < B; i += (long)gridDim.x * blockDim.x) {
        const int kk = k[i], ee = e[i];
        // bin b holds b k_max / n_bins < k <= (b + 1) k_max / n_bins, also when n_bins does not divide k_max:  b = ceil(k n_bins / k_max) - 1
        const long km = k_max > 0 ? k_max : 1;
        long b = kk <= 0 ? 0 : ((long)kk * n_bins + km - 1) / km - 1;
        if (b >= n_bins) b = n_bins - 1;
        atomicAdd(&s_h[b], 1ull);
        atomicAdd(&s_h[n_bins + (ee > 0 ? 0 : (ee == -1 ? 1 : 2))], 1ull);
        atomicAdd(&s_h[n_bins + 3], (unsigned long long)(kk > 0 ? kk : 0));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < n_bins + 4; i += blockDim.x)
        if (s_h[i]) atomicAdd(i < n_bins ? &hist[i] : &counts[i - n_bins], s_h[i]);
}

extern "C" {

int spcies_hip_abi_version(void) { return SPCIES_HIP_ABI_VERSION; }

const char *spcies_hip_last_error(void) { return g_last_error.c_str(); }

int spcies_hip_device_count(int *count) {
    if (!count) return fail(SPCIES_HIP_EINVAL, "count is NULL");
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) {
        *count = 0;
        return fail(SPCIES_HIP_ENODEV, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = c;
    return 0;
}

// releases everything a (possibly half-built) solver owns: the deleter of create()'s guard and the body of destroy()
static void free_solver(Solver *s) {
    if (!s) return;
    if (s->device_bound) hipSetDevice(s->device);
    if (s->d_hist) hipFree(s->d_hist);
    tvr::plan_free(s->tvrp);
    if (s->d_consts) hipFree(s->d_consts);
    if (s->d_scratch) hipFree(s->d_scratch);
    if (s->d_io) hipFree(s->d_io);
    if (s->d_part) hipFree(s->d_part);
    if (s->d_idx) hipFree(s->d_idx);
    if (s->d_recs) hipFree(s->d_recs);
    hgemm::plan_free(s->hgemm);
    hfused::plan_free(s->hfused);
    csfused::plan_free(s->csf);
    fr::plan_free(s->frplan);
    er::plan_free(s->erplan);
    ar::plan_free(s->arplan);
    if (s->srtc.mod) rtc::unload_module(s->srtc.mod);
    hdense::plan_free(s->hd_plan);
    bsp::plan_free(s->bsp);
    if (s->d_eng) hipFree(s->d_eng);
    if (s->d_eng_in) hipFree(s->d_eng_in);
    mfma_plan_free(s->mfma);
    mfma4_plan_free(s->mfma4);
    rtc::module_free(s->mfma4_rtc);
    g4::plan_free(s->g4plan);
    if (s->stream) hipStreamDestroy(s->stream);
    delete s;
}

int spcies_hip_create(const void *blob, size_t bytes, int device, spcies_hip_handle *out) {
    if (!out) return fail(SPCIES_HIP_EINVAL, "out is NULL");
    *out = nullptr;
    std::unique_ptr<Solver, void (*)(Solver *)> s(new Solver, free_solver);  // an error return below frees what was built so far
    int rc = parse_blob(blob, bytes, *s);
    if (rc) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SPCIES_HIP_ENODEV, "no HIP device available (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(SPCIES_HIP_EINVAL, "device %d out of range (0..%d)", device, ndev - 1);
    s->device = device;
    SPCIES_HIP_CHECK(hipSetDevice(device));
    s->device_bound = true;
    {
        hipDeviceProp_t prop;
        SPCIES_HIP_CHECK(hipGetDeviceProperties(&prop, device));
        if (prop.multiProcessorCount > 0) s->num_cu = prop.multiProcessorCount;
    }
    SPCIES_HIP_CHECK(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    rc = upload_consts(*s);
    if (rc) return rc;
    if (s->is_hdense()) {
        const hdense::Host &hh = s->hd_host;
        hfused::NosplitHost fh{hh.n, hh.m, hh.N, hh.dim, hh.n_s, hh.n_box, hh.n_soc, hh.use_soc, hh.symmetric, hh.k_max,
                               hh.tol_p, hh.tol_d, hh.rho, hh.rho_i, hh.alpha, hh.M1.data(), hh.M2.data(),
                               hh.A.data(), hh.QQ.data(), hh.Te.data(), hh.Se.data(), hh.LB.data(), hh.UB.data(), hh.LBy.data(), hh.UBy.data(),
                               hh.d.empty() ? nullptr : hh.d.data(), hh.C_val.data(), hh.C_row.data(), hh.C_col.data()};
        rc = hfused::plan_build_nosplit(s->hfused, fh);
        if (rc) return rc;
    }
    if (s->is_cs()) {  // MPCT ADMM cs: the dense operator of the FUSED variant (cs_fused.hpp)
        const CsDev &cd = s->cdev;
        const double *F = s->soc_f64.data();
        const int *I = s->soc_i32.data();
        csfused::Host ch{cd.n, cd.m, cd.N, cd.dim, cd.nrow, cd.scalar_rho, cd.rho, cd.scalar_rho ? nullptr : F + cd.rho_v,
                         F + cd.Tz, F + cd.Sz, F + cd.LB, F + cd.UB, F + cd.L_val, F + cd.Dinv, F + cd.AHi_val, F + cd.HiA_val, F + cd.Hi_val,
                         I + cd.L_col, I + cd.L_row, I + cd.AHi_col, I + cd.AHi_row, I + cd.HiA_col, I + cd.HiA_row, I + cd.Hi_col, I + cd.Hi_row};
        const char *ev = getenv("SPCIES_HIP_CSFUSED");
        if (ev && ev[0] == '0') s->csf.why = "SPCIES_HIP_CSFUSED=0";
        else {
            rc = csfused::plan_build(s->csf, ch);
            if (rc) return rc;
        }
    }
    if (s->is_hmpc() && !s->h_M1.empty()) {
        const HmpcDev &hd = s->hdev;
        const double *F = s->soc_f64.data();
        if (hd.coupled) {  // coupled output constraints: the GEMM variant's projection kernel knows z and cone rows only
            s->hgemm.why = "coupled output constraints: FUSED, TILE and STREAM are built";
        } else {
            rc = hgemm::plan_build(s->hgemm, hd, s->h_M1, s->h_M2, s->h_bh_nat, F + hd.A, F + hd.QQ, F + hd.Te, F + hd.Se, F + hd.LB,
                                   hd.dim - 3 * (hd.n + hd.m), F + hd.UB, F + hd.LBy, F + hd.UBy);
            if (rc) return rc;
        }
        hfused::SplitHost fh{hd.n, hd.m, hd.N, hd.dim, hd.n_s, hd.n_eq, hd.n_soc, hd.use_soc, hd.symmetric, hd.k_max, hd.coupled, hd.n_y,
                             hd.tol_p, hd.tol_d, hd.rho, hd.rho_i, hd.sigma, hd.sigma_i, hd.alpha,
                             s->h_M1.data(), s->h_M2.data(), s->h_bh_nat.data(),
                             F + hd.A, F + hd.QQ, F + hd.Te, F + hd.Se, F + hd.LB, F + hd.UB, F + hd.LBy, F + hd.UBy};
        rc = hfused::plan_build_split(s->hfused, fh);
        if (rc) return rc;
    }
    // ellipMPC soc: compile the controller's block program now (hiprtc, a few seconds; SPCIES_HIP_BSP=0 turns it off).  A
    // failure is not an error: AUTO then runs TILE, and the reason is reported if BSP is asked for.
    if (s->is_soc() && !s->is_hmpc() && !s->bsp.src.empty()) {
        const char *ev = getenv("SPCIES_HIP_BSP");
        if (!(ev && ev[0] == '0') && bsp::finish_soc(s->bsp, s->sdev, s->soc_f64.data(), s->soc_i32.data()) != 0) s->bsp.why = g_last_error, s->bsp.build_failed = true;
    }
    // laxMPC ADMM with vector rho / stage-wise bounds: the register-resident MFMA4 kernels do not take them, and the block
    // program is 1.4x faster than MFMA4G there (12.5 against 17.4 ms at the C2 shape)
    if ((s->formulation == SPCIES_LAXMPC || s->formulation == SPCIES_EQUMPC) && s->method == SPCIES_ADMM && s->host.gen && !s->tv && !s->eng &&
        !s->host.ellip) {
        const char *ev = getenv("SPCIES_HIP_BSP");
        if (!(ev && ev[0] == '0')) {
            rc = bsp::build_ellip(s->bsp, s->host);
            if (rc) return rc;
            if (!s->bsp.src.empty() && bsp::finish_ellip(s->bsp, s->host) != 0) s->bsp.why = g_last_error, s->bsp.build_failed = true;
        }
    }
    if (s->host.ellip && !s->bsp.src.empty()) {  // ellipMPC ADMM: the same kind of program (ellip_bsp.hpp)
        const char *ev = getenv("SPCIES_HIP_BSP");
        if (!(ev && ev[0] == '0') && bsp::finish_ellip(s->bsp, s->host) != 0) s->bsp.why = g_last_error, s->bsp.build_failed = true;
    }
    if (s->eng) {
        SPCIES_HIP_CHECK(hipMalloc((void **)&s->d_eng, s->eng_v.size() * sizeof(double)));
        SPCIES_HIP_CHECK(hipMemcpy(s->d_eng, s->eng_v.data(), s->eng_v.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    if (s->method == SPCIES_ADMM && (s->formulation == SPCIES_LAXMPC || s->formulation == SPCIES_EQUMPC) && !s->tv && !s->host.ellip) {
        if (s->host.gen) {
            s->mfma.why = s->mfma4.why = "vector rho / stage-wise bounds: use MFMA4G or STREAM";
        } else {
            rc = mfma_plan_build(s->mfma, s->host);
            if (rc) return rc;
            rc = mfma4_plan_build(s->mfma4, s->host);
            if (rc) return rc;
        }
        rc = g4::admm_plan_build(s->g4plan, s->host);
        if (rc) return rc;
        // no build-time MFMA4 kernel of this shape: specialise one now (hiprtc, about a second; SPCIES_HIP_RTC=0 turns it
        // off).  A failure is not an error: AUTO then runs MFMA4G, and the reason is reported if MFMA4 is asked for.
        const char *ev = getenv("SPCIES_HIP_RTC");
        if (s->mfma4.needs_rtc && !(ev && ev[0] == '0')) {
            if (ensure_mfma4_rtc(*s) != 0) s->mfma4.why = g_last_error, s->mfma4.build_failed = true;
        }
        // MFMA4R: where neither MFMA4 nor MFMA holds the controller (more than 112 slab registers, n + m > 16, blocks past the LDS) - or on
        // request (SPCIES_AR_ALWAYS=1: tests and comparisons at shapes MFMA4 serves).  A failure is not an error: AUTO then runs MFMA4G.
        // (vector rho / stage-wise bounds: where the block program - BSP, the faster one while its table fits the LDS - is not available)
        if (!s->tv && !s->host.ellip && ((!s->mfma4.ok && !s->mfma.ok && !(s->host.gen && s->bsp.ok)) || getenv("SPCIES_AR_ALWAYS"))) {
            rc = ar::plan_build(s->arplan, s->host);
            if (rc) return rc;
        } else {
            s->arplan.why = "MFMA4 holds this controller in registers (MFMA4R is built for the shapes past it; SPCIES_AR_ALWAYS=1 builds it anyway)";
        }
        // kernel experiments: SPCIES_MFMA4_RTC_FLAGS="-DX=1 ..." re-specialises a built-in shape with extra compiler options
        if (s->mfma4.ok && !s->mfma4_rtc.ok && getenv("SPCIES_MFMA4_RTC_FLAGS")) {
            const Mfma4Layout &L = s->mfma4.lay;
            rc = rtc::compile_mfma4(s->mfma4_rtc, L.N, L.KX, L.KS, L.terminal, s->mfma4.unit);
            if (rc) return rc;
        }
    }
    if (s->method == SPCIES_EADMM) {
        g4::EadmmGHost eh{&s->e_rho, &s->e_rho0, &s->e_rhos, &s->e_LB0, &s->e_UB0, &s->e_LBs, &s->e_UBs, &s->e_S, &s->e_H1i, &s->e_W2, &s->e_H3i};
        eh.diag = !s->e_general;
        eh.Q_bi = &s->e_Qbi; eh.Q_mi = &s->e_Qmi; eh.R_bi = &s->e_Rbi; eh.R_mi = &s->e_Rmi;
        rc = g4::eadmm_plan_build(s->g4plan, s->host, eh);
        if (rc) return rc;
        // MFMA4R: the whole iteration state on the chip, the kernel specialised for this controller (hiprtc unless the shape was
        // instantiated at build time).  A failure is not an error: AUTO then runs MFMA4G.
        {
            er::Host eh2{s->host.n, s->host.m, s->host.N, s->host.k_max, s->host.tol, s->host.AB.data(), s->host.Alpha.data(), s->host.Beta.data(),
                         s->host.T.data(), s->e_S.data(), s->e_rho.data(), s->e_rho0.data(), s->e_rhos.data(), s->host.LB.data(), s->host.UB.data(),
                         s->e_LB0.data(), s->e_UB0.data(), s->e_LBs.data(), s->e_UBs.data(), s->e_H1i.data(), s->e_W2.data(), s->e_H3i.data()};
            if (s->e_general) {  // IS_DIAG == 0: the dense inverses of the blocks of H3 instead of the vector H3i
                eh2.general = true;
                eh2.Q_bi = s->e_Qbi.data(); eh2.Q_mi = s->e_Qmi.data(); eh2.R_bi = s->e_Rbi.data(); eh2.R_mi = s->e_Rmi.data();
                eh2.AB_bi = s->e_ABbi.data(); eh2.AB_mi = s->e_ABmi.data();
            }
            rc = er::plan_build(s->erplan, eh2);
            if (rc) return rc;
        }
    }
    if (s->method == SPCIES_FISTA && !s->tv) {
        g4::FistaGHost fh{&s->QRi, &s->Td, &s->Ti};
        rc = g4::fista_plan_build(s->g4plan, s->host, fh);
        if (rc) return rc;
        // MFMA4R: the kernel is specialised for this controller now (hiprtc; SPCIES_HIP_RTC=0 turns it off).  A failure is
        // not an error: AUTO then runs MFMA4G, and the reason is reported if MFMA4R is asked for.
        const char *ev = getenv("SPCIES_HIP_RTC");
        if (ev && ev[0] == '0') {
            s->frplan.why = "run-time specialisation switched off (SPCIES_HIP_RTC=0)";
        } else {
            fr::Host fh2{s->host.n, s->host.m, s->host.N, s->host.k_max, s->host.terminal, s->host.tol, s->host.AB.data(), s->host.Alpha.data(),
                         s->host.Beta.data(), s->host.Q.data(), s->host.R.data(), s->QRi.data(), s->Td.data(), s->Ti.data(), s->host.LB.data(),
                         s->host.UB.data()};
            rc = fr::plan_build(s->frplan, fh2);
            if (rc) return rc;
        }
    }
    // what AUTO had to give up (a failed run-time specialisation is not an error, but the caller can ask)
    if (s->tv && (s->method == SPCIES_ADMM || s->method == SPCIES_FISTA)) {  // time-varying ADMM / FISTA: the register-resident variant (hiprtc for a horizon without a build-time kernel)
        const char *ev = getenv("SPCIES_HIP_TVR");
        if (ev && ev[0] == '0') s->tvrp.why = "SPCIES_HIP_TVR=0";
        else {
            rc = tvr::plan_build(s->tvrp, s->host.n, s->host.m, s->host.N, s->host.terminal, s->method == SPCIES_FISTA);
            if (rc) return rc;
        }
    }
    // Two kinds of reasons: the variant does not APPLY to this controller (general Q and R, a shape outside the packer, state beyond
    // registers + LDS, switched off by the caller) - and the variant applies but could not be BUILT (hiprtc missing, compile error).
    auto note = [&](const char *name, bool wanted, bool ok, const std::string &why, bool build_failed) {
        if (!wanted || ok) return;
        s->notes += std::string(s->notes.empty() ? "" : "; ") + name + " unavailable: " + why;
        if (build_failed) s->build_failures += std::string(s->build_failures.empty() ? "" : "; ") + name + ": " + why;
    };
    note("MFMA4R", s->method == SPCIES_FISTA && !s->tv, s->frplan.ok, s->frplan.why, s->frplan.build_failed);
    note("MFMA4R", s->method == SPCIES_EADMM, s->erplan.ok, s->erplan.why, s->erplan.build_failed);
    note("MFMA4R", s->method == SPCIES_ADMM && !s->tv && !s->host.gen && !s->host.ellip && !s->mfma4.ok && !s->mfma.ok &&
                       (s->formulation == SPCIES_LAXMPC || s->formulation == SPCIES_EQUMPC), s->arplan.ok, s->arplan.why, s->arplan.build_failed);
    note("MFMA4R", s->tv && (s->method == SPCIES_ADMM || s->method == SPCIES_FISTA), s->tvrp.ok, s->tvrp.why, s->tvrp.build_failed);
    note("BSP", (s->is_soc() && !s->is_hmpc()) || s->host.ellip, s->bsp.ok, s->bsp.why, s->bsp.build_failed);
    note("MFMA4", s->mfma4.needs_rtc, s->mfma4.ok, s->mfma4.why, s->mfma4.build_failed);
    note("FUSED", s->is_hmpc() || s->is_hdense(), s->hfused.ok, s->hfused.why, s->hfused.build_failed);
    note("FUSED", s->is_cs(), s->csf.ok, s->csf.why, s->csf.build_failed);
    if (!s->notes.empty() && getenv("SPCIES_HIP_VERBOSE")) fprintf(stderr, "[spcies_hip] %s\n", s->notes.c_str());
    // SPCIES_HIP_STRICT=1: a faster variant that applies to this controller but could not be built is an error instead of a silent
    // 2-12x slower default - for deployments that must notice.  A variant that does not apply by design is a note, never an error.
    if (!s->build_failures.empty()) {
        const char *strict = getenv("SPCIES_HIP_STRICT");
        if (strict && strict[0] == '1') return fail(SPCIES_HIP_ENOSUP, "SPCIES_HIP_STRICT: %s", s->build_failures.c_str());
    }
    *out = reinterpret_cast<spcies_hip_handle>(s.release());
    return 0;
}

int spcies_hip_get_notes(spcies_hip_handle h, const char **notes) {
    if (!h || !notes) return fail(SPCIES_HIP_EINVAL, "NULL argument");
    *notes = reinterpret_cast<Solver *>(h)->notes.c_str();
    return 0;
}

int spcies_hip_get_extra_width(spcies_hip_handle h, long *doubles_per_instance) {
    if (!h || !doubles_per_instance) return fail(SPCIES_HIP_EINVAL, "NULL argument");
    Solver *s = reinterpret_cast<Solver *>(h);
    *doubles_per_instance = s->tv ? (long)s->tv_model_size() : 1;
    return 0;
}

int spcies_hip_host_alloc(size_t bytes, void **ptr) {
    if (!ptr) return fail(SPCIES_HIP_EINVAL, "NULL argument");
    *ptr = nullptr;
    if (bytes == 0) return 0;
    SPCIES_HIP_CHECK(hipHostMalloc(ptr, bytes, hipHostMallocPortable));  // page-locked, visible to every device of the process
    return 0;
}

int spcies_hip_host_free(void *ptr) {
    if (ptr) SPCIES_HIP_CHECK(hipHostFree(ptr));
    return 0;
}

int spcies_hip_rtc_cache_stats(long *hits, long *misses) {
    const rtc::CacheStats st = rtc::CodeCache::instance().stats();  // (the cache's own short lock: never waits for a compilation)
    if (hits) *hits = st.mem_hits + st.disk_hits;
    if (misses) *misses = st.compiles;
    return 0;
}

int spcies_hip_rtc_cache_stats_ex(long *out, int n) {
    if (!out || n < 0) return fail(SPCIES_HIP_EINVAL, "NULL argument");
    const rtc::CacheStats st = rtc::CodeCache::instance().stats();
    const long all[6] = {st.mem_hits, st.disk_hits, st.compiles, st.evictions, st.disk_writes, st.disk_errors};
    for (int i = 0; i < n && i < 6; i++) out[i] = all[i];
    for (int i = 6; i < n; i++) out[i] = 0;
    return 0;
}

int spcies_hip_rtc_cache_selftest(const char *text, int work_ms, int drop_memory, int *source, unsigned long long *checksum) {
    if (!text) return fail(SPCIES_HIP_EINVAL, "NULL argument");
    if (drop_memory) rtc::CodeCache::instance().clear_memory();
    const rtc::CacheKey key = rtc::make_key("selftest compiler", "selftest.hip", {"kernel"}, {"-O3"}, text);
    // the stand-in compiler: takes work_ms, returns bytes that depend on the text only
    auto compile = [&](rtc::CodeObject &out) -> int {
        if (work_ms == -2) { fail(SPCIES_HIP_EHIP, "selftest: the stand-in compiler process died"); return rtc::CodeCache::RC_COMPILER_DIED; }
        if (work_ms < 0) return fail(SPCIES_HIP_EHIP, "selftest: the stand-in compiler was told to fail");
        usleep((useconds_t)work_ms * 1000);
        const size_t len = strlen(text);
        out.code.resize(std::max<size_t>(4096, len));  // (grows with the text: the size-cap test writes big ones)
        for (size_t i = 0; i < out.code.size(); i++) out.code[i] = (char)(text[i % (len ? len : 1)] ^ (char)(i * 31));
        out.lowered = {std::string("lowered_") + std::to_string(len)};
        return 0;
    };
    std::shared_ptr<const rtc::CodeObject> co;
    int src = -1;
    const int rc = rtc::CodeCache::instance().get(key, compile, &co, &src);
    if (rc) return rc;
    if (source) *source = src;
    if (checksum) *checksum = rtc::fnv1a64(co->code.data(), co->code.size(), rtc::fnv1a64(co->lowered[0].data(), co->lowered[0].size()));
    return 0;
}

int spcies_hip_rtc_compile_selftest(const char *src, int *isolated, unsigned long *code_bytes) {
    if (!src) return fail(SPCIES_HIP_EINVAL, "NULL argument");
    if (isolated) *isolated = 0;
    if (code_bytes) *code_bytes = 0;
    std::lock_guard<std::mutex> lk(rtc::rtc_mutex());
    int rc = rtc::hiprtc().open();
    if (rc) return rc;
    rtc::CodeObject co;
    const int hr = rtc::compile_in_helper(rtc::hiprtc_library_path(), src, "selftest.hip", {"selftest_kernel"},
                                          {"--offload-arch=gfx950", "-O3", "-std=c++17"}, true, co);
    if (hr == 0) return fail(SPCIES_HIP_ENOSUP, "no compiler process (spcies_rtc_helper not found next to the library, or SPCIES_HIP_RTC_ISOLATE=0)");
    if (isolated) *isolated = 1;
    if (hr < 0) return SPCIES_HIP_EHIP;
    if (code_bytes) *code_bytes = (unsigned long)co.code.size();
    return 0;
}

int spcies_hip_k_histogram_device(spcies_hip_handle h, const int *k, const int *e_flag, long B, int n_bins, long *hist, long *counts,
                                  void *stream) {
    if (!h || !hist || !counts) return fail(SPCIES_HIP_EINVAL, "NULL argument");
    if (n_bins < 1 || n_bins > 1024) return fail(SPCIES_HIP_EINVAL, "k_histogram: 1 <= n_bins <= 1024");
    if (B < 0) return fail(SPCIES_HIP_EINVAL, "negative batch");
    for (int i = 0; i < n_bins; i++) hist[i] = 0;
    for (int i = 0; i < 4; i++) counts[i] = 0;
    if (B == 0) return 0;
    if (!k || !e_flag) return fail(SPCIES_HIP_EINVAL, "NULL buffer");
    Solver *s = reinterpret_cast<Solver *>(h);
    std::lock_guard<std::mutex> lk(s->mu);
    SPCIES_HIP_CHECK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    // a small buffer owned by the handle (1024 bins + 4 counters): no hipMalloc / hipFree - a device synchronisation each - per call
    if (!s->d_hist) SPCIES_HIP_CHECK(hipMalloc((void **)&s->d_hist, (1024 + 4) * sizeof(unsigned long long)));
    unsigned long long *d = s->d_hist;
    hipError_t er = hipMemsetAsync(d, 0, (n_bins + 4) * sizeof(unsigned long long), st);
    if (er == hipSuccess) {
        long blocks = (B + 255) / 256;
        if (blocks > 1024) blocks = 1024;
        k_histogram_kernel<<<(unsigned)blocks, 256, (n_bins + 4) * sizeof(unsigned long long), st>>>(k, e_flag, B, s->host.k_max, n_bins, d, d + n_bins);
        er = hipGetLastError();
    }
    std::vector<unsigned long long> hst(n_bins + 4);
    if (er == hipSuccess) er = hipMemcpyAsync(hst.data(), d, (n_bins + 4) * sizeof(unsigned long long), hipMemcpyDeviceToHost, st);
    if (er == hipSuccess) er = hipStreamSynchronize(st);
    if (er != hipSuccess) return fail(SPCIES_HIP_EHIP, "k_histogram: %s", hipGetErrorString(er));
    for (int i = 0; i < n_bins; i++) hist[i] = (long)hst[i];
    for (int i = 0; i < 4; i++) counts[i] = (long)hst[n_bins + i];
    return 0;
}

int spcies_hip_destroy(spcies_hip_handle h) {
    if (!h) return 0;
    free_solver(reinterpret_cast<Solver *>(h));
    return 0;
}

int spcies_hip_get_info(spcies_hip_handle h, spcies_hip_info *info) {
    if (!h || !info) return fail(SPCIES_HIP_EINVAL, "NULL argument");
    Solver *s = reinterpret_cast<Solver *>(h);
    info->formulation = s->formulation;
    info->method = s->method;
    info->submethod = s->submethod;
    info->n = s->host.n; info->m = s->host.m; info->N = s->host.N; info->dim = s->is_soc() ? s->soc_dim() : (s->is_hdense() ? s->hd_host.dim : (s->is_cs() ? s->cdev.dim : s->host.dim()));
    info->k_max = s->host.k_max; info->tol = s->host.tol; info->rho = s->host.rho;
    info->variant = resolve_variant(*s);
    info->dim_lambda = s->lam_dim();
    info->device = s->device;
    return 0;
}

int spcies_hip_set_variant(spcies_hip_handle h, int variant) {
    if (!h) return fail(SPCIES_HIP_EINVAL, "NULL handle");
    Solver *s = reinterpret_cast<Solver *>(h);
    if (variant < SPCIES_VARIANT_AUTO || variant > SPCIES_VARIANT_MFMA4R) return fail(SPCIES_HIP_EINVAL, "unknown variant %d", variant);
    if (variant == SPCIES_VARIANT_FUSED && s->is_cs()) {
        if (!s->csf.ok) return fail(SPCIES_HIP_ENOSUP, "FUSED variant not available: %s", s->csf.why.c_str());
        s->variant = variant;
        return 0;
    }
    if (variant == SPCIES_VARIANT_FUSED) {
        if (!((s->is_hmpc() || s->is_hdense()) && s->hfused.ok))
            return fail(SPCIES_HIP_ENOSUP, "FUSED variant: built for the HMPC solvers whose blob carries the dense M1, M2 (%s)", s->hfused.why.c_str());
        s->variant = variant;
        return 0;
    }
    if (variant == SPCIES_VARIANT_BSP) {
        const bool soc = s->is_soc() && !s->is_hmpc();
        const bool lax = (s->formulation == SPCIES_LAXMPC || s->formulation == SPCIES_EQUMPC) && s->method == SPCIES_ADMM && !s->tv && !s->eng;
        if (!soc && !s->host.ellip && !lax)
            return fail(SPCIES_HIP_ENOSUP, "BSP variant: built for the ellipMPC solvers (ADMM and ADMM soc) and, on request, laxMPC / equMPC ADMM");
        if (lax && !s->bsp.ok && s->bsp.src.empty()) {  // laxMPC ADMM: the program is only generated when it is asked for
            int rc = bsp::build_ellip(s->bsp, s->host);
            if (rc) return rc;
        }
        if (!s->bsp.ok) {  // not compiled at create time (SPCIES_HIP_BSP=0, or it failed): try now and report
            if (s->bsp.src.empty()) return fail(SPCIES_HIP_ENOSUP, "BSP variant not available: %s", s->bsp.why.c_str());
            SPCIES_HIP_CHECK(hipSetDevice(s->device));
            int rc = soc ? bsp::finish_soc(s->bsp, s->sdev, s->soc_f64.data(), s->soc_i32.data()) : bsp::finish_ellip(s->bsp, s->host);
            if (rc) return rc;
        }
        s->variant = variant;
        return 0;
    }
    if (s->is_hdense()) {
        if (variant != SPCIES_VARIANT_AUTO && variant != SPCIES_VARIANT_GEMM && variant != SPCIES_VARIANT_STREAM)
            return fail(SPCIES_HIP_ENOSUP, "HMPC without the splitting: variants GEMM and STREAM are built");
        s->variant = variant;
        return 0;
    }
    if (variant == SPCIES_VARIANT_GEMM && !(s->is_hmpc() && s->hgemm.ok))
        return fail(SPCIES_HIP_ENOSUP, "GEMM variant: built for HMPC split solvers whose blob carries M1, M2 (%s)", s->hgemm.why.c_str());
    if (s->is_cs()) {
        if (variant != SPCIES_VARIANT_AUTO && variant != SPCIES_VARIANT_STREAM && !(variant == SPCIES_VARIANT_TILE && s->tdev.lpi))
            return fail(SPCIES_HIP_ENOSUP, "MPCT ADMM cs: variants STREAM and TILE (when the right-hand side fits the LDS) are built");
        s->variant = variant;
        return 0;
    }
    if (variant == SPCIES_VARIANT_MFMA4R && s->tv) {  // time-varying ADMM: factors in registers (admm_tvr.hpp)
        if (!tvr_ok(*s)) return fail(SPCIES_HIP_ENOSUP, "MFMA4R (time-varying ADMM) unavailable: %s", s->tvrp.why.c_str());
        s->variant = variant;
        return 0;
    }
    if (variant == SPCIES_VARIANT_TILE && !(s->is_soc() && s->tdev.lpi))
        return fail(SPCIES_HIP_ENOSUP, "TILE variant: built for the sparse-KKT solvers (ellipMPC soc, HMPC) whose LDL right-hand side fits the LDS");
    if (variant == SPCIES_VARIANT_MFMA4R && s->method == SPCIES_EADMM && !s->erplan.ok)
        return fail(SPCIES_HIP_ENOSUP, "MFMA4R variant not available for this solver: %s", s->erplan.why.c_str());
    if (variant == SPCIES_VARIANT_MFMA4R && s->method == SPCIES_ADMM) {  // lax / equ ADMM: built on request where MFMA4 serves the controller (admm_r.hpp)
        if (!s->arplan.ok && !s->host.ellip && (s->formulation == SPCIES_LAXMPC || s->formulation == SPCIES_EQUMPC)) {
            SPCIES_HIP_CHECK(hipSetDevice(s->device));
            int rc = ar::plan_build(s->arplan, s->host);
            if (rc) return rc;
        }
        if (!s->arplan.ok) return fail(SPCIES_HIP_ENOSUP, "MFMA4R variant not available for this solver: %s", s->arplan.why.c_str());
        s->variant = variant;
        return 0;
    }
    if (variant == SPCIES_VARIANT_MFMA4R && s->method != SPCIES_EADMM && !s->frplan.ok)
        return fail(SPCIES_HIP_ENOSUP, "MFMA4R variant not available for this solver: %s", s->frplan.why.c_str());
    if (variant == SPCIES_VARIANT_MFMA4G && !s->g4plan.ok)
        return fail(SPCIES_HIP_ENOSUP, "MFMA4G variant not available for this solver: %s", s->g4plan.why.c_str());
    if (variant == SPCIES_VARIANT_MFMA4 && !s->mfma4.ok) {
        SPCIES_HIP_CHECK(hipSetDevice(s->device));
        int rc = ensure_mfma4_rtc(*s);
        if (rc) return rc;
    }
    if (variant == SPCIES_VARIANT_MFMA && !s->mfma.ok)
        return fail(SPCIES_HIP_ENOSUP, "MFMA variant not available for this shape: %s", s->mfma.why.c_str());
    if (variant == SPCIES_VARIANT_STREAM && !(s->method == SPCIES_EADMM ? (eadmm_stream_shape_built(s->host.n, s->host.m) && !s->e_general) : stream_shape_built(s->host.n, s->host.m))) {
        if (!stream_rtc_applies(*s)) return fail(SPCIES_HIP_ENOSUP, "STREAM variant not instantiated for n=%d m=%d", s->host.n, s->host.m);
        SPCIES_HIP_CHECK(hipSetDevice(s->device));
        int rc = ensure_stream_rtc(*s);  // (the plain lax / equ ADMM solvers: specialised now, for any plant size)
        if (rc) return rc;
    }
    s->variant = variant;
    return 0;
}

int spcies_hip_set_exit(spcies_hip_handle h, int k_max, double tol) {
    if (!h) return fail(SPCIES_HIP_EINVAL, "NULL handle");
    Solver *s = reinterpret_cast<Solver *>(h);
    if (k_max > 0) s->host.k_max = s->dev.k_max = s->fdev.k_max = s->edev.k_max = s->sdev.k_max = s->hdev.k_max = s->hd_host.k_max = k_max;
    if (tol >= 0)
        s->hd_host.tol_p = s->hd_host.tol_d = s->host.tol = s->dev.tol = s->fdev.tol = s->edev.tol = s->sdev.tol_p = s->sdev.tol_d = s->hdev.tol_p = s->hdev.tol_d = tol;
    return 0;
}

int spcies_hip_reserve(spcies_hip_handle h, long B) {
    if (!h || B < 0) return fail(SPCIES_HIP_EINVAL, "bad argument");
    Solver *s = reinterpret_cast<Solver *>(h);
    std::lock_guard<std::mutex> lk(s->mu);
    SPCIES_HIP_CHECK(hipSetDevice(s->device));
    if (s->is_hdense())
        return ensure_scratch(*s, std::max(hdense::scratch_bytes(s->hd_plan.dev, B), hdense::stream_scratch_bytes(s->hd_plan.dev, B)));
    size_t need = stream_scratch_bytes(*s, B, true);
    if (s->g4plan.ok && s->method == SPCIES_FISTA) need = std::max(need, g4::fista_state_bytes(s->g4plan, s->host, B));
    if (s->g4plan.ok && s->method == SPCIES_ADMM && !s->is_soc()) need = std::max(need, g4::admm_state_bytes(s->g4plan, s->host, B));
    need = std::max(need, tile_scratch_bytes(*s, B));
    if (s->hgemm.ok) need = std::max(need, hgemm::scratch_bytes(s->hgemm.dev, B));
    if (s->g4plan.ok && s->method == SPCIES_EADMM) need = std::max(need, g4::eadmm_state_bytes(s->g4plan, s->host, B));
    return ensure_scratch(*s, need);
}

// map the (z, v, lambda) triple of the classic entry points onto the record fields
static int classic_fields(Solver *s, double *z, double *v, double *lambda, double **f) {
    for (int i = 0; i < 6; i++) f[i] = nullptr;
    if (s->is_soc()) return fail(SPCIES_HIP_EINVAL, "this solver has a 6-field record (z, s, z_hat, s_hat, lambda, mu): use the _ex entry points");
    if (s->method == SPCIES_EADMM) {
        if (z || v || lambda) return fail(SPCIES_HIP_EINVAL, "EADMM record is (z1, z2, z3, lambda): use the _ex entry points");
    } else if (s->method == SPCIES_FISTA) {
        if (v) return fail(SPCIES_HIP_EINVAL, "FISTA solvers have no v output (sol fields: z, lambda)");
        f[0] = z; f[1] = lambda;
    } else {
        f[0] = z; f[1] = v; f[2] = lambda;
    }
    return 0;
}

int spcies_hip_get_sol_layout(spcies_hip_handle h, int *n_fields, int *dims, const char **names) {
    if (!h || !n_fields) return fail(SPCIES_HIP_EINVAL, "NULL argument");
    Solver *s = reinterpret_cast<Solver *>(h);
    *n_fields = s->n_fields();
    for (int i = 0; i < s->n_fields(); i++) {
        if (dims) dims[i] = s->field_dim(i);
        if (names) names[i] = s->field_name(i);
    }
    return 0;
}

int spcies_hip_solve_batch_device_ex(spcies_hip_handle h, const double *x0, const double *xr, const double *ur,
                                     int ref_stride, const double *extra, int extra_stride, long B, double *u, int *k,
                                     int *e_flag, double *const *fields, int n_fields, void *stream) {
    if (!h) return fail(SPCIES_HIP_EINVAL, "NULL handle");
    if (B < 0) return fail(SPCIES_HIP_EINVAL, "negative batch");
    if (B == 0) return 0;
    if (!x0 || !xr || !ur || !u || !k || !e_flag) return fail(SPCIES_HIP_EINVAL, "NULL buffer");
    Solver *s = reinterpret_cast<Solver *>(h);
    if (fields && n_fields != s->n_fields()) return fail(SPCIES_HIP_EINVAL, "this solver's record has %d fields", s->n_fields());
    double *f[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    if (fields) for (int i = 0; i < n_fields; i++) f[i] = fields[i];
    std::lock_guard<std::mutex> lk(s->mu);
    SPCIES_HIP_CHECK(hipSetDevice(s->device));
    return solve_device(*s, x0, xr, ur, ref_stride, B, u, k, e_flag, f, extra, extra_stride, (hipStream_t)stream);
}

int spcies_hip_solve_batch_device(spcies_hip_handle h, const double *x0, const double *xr, const double *ur,
                                  int ref_stride, long B, double *u, int *k, int *e_flag, double *z, double *v,
                                  double *lambda, void *stream) {
    if (!h) return fail(SPCIES_HIP_EINVAL, "NULL handle");
    Solver *s = reinterpret_cast<Solver *>(h);
    double *f[6];
    int rc = classic_fields(s, z, v, lambda, f);
    if (rc) return rc;
    return spcies_hip_solve_batch_device_ex(h, x0, xr, ur, ref_stride, nullptr, 0, B, u, k, e_flag,
                                            (z || v || lambda) ? f : nullptr, s->n_fields(), stream);
}

int spcies_hip_solve_batch_ex(spcies_hip_handle h, const double *x0, const double *xr, const double *ur, int ref_stride,
                              const double *extra, int extra_stride, long B, double *u, int *k, int *e_flag,
                              double *const *fields, int n_fields, spcies_hip_timing *timing) {
    if (!h) return fail(SPCIES_HIP_EINVAL, "NULL handle");
    if (B < 0) return fail(SPCIES_HIP_EINVAL, "negative batch");
    if (timing) *timing = spcies_hip_timing{0, 0, 0, 0};
    if (B == 0) return 0;
    if (!x0 || !xr || !ur || !u || !k || !e_flag) return fail(SPCIES_HIP_EINVAL, "NULL buffer");
    Solver *s = reinterpret_cast<Solver *>(h);
    if (fields && n_fields != s->n_fields()) return fail(SPCIES_HIP_EINVAL, "this solver's record has %d fields", s->n_fields());
    std::lock_guard<std::mutex> lk(s->mu);
    SPCIES_HIP_CHECK(hipSetDevice(s->device));
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    const size_t n = s->host.n, m = s->host.m;
    const size_t nref = ref_stride ? (size_t)B : 1;
    // device staging: x0 | xr | ur | u | extra | fields... | k | e   (doubles first, ints last)
    const size_t ex_w = s->tv ? (size_t)s->tv_model_size() : 1;  // doubles per instance in `extra`
    const size_t nex = extra ? (extra_stride ? (size_t)B : 1) * ex_w : 0;
    size_t nd = (size_t)B * n + nref * n + nref * m + (size_t)B * m + nex;
    const size_t o_x0 = 0, o_xr = (size_t)B * n, o_ur = o_xr + nref * n, o_u = o_ur + nref * m, o_ex = o_u + (size_t)B * m;
    size_t o_f[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; fields && i < n_fields; i++)
        if (fields[i]) { o_f[i] = nd; nd += (size_t)B * s->field_dim(i); }
    size_t need = nd * sizeof(double) + 2 * (size_t)B * sizeof(int);
    if (need > s->io_bytes) {
        if (s->d_io) SPCIES_HIP_CHECK(hipFree(s->d_io));
        s->d_io = nullptr; s->io_bytes = 0;
        SPCIES_HIP_CHECK(hipMalloc((void **)&s->d_io, need));
        s->io_bytes = need;
    }
    double *d = s->d_io;
    int *dk = reinterpret_cast<int *>(d + nd), *de = dk + B;
    hipStream_t st = s->stream;
    SPCIES_HIP_CHECK(hipMemcpyAsync(d + o_x0, x0, (size_t)B * n * 8, hipMemcpyHostToDevice, st));
    SPCIES_HIP_CHECK(hipMemcpyAsync(d + o_xr, xr, nref * n * 8, hipMemcpyHostToDevice, st));
    SPCIES_HIP_CHECK(hipMemcpyAsync(d + o_ur, ur, nref * m * 8, hipMemcpyHostToDevice, st));
    if (nex) SPCIES_HIP_CHECK(hipMemcpyAsync(d + o_ex, extra, nex * 8, hipMemcpyHostToDevice, st));
    SPCIES_HIP_CHECK(hipStreamSynchronize(st));
    auto t1 = clk::now();
    double *f[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; fields && i < n_fields; i++)
        if (fields[i]) f[i] = d + o_f[i];
    int rc = solve_device(*s, d + o_x0, d + o_xr, d + o_ur, ref_stride, B, d + o_u, dk, de, f, nex ? d + o_ex : nullptr,
                          extra_stride, st);
    if (rc) return rc;
    SPCIES_HIP_CHECK(hipStreamSynchronize(st));
    auto t2 = clk::now();
    SPCIES_HIP_CHECK(hipMemcpyAsync(u, d + o_u, (size_t)B * m * 8, hipMemcpyDeviceToHost, st));
    SPCIES_HIP_CHECK(hipMemcpyAsync(k, dk, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    SPCIES_HIP_CHECK(hipMemcpyAsync(e_flag, de, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, st));
    for (int i = 0; fields && i < n_fields; i++)
        if (fields[i])
            SPCIES_HIP_CHECK(hipMemcpyAsync(fields[i], f[i], (size_t)B * s->field_dim(i) * 8, hipMemcpyDeviceToHost, st));
    SPCIES_HIP_CHECK(hipStreamSynchronize(st));
    auto t3 = clk::now();
    if (timing) {
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        timing->update_time = ms(t0, t1);
        timing->solve_time = ms(t1, t2);
        timing->polish_time = ms(t2, t3);
        timing->run_time = ms(t0, t3);
    }
    return 0;
}

int spcies_hip_solve_batch(spcies_hip_handle h, const double *x0, const double *xr, const double *ur, int ref_stride,
                           long B, double *u, int *k, int *e_flag, double *z, double *v, double *lambda,
                           spcies_hip_timing *timing) {
    if (!h) return fail(SPCIES_HIP_EINVAL, "NULL handle");
    Solver *s = reinterpret_cast<Solver *>(h);
    double *f[6];
    int rc = classic_fields(s, z, v, lambda, f);
    if (rc) return rc;
    return spcies_hip_solve_batch_ex(h, x0, xr, ur, ref_stride, nullptr, 0, B, u, k, e_flag,
                                     (z || v || lambda) ? f : nullptr, s->n_fields(), timing);
}

// the per-instance residual history the dense MATLAB solvers record with genHist > 0 (hRp(k) = ||z - v||_inf, hRd(k) =
// ||v - v_prev||_inf, platforms/Matlab/spcies_laxMPC_ADMM_solver.m:253-261, 311-319).  A diagnostic, not a hot path: the trace is read
// off the SAME kernels a solve runs - iteration j of every instance is the record (z, v) of a solve stopped at k_max = j with the exit
// test off - so what is plotted is what the selected variant computes, and no kernel carries trace code.  K (K + 1) / 2 iterations.
int spcies_hip_residual_trace(spcies_hip_handle h, const double *x0, const double *xr, const double *ur, int ref_stride, long B, int K,
                              double *r_p, double *r_d, int *k_exit) {
    if (!h || !r_p || !r_d) return fail(SPCIES_HIP_EINVAL, "NULL argument");
    if (K < 1 || B < 0) return fail(SPCIES_HIP_EINVAL, "residual trace: K >= 1, B >= 0");
    Solver *s = reinterpret_cast<Solver *>(h);
    const bool lax = (s->formulation == SPCIES_LAXMPC || s->formulation == SPCIES_EQUMPC) && s->method == SPCIES_ADMM && !s->tv && !s->host.ellip;
    if (!lax) return fail(SPCIES_HIP_ENOSUP, "residual trace: built for the lax / equ MPC ADMM solvers (record z, v, lambda)");
    if (B == 0) return 0;
    const size_t dim = (size_t)s->host.dim(), m = (size_t)s->host.m;
    const int k_max0 = s->host.k_max;
    const double tol0 = s->host.tol;
    std::vector<double> u((size_t)B * m), z((size_t)B * dim), v((size_t)B * dim), lam((size_t)B * dim), vprev((size_t)B * dim, 0.0);
    std::vector<int> k((size_t)B), e((size_t)B), kx((size_t)B);
    // the exit iteration under the handle's own settings: entries behind it stay zero, as in the MATLAB record
    int rc = spcies_hip_solve_batch(h, x0, xr, ur, ref_stride, B, u.data(), kx.data(), e.data(), nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    for (long i = 0; i < B * (long)K; i++) r_p[i] = r_d[i] = 0.0;
    for (int j = 1; j <= K && rc == 0; j++) {
        spcies_hip_set_exit(h, j, 0.0);  // exactly j iterations (tol = 0: the reference tests' fixed-iteration setting)
        rc = spcies_hip_solve_batch(h, x0, xr, ur, ref_stride, B, u.data(), k.data(), e.data(), z.data(), v.data(), lam.data(), nullptr);
        if (rc) break;
        for (long i = 0; i < B; i++) {
            if (j <= kx[i]) {
                double rp = 0.0, rd = 0.0;
                for (size_t c = 0; c < dim; c++) {
                    rp = std::max(rp, std::fabs(z[i * dim + c] - v[i * dim + c]));
                    rd = std::max(rd, std::fabs(v[i * dim + c] - vprev[i * dim + c]));
                }
                r_p[i * (long)K + (j - 1)] = rp;
                r_d[i * (long)K + (j - 1)] = rd;
            }
        }
        vprev = v;
    }
    spcies_hip_set_exit(h, k_max0, tol0);
    if (rc) return rc;
    if (k_exit)
        for (long i = 0; i < B; i++) k_exit[i] = kx[i];
    return 0;
}

int spcies_hip_time_device(spcies_hip_handle h, const double *x0, const double *xr, const double *ur, int ref_stride,
                           long B, double *u, int *k, int *e_flag, void *stream, int reps, double *ms_per_launch) {
    if (!h || !ms_per_launch || reps <= 0) return fail(SPCIES_HIP_EINVAL, "bad argument");
    Solver *s = reinterpret_cast<Solver *>(h);
    SPCIES_HIP_CHECK(hipSetDevice(s->device));
    hipStream_t st = (hipStream_t)stream;
    hipEvent_t a, b;
    SPCIES_HIP_CHECK(hipEventCreate(&a));
    SPCIES_HIP_CHECK(hipEventCreate(&b));
    SPCIES_HIP_CHECK(hipEventRecord(a, st));
    for (int i = 0; i < reps; i++) {
        int rc = spcies_hip_solve_batch_device(h, x0, xr, ur, ref_stride, B, u, k, e_flag, nullptr, nullptr, nullptr, stream);
        if (rc) { hipEventDestroy(a); hipEventDestroy(b); return rc; }
    }
    SPCIES_HIP_CHECK(hipEventRecord(b, st));
    SPCIES_HIP_CHECK(hipEventSynchronize(b));
    float ms = 0;
    SPCIES_HIP_CHECK(hipEventElapsedTime(&ms, a, b));
    hipEventDestroy(a);
    hipEventDestroy(b);
    *ms_per_launch = (double)ms / reps;
    return 0;
}

int spcies_hip_closed_loop(spcies_hip_handle h, const double *AB_plant, const double *x0, const double *xr, const double *ur,
                           int ref_stride, long B, int steps, double *x_traj, double *u_traj, int *k_traj, int *e_traj,
                           spcies_hip_timing *timing) {
    if (!h) return fail(SPCIES_HIP_EINVAL, "NULL handle");
    if (B < 0 || steps < 0) return fail(SPCIES_HIP_EINVAL, "negative batch / steps");
    if (timing) *timing = spcies_hip_timing{0, 0, 0, 0};
    if (B == 0 || steps == 0) return 0;
    if (!AB_plant || !x0 || !xr || !ur) return fail(SPCIES_HIP_EINVAL, "NULL buffer");
    Solver *s = reinterpret_cast<Solver *>(h);
    if (s->tv || (s->is_soc() && !s->is_hmpc()))
        return fail(SPCIES_HIP_ENOSUP, "closed loop: solvers with extra inputs (ellipMPC r, time-varying model) are not driven here");
    std::lock_guard<std::mutex> lk(s->mu);
    SPCIES_HIP_CHECK(hipSetDevice(s->device));
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    const size_t n = s->host.n, m = s->host.m, nm = n + m, nref = ref_stride ? (size_t)B : 1;
    // device buffers: AB | xr | ur | x_traj [steps+1][B][n] | u_traj [steps][B][m] | k_traj, e_traj [steps][B]
    const size_t o_ab = 0, o_xr = n * nm, o_ur = o_xr + nref * n, o_x = o_ur + nref * m, o_u = o_x + (size_t)(steps + 1) * B * n;
    const size_t nd = o_u + (size_t)steps * B * m;
    const size_t need = nd * sizeof(double) + 2 * (size_t)steps * B * sizeof(int);
    double *d = nullptr;
    SPCIES_HIP_CHECK(hipMalloc((void **)&d, need));
    int *dk = reinterpret_cast<int *>(d + nd), *de = dk + (size_t)steps * B;
    hipStream_t st = s->stream;
    auto cleanup = [&](int rc) { hipFree(d); return rc; };
#define SPCIES_CL_CHECK(expr)                                                                                   \
    do {                                                                                                        \
        hipError_t e__ = (expr);                                                                                \
        if (e__ != hipSuccess) return cleanup(fail(SPCIES_HIP_EHIP, "%s failed: %s", #expr, hipGetErrorString(e__))); \
    } while (0)
    SPCIES_CL_CHECK(hipMemcpyAsync(d + o_ab, AB_plant, n * nm * 8, hipMemcpyHostToDevice, st));
    SPCIES_CL_CHECK(hipMemcpyAsync(d + o_xr, xr, nref * n * 8, hipMemcpyHostToDevice, st));
    SPCIES_CL_CHECK(hipMemcpyAsync(d + o_ur, ur, nref * m * 8, hipMemcpyHostToDevice, st));
    SPCIES_CL_CHECK(hipMemcpyAsync(d + o_x, x0, (size_t)B * n * 8, hipMemcpyHostToDevice, st));
    SPCIES_CL_CHECK(hipStreamSynchronize(st));
    auto t1 = clk::now();
    double *f[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    for (int t = 0; t < steps; t++) {
        double *xt = d + o_x + (size_t)t * B * n, *ut = d + o_u + (size_t)t * B * m;
        int rc = solve_device(*s, xt, d + o_xr, d + o_ur, ref_stride, B, ut, dk + (size_t)t * B, de + (size_t)t * B, f, nullptr, 0, st);
        if (rc) return cleanup(rc);
        hipLaunchKernelGGL(plant_step_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, d + o_ab, (int)n, (int)m, xt, ut, B,
                           xt + (size_t)B * n);
        SPCIES_CL_CHECK(hipGetLastError());
    }
    SPCIES_CL_CHECK(hipStreamSynchronize(st));
    auto t2 = clk::now();
    if (x_traj) SPCIES_CL_CHECK(hipMemcpyAsync(x_traj, d + o_x, (size_t)(steps + 1) * B * n * 8, hipMemcpyDeviceToHost, st));
    if (u_traj) SPCIES_CL_CHECK(hipMemcpyAsync(u_traj, d + o_u, (size_t)steps * B * m * 8, hipMemcpyDeviceToHost, st));
    if (k_traj) SPCIES_CL_CHECK(hipMemcpyAsync(k_traj, dk, (size_t)steps * B * sizeof(int), hipMemcpyDeviceToHost, st));
    if (e_traj) SPCIES_CL_CHECK(hipMemcpyAsync(e_traj, de, (size_t)steps * B * sizeof(int), hipMemcpyDeviceToHost, st));
    SPCIES_CL_CHECK(hipStreamSynchronize(st));
#undef SPCIES_CL_CHECK
    auto t3 = clk::now();
    if (timing) {
        auto ms = [](clk::duration x) { return std::chrono::duration<double, std::milli>(x).count(); };
        timing->update_time = ms(t1 - t0);
        timing->solve_time = ms(t2 - t1);
        timing->polish_time = ms(t3 - t2);
        timing->run_time = ms(t3 - t0);
    }
    return cleanup(0);
}

}  // extern "C"
