/*
 * spcies_hip.h - C-ABI of the MI355X (gfx950) batched MPC solve engine.
 *
 * This is the drop-in boundary for the hot path of the Spcies solver family: it replaces the call
 * a generated solver's gateway makes into the generated C function,
 *
 *     void laxMPC_ADMM(double *x0_in, double *xr_in, double *ur_in,
 *                      double *u_opt, int *k_in, int *e_flag, sol_<name> *sol);
 *         reference: formulations/+laxMPC/header_laxMPC_ADMM_C.h:27
 *                    (called from formulations/+laxMPC/struct_laxMPC_ADMM_C_Matlab.c:149 and
 *                     examples/cl_in_C/main_cl_in_C.c:103)
 *     void equMPC_ADMM(...same signature...)
 *         reference: formulations/+equMPC/header_equMPC_ADMM_C.h
 *     void laxMPC_FISTA(...same signature...), void equMPC_FISTA(...)
 *         reference: formulations/+laxMPC/header_laxMPC_FISTA_C.h:27, +equMPC/header_equMPC_FISTA_C.h
 *         (sol record holds z and lambda only: pass v = NULL; lambda is [B][N*n])
 *     void ellipMPC_ADMM_soc(double *x0_in, double *xr_in, double *ur_in, double *r_ellip, double *u_opt, ...)
 *         reference: formulations/+ellipMPC/header_ellipMPC_ADMM_soc_C.h:26 (extra input r, 6-field record: _ex)
 *     void HMPC_ADMM(...same signature as laxMPC_ADMM...)   (ADMM or SADMM, split, sparse KKT, box constraints)
 *         reference: formulations/+HMPC/header_HMPC_ADMM_split_C.h:27 (6-field record: _ex)
 *     void MPCT_EADMM(...same signature...)
 *         reference: formulations/+MPCT/header_MPCT_EADMM_C.h:26 (record z1, z2, z3, lambda: _ex entry points)
 *
 * with a batched equivalent: B independent (x0, xr, ur) instances per call.  The reference bakes the
 * controller's constants into the generated C file (`$INSERT_CONSTANTS$`,
 * formulations/+laxMPC/cons_laxMPC_ADMM_C.m:72-130); here they travel in a "problem blob"
 * (layout below) that the host generator packs once per controller design.
 *
 * Conventions
 *  - every entry point returns 0 on success, a negative SPCIES_HIP_E* code otherwise;
 *    spcies_hip_last_error() gives the text (thread-local).
 *  - the caller owns every buffer; the handle owns device constants and scratch
 *    (reference: caller-owned buffers, solver allocates nothing, struct_laxMPC_ADMM_C_Matlab.c:26).
 *  - per-instance results keep the reference semantics: k = iterations run,
 *    e_flag = 1 converged / -1 hit k_max (code_laxMPC_ADMM_C.c:624-631), u = first input move
 *    (:642-651), z / v / lambda = the DEBUG copy-out (:657-686), laid out [B][dim] (instance
 *    contiguous == MATLAB dim x B column-major).
 *  - x0 is [B][n]; xr is [B][n] and ur is [B][m] when ref_stride != 0, else one shared reference
 *    ([n], [m]) for the whole batch.
 *  - no CPU fallback exists: without a usable HIP device the calls fail with SPCIES_HIP_ENODEV.
 *  - ONE handle = ONE solve in flight: a handle owns a single set of device scratch, so calls on the same handle are
 *    serialised (internal mutex) and must not be queued on two streams at once; use one handle per stream (handles are
 *    independent and may live on different devices of one process).  A record field left NULL on a variant whose kernel
 *    writes the whole record (MFMA, MFMA4, BSP, MFMA4R) lands in handle-owned scratch that is allocated on first use:
 *    make one such call before capturing into a hipGraph.
 *  - the device-buffer entry points are asynchronous on the caller's stream EXCEPT for the GEMM variant (HMPC, only when
 *    selected explicitly or when FUSED is unavailable): it copies an "instances still active" count back to the host
 *    every 16 iterations and synchronises the stream, so it is neither asynchronous nor hipGraph-capturable.
 *  - run-time specialised variants (MFMA4 for shapes without a build-time kernel, MFMA4R, BSP, FUSED for other shapes)
 *    compile at spcies_hip_create (seconds); SPCIES_HIP_RTC=0 in the environment switches them off and AUTO then uses
 *    the generic kernels.  spcies_hip_get_info().variant tells which variant a solve will run.
 */
#ifndef SPCIES_HIP_H
#define SPCIES_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPCIES_HIP_ABI_VERSION 1

/* error codes */
#define SPCIES_HIP_OK 0
#define SPCIES_HIP_EINVAL (-1)   /* bad argument / malformed blob           */
#define SPCIES_HIP_ENODEV (-2)   /* no usable HIP device                    */
#define SPCIES_HIP_EHIP (-3)     /* a HIP runtime call failed               */
#define SPCIES_HIP_ENOSUP (-4)   /* formulation / shape / variant not built */
#define SPCIES_HIP_ENOMEM (-5)

/* ---- problem blob ------------------------------------------------------------------------------
 * little-endian; header, then n_arrays directory entries, then 64-byte aligned payloads.        */
#define SPCIES_BLOB_MAGIC "SPCSBLB1"
#define SPCIES_BLOB_VERSION 1u

enum spcies_formulation { SPCIES_LAXMPC = 1, SPCIES_EQUMPC = 2, SPCIES_MPCT = 3, SPCIES_ELLIPMPC = 4, SPCIES_HMPC = 5 };
enum spcies_method { SPCIES_ADMM = 1, SPCIES_FISTA = 2, SPCIES_EADMM = 3, SPCIES_SADMM = 4 };

/* array ids: names are the reference's constant names (cons_laxMPC_ADMM_C.m:82-118) */
enum spcies_array_id {
    SPCIES_A_AB = 1,    /* [n][n+m]       row-major [A B]                                  */
    SPCIES_A_ALPHA = 2, /* [N-1][n][n]    super-diagonal blocks of chol(W)                 */
    SPCIES_A_BETA = 3,  /* [N][n][n]      diagonal blocks, upper triangle, inverted diag   */
    SPCIES_A_HI = 4,    /* [N-1][n+m]                                                      */
    SPCIES_A_HI_0 = 5,  /* [m]                                                             */
    SPCIES_A_HI_N = 6,  /* [n][n]                                                          */
    SPCIES_A_Q = 7,     /* [n]   negated diag(Q)                                           */
    SPCIES_A_R = 8,     /* [m]   negated diag(R)                                           */
    SPCIES_A_T = 9,     /* [n][n] negated T                                                */
    SPCIES_A_LB = 10,   /* [n+m]                                                           */
    SPCIES_A_UB = 11,   /* [n+m]                                                           */
    /* FISTA solvers (cons_laxMPC_FISTA_C.m:94-107) */
    SPCIES_A_QRI = 12,  /* [n+m] -1/diag([Q, R])                                           */
    SPCIES_A_TDIAG = 13,/* [n]   negated diag(T)                                           */
    SPCIES_A_TI = 14,   /* [n]   -1/diag(T)                                                */
    /* MPCT EADMM (cons_MPCT_EADMM_C.m:82-100); T (id 9) is [n][n] negated, LB/UB ids 10/11 */
    SPCIES_A_S = 15,       /* [m][m] negated S                                             */
    SPCIES_A_RHO_MAT = 16, /* [N+1][n+m]                                                   */
    SPCIES_A_RHO_0 = 17,   /* [n+m]                                                        */
    SPCIES_A_RHO_S = 18,   /* [n+m]                                                        */
    SPCIES_A_LB_0 = 19, SPCIES_A_UB_0 = 20, SPCIES_A_LB_S = 21, SPCIES_A_UB_S = 22, /* [n+m] each */
    SPCIES_A_H1I = 23,     /* [N+1][n+m]                                                   */
    SPCIES_A_W2 = 24,      /* [n+m][n+m]                                                   */
    SPCIES_A_H3I = 25,     /* [N+1][n+m]                                                   */
    /* ellipMPC ADMM soc (cons_ellipMPC_ADMM_soc_C.m:82-110).  Here Q (7), R (8), T (9) are dense negated
     * [n][n] / [m][m] / [n][n]; LB / UB (10 / 11) have dim-n-1 entries; index arrays are dtype 1 (i32), 0-based. */
    SPCIES_A_A = 26, SPCIES_A_PHIP = 27,                                  /* [n][n] each               */
    SPCIES_A_L_VAL = 28, SPCIES_A_L_COL = 29, SPCIES_A_L_ROW = 30,        /* CSC of L - I              */
    SPCIES_A_DINV = 31,                                                   /* [n_eq + n_s]              */
    SPCIES_A_GHHHI_VAL = 32, SPCIES_A_GHHHI_COL = 33, SPCIES_A_GHHHI_ROW = 34, /* CSR of -Gh Hh^-1     */
    SPCIES_A_HHIGH_VAL = 35, SPCIES_A_HHIGH_COL = 36, SPCIES_A_HHIGH_ROW = 37, /* CSR of -Hh^-1 Gh'    */
    SPCIES_A_HHI_VAL = 38, SPCIES_A_HHI_COL = 39, SPCIES_A_HHI_ROW = 40,  /* CSR of -Hh^-1             */
    /* HMPC ADMM / SADMM split (cons_HMPC_ADMM_split_C.m:121-142): A (26), QQ as Q (7, dense [n][n]), LB / UB with
     * dim-3(n+m) entries, L_* / Dinv (28-31) for the KKT factor; header flags bit1 = use_soc,
     * reserved = {sigma, 1/sigma, tol_d, alpha_SADMM} */
    SPCIES_A_TE = 41, SPCIES_A_SE = 42,      /* [n][n], [m][m]                                      */
    SPCIES_A_LBY = 43, SPCIES_A_UBY = 44,    /* [n+m]                                               */
    SPCIES_A_IDX_X0 = 45,                    /* i32 [n]: rows of bh that hold -A x0                 */
    SPCIES_A_BH = 46,                        /* [n_eq + n_s]                                        */
    /* time-varying lax/equ MPC ADMM (header flags bit2; cons_laxMPC_ADMM_C.m:97-109): the blob holds only T (9)
     * and T_rho_i = inv(T + rho I); A, B, Q, R, LB, UB arrive with every call through the `extra` argument of
     * the _ex entry points, packed per instance as A [n*n] and B [n*m] COLUMN-major, Q [n], R [m], LB [n+m],
     * UB [n+m] (the six extra inputs of the 9-argument mex gateway, struct_laxMPC_ADMM_C_Matlab.c:57-103);
     * extra_stride = n*n + n*m + n + m + 2(n+m) for one model per instance, 0 for one shared model          */
    SPCIES_A_T_RHO_I = 47,                   /* [n][n]                                              */
    /* option in_engineering (header flags bit3; cons_laxMPC_ADMM_C.m:110-116): the solver's arguments are in
     * engineering units: x = scaling_x o (x_in - OpPoint_x) for x0 and xr, ur likewise with scaling_u / OpPoint_u,
     * u_out = v_0 o scaling_i_u + OpPoint_u (code_laxMPC_ADMM_C.c:83-100, 642-646)                          */
    SPCIES_A_SCALING_X = 48,                 /* [n]                                                 */
    SPCIES_A_SCALING_U = 49,                 /* [m]                                                 */
    SPCIES_A_SCALING_I_U = 50,               /* [m]                                                 */
    SPCIES_A_OPPOINT_X = 51,                 /* [n]                                                 */
    SPCIES_A_OPPOINT_U = 52,                 /* [m]                                                 */
    /* ellipMPC ADMM with the P-projection onto the terminal ellipsoid (formulation 4, submethod 0;
     * cons_ellipMPC_ADMM_C.m:74-110): the lax arrays 1-9 (no LB / UB) plus these; r in header reserved[4]   */
    SPCIES_A_P = 53,                         /* [n][n]                                              */
    SPCIES_A_P_HALF = 54,                    /* [n][n] sqrtm(P)                                     */
    SPCIES_A_PINV_HALF = 55,                 /* [n][n] P^-1 P_half                                  */
    SPCIES_A_C_ELL = 56,                     /* [n] centre of the ellipsoid                         */
    SPCIES_A_LBZ = 57,                       /* [N-1][n+m]                                          */
    SPCIES_A_UBZ = 58,                       /* [N-1][n+m]                                          */
    SPCIES_A_LBU0 = 59,                      /* [m]                                                 */
    SPCIES_A_UBU0 = 60,                      /* [m]                                                 */
    /* lax/equ MPC ADMM switches.  Vector rho (header flags bit0 clear; cons_laxMPC_ADMM_C.m:123-129): rho_0 (17)
     * [m], these, and Hi computed with them.  VAR_BOUNDS (flags bit4; :82-90): LB0 / UB0 (19, 20) [m], LB / UB
     * (10, 11) become [N-1][n+m], LBN / UBN [n]                                                             */
    SPCIES_A_RHO_V = 61,                     /* [N-1][n+m]                                          */
    SPCIES_A_RHO_N = 62,                     /* [n]                                                 */
    SPCIES_A_RHO_I_V = 63,                   /* [N-1][n+m] reciprocals as the generator prints them */
    SPCIES_A_RHO_I_0 = 64,                   /* [m]                                                 */
    SPCIES_A_RHO_I_N = 65,                   /* [n]                                                 */
    SPCIES_A_LBN = 66,                       /* [n]                                                 */
    SPCIES_A_UBN = 67,                       /* [n]                                                 */
    /* HMPC split, NON_SPARSE path (the reference's default option sparse = false; compute_HMPC_ADMM_split_
     * ingredients.m:236-239, code_HMPC_ADMM_split_C.c:174-190): optional, enable the GEMM variant                */
    SPCIES_A_M1 = 68,                        /* [dim + n_s][dim + n_s]                              */
    SPCIES_A_M2 = 69,                        /* [dim + n_s][n_eq + n_s]                             */
    SPCIES_A_BH_NAT = 70,                    /* [n_eq + n_s] bh in natural order (x0 rows first)    */
    /* HMPC ADMM / SADMM WITHOUT the splitting (formulation 5, submethod 0 - the reference's default HMPC solver;
     * cons_HMPC_ADMM_C.m:88-131): A (26), QQ as Q (7), Te, Se, LBy, UBy (41-44), LB / UB (10, 11) with
     * n_box = dim - 3(n+m) entries, M1 (68) [dim][dim], M2 (69) [dim][n], and these; n_s = n_box + 3 n_soc;
     * header flags bit1 = use_soc, reserved = {-, -, tol_d, alpha_SADMM}.  Record: z [dim], s [n_s], lambda [n_s] */
    SPCIES_A_C_VAL = 71, SPCIES_A_C_COL = 72, SPCIES_A_C_ROW = 73,    /* CSR of C  [n_s x dim] (0-based)  */
    SPCIES_A_CT_VAL = 74, SPCIES_A_CT_COL = 75, SPCIES_A_CT_ROW = 76, /* CSR of C' [dim x n_s]            */
    SPCIES_A_D = 77,                         /* [n_s] d (read with use_soc only)                    */
    /* MPCT ADMM on the extended state space (formulation 3, method ADMM, submethod 3 = 'cs'; cons_MPCT_ADMM_cs_C.m:66-112):
     * LB / UB (10, 11) [2N(n+m)], L_* / Dinv (28-31) with nrow = 2n + (2n+m)(N-1) + n rows, and these; header flags bit0 =
     * scalar rho (then rho, rho_i come from the header).  Record: z, v, lambda [2N(n+m)]                              */
    SPCIES_A_TZ = 78, SPCIES_A_SZ = 79,      /* [n][n] = -T/N, [m][m] = -S/N                        */
    SPCIES_A_AHI_VAL = 80, SPCIES_A_AHI_COL = 81, SPCIES_A_AHI_ROW = 82, /* CSR of -Aeq Hhat^-1  [nrow x dim] */
    SPCIES_A_HIA_VAL = 83, SPCIES_A_HIA_COL = 84, SPCIES_A_HIA_ROW = 85, /* CSR of -Hhat^-1 Aeq' [dim x nrow] */
    SPCIES_A_HI_VAL = 86, SPCIES_A_HI_COL = 87, SPCIES_A_HI_ROW = 88,    /* CSR of -Hhat^-1      [dim x dim]  */
    SPCIES_A_RHO_CS = 89, SPCIES_A_RHO_I_CS = 90,                         /* [dim] (vector rho)               */
    /* MPCT EADMM with general (non-diagonal) Q, R - header flags bit5; IS_DIAG == 0 of the generated solver
     * (cons_MPCT_EADMM_C.m:99-108, code_MPCT_EADMM_C.c:184-217, 321-366): these six INSTEAD of H3i (25)          */
    SPCIES_A_Q_BI = 91, SPCIES_A_Q_MI = 92,   /* [n][n]   (Q + rho_base I)^-1, (Q + rho_mult rho_base I)^-1  */
    SPCIES_A_R_BI = 93, SPCIES_A_R_MI = 94,   /* [m][m]                                                       */
    SPCIES_A_AB_BI = 95, SPCIES_A_AB_MI = 96  /* [n][n+m] [A B] blkdiag(Q_bi, R_bi), [A B] blkdiag(Q_mi, R_bi) */
};

typedef struct {
    char magic[8];         /* SPCIES_BLOB_MAGIC                              */
    uint32_t version;      /* SPCIES_BLOB_VERSION                            */
    uint32_t header_bytes; /* sizeof(spcies_blob_header) = 128               */
    uint32_t formulation;  /* enum spcies_formulation                        */
    uint32_t method;       /* enum spcies_method                             */
    uint32_t submethod;    /* 0 = none                                       */
    uint32_t flags;        /* bit0: scalar rho, bit1: use_soc, bit2: time-varying, bit3: in_engineering, bit4: VAR_BOUNDS, bit5: general Q, R (MPCT EADMM), bit6: HMPC coupled output constraints */
    uint32_t n, m, N, k_max;
    uint32_t n_arrays;
    uint32_t reserved0;
    uint64_t total_bytes;
    double tol, rho, rho_i;
    double reserved[5];    /* ellipMPC soc: [0] sigma, [1] 1/sigma, [2] tol_d (tol is tol_p)   */
} spcies_blob_header;

typedef struct {
    uint32_t id;     /* enum spcies_array_id      */
    uint32_t dtype;  /* 0 = f64, 1 = i32          */
    uint64_t offset; /* from blob start, 64-B aligned */
    uint64_t count;  /* elements                  */
    uint32_t dims[4];
    uint32_t pad[2];
} spcies_blob_entry; /* 48 bytes */

/* ---- engine ------------------------------------------------------------------------------------ */
typedef struct spcies_hip_solver_s *spcies_hip_handle;

/* kernel variants (spcies_hip_set_variant) */
#define SPCIES_VARIANT_AUTO 0
#define SPCIES_VARIANT_STREAM 1 /* one lane per instance, reference operation order, state streamed through HBM */
#define SPCIES_VARIANT_MFMA 2   /* 16 instances per wavefront on v_mfma_f64_16x16x4, state in registers          */
#define SPCIES_VARIANT_MFMA4 3  /* same on v_mfma_f64_4x4x4 (4 blocks): no row padding, zero blocks skipped      */
#define SPCIES_VARIANT_MFMA4G 4 /* v_mfma_f64_4x4x4 with a rolled stage loop: any N, state streamed through HBM   */
#define SPCIES_VARIANT_TILE 5   /* sparse-KKT solvers: 4-64 lanes per instance, LDL right-hand side in LDS          */
#define SPCIES_VARIANT_GEMM 6   /* HMPC (split NON_SPARSE path; no-split solver): one dgemm per iteration for the batch */
#define SPCIES_VARIANT_BSP 7    /* ellipMPC ADMM / soc: the KKT iteration as a per-controller program of 4x4 MFMA blocks; also laxMPC / equMPC ADMM
                                 * (AUTO's choice there with vector rho or stage-wise bounds, on request otherwise) */
#define SPCIES_VARIANT_FUSED 8  /* HMPC dense paths and MPCT ADMM cs: product, projections, duals and exit test in one v_mfma_f64_4x4x4 kernel */
#define SPCIES_VARIANT_MFMA4R 9 /* FISTA, MPCT EADMM, lax / equ ADMM (AUTO past MFMA4's register file), time-varying ADMM / FISTA:
                                   v_mfma_f64_4x4x4, unrolled on the horizon, iteration state in registers + LDS.  Time-varying: one wavefront per
                                   instance, its factors in registers (n + m <= 16) or - past the register file, n + m <= 32 - in the LDS */
/* Integer outputs.  STREAM runs the reference's operation order and returns its k / e_flag bit for bit.  The other variants
 * re-associate sums (1e-10 on the iterates): an exit test decided within rounding may fire one iteration apart on < 0.1 % of
 * instances.  tol = 0 is the reference tests' fixed-iteration setting (k = k_max, e_flag = -1): MPCT ADMM cs FUSED, whose
 * w-form reaches an exact floating-point fixed point the reference order does not, switches its exit test off after the first
 * iteration when tol <= 0, so it returns (k_max, -1) like the reference (code_MPCT_ADMM_cs_C.c:196-215); the all-zero instance
 * still returns (1, 1) as there. */

typedef struct {
    int formulation, method, submethod;
    int n, m, N, dim; /* dim = length of z / v / lambda per instance */
    int k_max;
    double tol, rho;
    int variant;      /* variant a solve would use now   */
    int device;
    int dim_lambda;   /* length of the lambda output per instance: dim (ADMM), N*n (FISTA: the dual y) */
} spcies_hip_info;

/* Batch-level mirror of the reference's four timers (docs/timing.md:9-22, sol_<name> fields of
 * header_laxMPC_ADMM_C.h:18-21), in milliseconds: update = H2D of inputs, solve = kernel,
 * polish = D2H of outputs, run = whole call. */
typedef struct {
    double update_time, solve_time, polish_time, run_time;
} spcies_hip_timing;

int spcies_hip_abi_version(void);
const char *spcies_hip_last_error(void);
int spcies_hip_device_count(int *count);

/* Parse + validate the blob, upload constants to `device`, derive the kernel-side packing. */
int spcies_hip_create(const void *blob, size_t bytes, int device, spcies_hip_handle *out);
int spcies_hip_destroy(spcies_hip_handle h);
int spcies_hip_get_info(spcies_hip_handle h, spcies_hip_info *info);
int spcies_hip_set_variant(spcies_hip_handle h, int variant);
/* A faster variant that AUTO cannot use is not an error: AUTO uses the next one.  *notes (owned by the handle, "" when nothing
 * was given up) says which faster variants are unavailable and why - both when the variant does not apply to the controller by
 * design (general Q and R, a shape outside the packer, state beyond registers + LDS, run-time specialisation switched off) and
 * when it applies but could not be built (hiprtc missing, compile error); SPCIES_HIP_VERBOSE=1 in the environment prints the
 * same line to stderr at create time.  SPCIES_HIP_STRICT=1 turns a FAILED BUILD - and only that - into an error of
 * spcies_hip_create (SPCIES_HIP_ENOSUP): for deployments that must not run on a slower variant unnoticed. */
int spcies_hip_get_notes(spcies_hip_handle h, const char **notes);
/* Override the blob's exit settings (the reference bakes them in as #defines k_max / tol,
 * cons_laxMPC_ADMM_C.m:76-77).  tol < 0 keeps the current value; k_max <= 0 keeps the current value. */
int spcies_hip_set_exit(spcies_hip_handle h, int k_max, double tol);
/* Pre-size device scratch for batches up to B instances (otherwise grown on demand). */
int spcies_hip_reserve(spcies_hip_handle h, long B);

/* Host-buffer entry point (what a mex / cgo / ctypes gateway binds).  z, v, lambda, timing may be NULL. */
int spcies_hip_solve_batch(spcies_hip_handle h, const double *x0, const double *xr, const double *ur,
                           int ref_stride, long B, double *u, int *k, int *e_flag, double *z, double *v,
                           double *lambda, spcies_hip_timing *timing);

/* Device-buffer entry point: all pointers are device memory on the handle's device; the launch is
 * asynchronous on `stream` (a hipStream_t, NULL = default stream).  Scratch must already be large
 * enough (spcies_hip_reserve) if the call is to be captured in a hipGraph. */
int spcies_hip_solve_batch_device(spcies_hip_handle h, const double *x0, const double *xr, const double *ur,
                                  int ref_stride, long B, double *u, int *k, int *e_flag, double *z,
                                  double *v, double *lambda, void *stream);

/* Solvers whose record is not (z, v, lambda) - and any solver, uniformly: `fields` holds one pointer per
 * field of the generated solver's sol_<name> struct, in the reference's order
 *     ADMM (lax/equ): z, v, lambda     FISTA: z, lambda     MPCT-EADMM: z1, z2, z3, lambda
 *     ellipMPC-ADMM-soc and HMPC-(S)ADMM-split: z, s, z_hat, s_hat, lambda, mu
 * (header_laxMPC_ADMM_C.h:14-22, header_laxMPC_FISTA_C.h:14-21, header_MPCT_EADMM_C.h:14-23,
 * header_ellipMPC_ADMM_soc_C.h:14-24); a NULL entry (or fields == NULL) skips that output.
 * `extra` carries formulation-specific extra inputs: the ellipsoid radius r of ellipMPC soc
 * (4th argument of ellipMPC_ADMM_soc, header_ellipMPC_ADMM_soc_C.h:26) as [B] (extra_stride = 1) or one
 * shared value (extra_stride = 0); NULL for the other solvers.  spcies_hip_get_sol_layout reports count, per-instance lengths, names. */
int spcies_hip_get_sol_layout(spcies_hip_handle h, int *n_fields, int *dims, const char **names);
int spcies_hip_solve_batch_ex(spcies_hip_handle h, const double *x0, const double *xr, const double *ur,
                              int ref_stride, const double *extra, int extra_stride, long B, double *u, int *k,
                              int *e_flag, double *const *fields, int n_fields, spcies_hip_timing *timing);
int spcies_hip_solve_batch_device_ex(spcies_hip_handle h, const double *x0, const double *xr, const double *ur,
                                     int ref_stride, const double *extra, int extra_stride, long B, double *u, int *k,
                                     int *e_flag, double *const *fields, int n_fields, void *stream);

/* Doubles per instance of the `extra` input when extra_stride != 0: 1 (ellipMPC radius), or the packed model
 * (A, B, Q, R, LB, UB; struct_laxMPC_ADMM_C_Matlab.c:57-103) of a time-varying solver. */
int spcies_hip_get_extra_width(spcies_hip_handle h, long *doubles_per_instance);

/* Page-locked host buffers visible to every device of the process (hipHostMalloc, portable): what a gateway hands the
 * host-buffer entry points when it wants the H2D / D2H copies at full PCIe rate and overlapped across devices. */
int spcies_hip_host_alloc(size_t bytes, void **ptr);
int spcies_hip_host_free(void *ptr);

/* Run-time specialised kernels (hiprtc) are compiled once per controller and MACHINE: code objects are served from a bounded
 * in-memory cache (SPCIES_HIP_RTC_CACHE_MB, default 256), then from the on-disk cache $SPCIES_HIP_CACHE_DIR |
 * $XDG_CACHE_HOME/spcies_hip | $HOME/.cache/spcies_hip (<sha256 of compiler identity, options, source>.hsaco, written
 * atomically, compiled under flock: the ranks of a multi-GPU job, a later MATLAB session or test process read the file instead
 * of compiling; SPCIES_HIP_DISK_CACHE=0 switches the disk off).  *hits = served from memory or disk, *misses = compilations. */
int spcies_hip_rtc_cache_stats(long *hits, long *misses);
/* out[0..n): memory hits, disk hits, compilations, evictions from memory, files written, failed file writes (n <= 6 used). */
int spcies_hip_rtc_cache_stats_ex(long *out, int n);
/* Test hook (no GPU, no hiprtc): runs the cache machinery above with a stand-in compiler that takes work_ms milliseconds
 * (work_ms < 0: fails) and returns bytes that depend on `text` only.  *source: 0 memory, 1 disk, 2 compiled; *checksum of the
 * code object handed back; drop_memory != 0 empties the in-memory cache first.  tests/test_rtc_disk_cache.py. */
int spcies_hip_rtc_cache_selftest(const char *text, int work_ms, int drop_memory, int *source, unsigned long long *checksum);
/* Test hook of the compiler process (spcies_amd/spcies_rtc_helper, rtc_helper.cpp: since round 5 hiprtc runs in a process of its own,
 * so that a compiler crash fails one build instead of taking the caller down; SPCIES_HIP_RTC_ISOLATE=0 compiles in-process).  Compiles
 * `src` - which must define `extern "C" __global__ void selftest_kernel(...)` - for gfx950 in the helper, bypassing both caches, without
 * loading the result (no device needed).  *isolated = 1 when the helper did it, 0 when no helper is available (then SPCIES_HIP_ENOSUP). */
int spcies_hip_rtc_compile_selftest(const char *src, int *isolated, unsigned long *code_bytes);

/* Batch statistics of a solve (the batch counterpart of the dense MATLAB solvers' genHist record,
 * platforms/Matlab/spcies_laxMPC_ADMM_solver.m:253-261): k, e_flag are DEVICE arrays [B] as a device solve left them;
 * hist[b], b < n_bins, counts the instances with b k_max / n_bins < k <= (b + 1) k_max / n_bins (host array);
 * counts[0..3] = { e_flag > 0, e_flag == -1, other, sum of k } (host array of 4). */
int spcies_hip_k_histogram_device(spcies_hip_handle h, const int *k, const int *e_flag, long B, int n_bins, long *hist, long *counts,
                                  void *stream);

/* Residual history of a solve (what the dense MATLAB solvers record with genHist > 0: hRp(k) = ||z - v||_inf, hRd(k) =
 * ||v - v_prev||_inf, platforms/Matlab/spcies_laxMPC_ADMM_solver.m:253-261, 311-319) for the lax / equ MPC ADMM solvers: host
 * buffers; r_p, r_d are [B][K], entry [i][j-1] = the residuals of iteration j of instance i, zero behind the iteration the instance
 * leaves at under the handle's own exit settings (returned in k_exit [B], may be NULL).  A diagnostic: iteration j is read off a
 * solve stopped at k_max = j on the handle's current variant (K (K + 1) / 2 iterations in all); the handle's settings are restored. */
int spcies_hip_residual_trace(spcies_hip_handle h, const double *x0, const double *xr, const double *ur, int ref_stride, long B, int K,
                              double *r_p, double *r_d, int *k_exit);

/* Time `reps` back-to-back device solves with hipEvents recorded on `stream` (the stream the
 * kernel is launched on); returns the mean milliseconds per launch in *ms_per_launch. */
int spcies_hip_time_device(spcies_hip_handle h, const double *x0, const double *xr, const double *ur,
                           int ref_stride, long B, double *u, int *k, int *e_flag, void *stream, int reps,
                           double *ms_per_launch);

/* Closed-loop batch simulation on the device (examples/cl_in_C/main_cl_in_C.c:98-117: solve, then
 * x+ = A x + B u with the plant AB [n][n+m] row-major, `steps` times) for B independent plants, without a
 * host round trip between sample times.  Host buffers: x0 [B][n] initial states, xr / ur as in solve_batch;
 * outputs (each may be NULL): x_traj [steps+1][B][n] (x_traj[0] = x0), u_traj [steps][B][m],
 * k_traj / e_traj [steps][B].  Solvers with extra inputs (ellipMPC radius, time-varying model) are not driven
 * by this entry point.                                                                                      */
int spcies_hip_closed_loop(spcies_hip_handle h, const double *AB_plant, const double *x0, const double *xr, const double *ur,
                           int ref_stride, long B, int steps, double *x_traj, double *u_traj, int *k_traj, int *e_traj,
                           spcies_hip_timing *timing);

/* ---- several devices from ONE process (no launcher, no torch): what a mex gateway or a plain-C caller binds to use every
 * GPU of a node (examples/cl_in_C/main_cl_in_C.c:103 use case, batched).  One single-device handle per entry of device_ids
 * (NULL = 0 .. n_dev-1; n_dev <= 0 = every visible device), all created from the same blob - each device parses and packs
 * it for itself, no collective; a solve splits the host batch into contiguous shards (spcies_hip_shard_range: sizes differ
 * by at most one, larger shards first - the split of spcies_amd/distributed.py), one host thread per device, each running
 * spcies_hip_solve_batch_ex on its shard; results land in the caller's buffers at the shard's offset.  A device may be
 * listed more than once (two handles on one GPU).  timing: per phase the slowest device, run_time the whole call.        */
typedef struct spcies_hip_multi_s *spcies_hip_multi_handle;
int spcies_hip_shard_range(long B, int n_shards, int shard, long *begin, long *count);
int spcies_hip_create_multi(const void *blob, size_t bytes, const int *device_ids, int n_dev, spcies_hip_multi_handle *out);
int spcies_hip_multi_destroy(spcies_hip_multi_handle m);
int spcies_hip_multi_count(spcies_hip_multi_handle m, int *n_dev);
int spcies_hip_multi_get(spcies_hip_multi_handle m, int i, spcies_hip_handle *single); /* per-device handle (info, reserve) */
int spcies_hip_multi_set_variant(spcies_hip_multi_handle m, int variant);
int spcies_hip_multi_set_exit(spcies_hip_multi_handle m, int k_max, double tol);
int spcies_hip_multi_solve_batch(spcies_hip_multi_handle m, const double *x0, const double *xr, const double *ur, int ref_stride,
                                 long B, double *u, int *k, int *e_flag, double *z, double *v, double *lambda,
                                 spcies_hip_timing *timing);
/* extra_width: doubles per instance in `extra` when extra_stride != 0 - spcies_hip_get_extra_width's value (1 for the
 * ellipMPC radius; the packed model of a time-varying solver); <= 0 means "that value", anything else that differs from it is
 * rejected with SPCIES_HIP_EINVAL.  NULL x0 / xr / ur / u / k / e_flag are rejected like in the single-device entry point. */
int spcies_hip_multi_solve_batch_ex(spcies_hip_multi_handle m, const double *x0, const double *xr, const double *ur, int ref_stride,
                                    const double *extra, int extra_stride, long extra_width, long B, double *u, int *k,
                                    int *e_flag, double *const *fields, int n_fields, spcies_hip_timing *timing);

#ifdef __cplusplus
}
#endif
#endif /* SPCIES_HIP_H */
