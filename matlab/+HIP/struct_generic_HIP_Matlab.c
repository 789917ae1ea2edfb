// Generic mex gateway of the HIP platform: one source for every solver the engine implements.
//
// Counterpart of the reference's struct_<formulation>_<method>_C_Matlab.c gateways:
//     [u, k, e_flag, sol] = $INSERT_NAME$(x0, xr, ur)                          lax/equ ADMM and FISTA, MPCT EADMM, ellipMPC ADMM,
//                                                                           HMPC ADMM / SADMM split
//     [u, k, e_flag, sol] = $INSERT_NAME$(x0, xr, ur, r)                       ellipMPC ADMM soc  (struct_ellipMPC_ADMM_soc_C_Matlab.c:24)
//     [u, k, e_flag, sol] = $INSERT_NAME$(x0, xr, ur, A, B, Q, R, LB, UB)      time-varying lax/equ ADMM (struct_laxMPC_ADMM_C_Matlab.c:29-31)
// with the reference's argument checks and error ids.  Extension: x0 may be n x B (one instance per column,
// MATLAB column-major == the engine's [B][n] layout); xr / ur n x 1 / m x 1 (shared) or n x B / m x B; r 1 x 1 or
// 1 x B; the model matrices n x n [x B] etc.  Outputs then gain a trailing batch dimension.  The record `sol` has
// the fields of the generated solver (spcies_hip_get_sol_layout) plus the four timings.
//
// NN_MODEL_ is 0, 1 (extra input r) or 6 (time-varying model); the blob is written next to the mex by the
// constructor (HIP.cons_generic) and loaded on the first call; the engine handle lives until the mex is cleared.
//
// Device selection (read once, at the first call; `clear <mex>` to change it):
//     SPCIES_HIP_DEVICE=<i>            one GPU, device i (default 0)
//     SPCIES_HIP_DEVICES=all | i,j,... several GPUs from this one MATLAB process (spcies_hip_create_multi: one host thread and one
//                                      handle per device, the batch split into contiguous shards, no collective)
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"
#include "spcies_hip.h"

$INSERT_DEFINES$ /* nn_, mm_, N_EXTRA_ (0, 1 or 6) and BLOB_PATH, as HIP.cons_generic prints them */

static spcies_hip_handle g_handle = NULL;      /* device 0 of g_multi when several devices are used */
static spcies_hip_multi_handle g_multi = NULL; /* SPCIES_HIP_DEVICES */

static void at_exit(void) {
    if (g_multi) spcies_hip_multi_destroy(g_multi);
    else if (g_handle) spcies_hip_destroy(g_handle);
    g_handle = NULL;
    g_multi = NULL;
}

static void ensure_handle(void) {
    if (g_handle) return;
    FILE *f = fopen(BLOB_PATH, "rb");
    if (!f) mexErrMsgIdAndTxt("Spcies:$FORM$:HIP:blob", "cannot open problem blob %s", BLOB_PATH);
    fseek(f, 0, SEEK_END);
    long bytes = ftell(f);
    fseek(f, 0, SEEK_SET);
    void *blob = mxMalloc((size_t)bytes);
    if (fread(blob, 1, (size_t)bytes, f) != (size_t)bytes) {
        fclose(f);
        mexErrMsgIdAndTxt("Spcies:$FORM$:HIP:blob", "short read on %s", BLOB_PATH);
    }
    fclose(f);
    int rc;
    const char *many = getenv("SPCIES_HIP_DEVICES"), *one = getenv("SPCIES_HIP_DEVICE");
    if (many && *many) {
        int ids[64], n_ids = 0;
        if (strcmp(many, "all") != 0) {
            const char *c = many;
            while (*c && n_ids < 64) {
                char *end = NULL;
                long d = strtol(c, &end, 10);
                if (end == c || d < 0) {
                    mxFree(blob);
                    mexErrMsgIdAndTxt("Spcies:$FORM$:HIP:devices", "SPCIES_HIP_DEVICES must be 'all' or a comma-separated list of device indices");
                }
                ids[n_ids++] = (int)d;
                c = (*end == ',') ? end + 1 : end;
                if (*end && *end != ',') break;
            }
        }
        rc = spcies_hip_create_multi(blob, (size_t)bytes, n_ids ? ids : NULL, n_ids, &g_multi);
        if (!rc) rc = spcies_hip_multi_get(g_multi, 0, &g_handle);
    } else {
        rc = spcies_hip_create(blob, (size_t)bytes, (one && *one) ? atoi(one) : 0, &g_handle);
    }
    mxFree(blob);
    if (rc) {
        if (g_multi) spcies_hip_multi_destroy(g_multi);
        g_multi = NULL;
        g_handle = NULL;
        mexErrMsgIdAndTxt("Spcies:$FORM$:HIP:create", "%s", spcies_hip_last_error());
    }
    mexAtExit(at_exit);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs != 3 + N_EXTRA_)
        mexErrMsgIdAndTxt("Spcies:$FORM$:nrhs:number", N_EXTRA_ == 6 ? "Nine inputs are required"
                                                          : (N_EXTRA_ == 1 ? "Four inputs are required" : "Three inputs are required"));
    if (nlhs == 0) mexErrMsgIdAndTxt("Spcies:$FORM$:nlhs:number", "At least one output is required");
    if (!mxIsDouble(prhs[0]) || mxGetNumberOfElements(prhs[0]) % nn_ != 0 || mxGetNumberOfElements(prhs[0]) == 0)
        mexErrMsgIdAndTxt("Spcies:$FORM$:nrhs:x0", "x0 must be of dimension %d (or %d x B)", nn_, nn_);
    const long B = (long)(mxGetNumberOfElements(prhs[0]) / nn_);
    const size_t nxr = mxGetNumberOfElements(prhs[1]), nur = mxGetNumberOfElements(prhs[2]);
    const int per_instance = (B > 1 && nxr == (size_t)nn_ * B);
    if (!mxIsDouble(prhs[1]) || !(nxr == nn_ || nxr == (size_t)nn_ * B))
        mexErrMsgIdAndTxt("Spcies:$FORM$:nrhs:xr", "xr must be of dimension %d", nn_);
    if (!mxIsDouble(prhs[2]) || nur != (per_instance ? (size_t)mm_ * B : (size_t)mm_))
        mexErrMsgIdAndTxt("Spcies:$FORM$:nrhs:ur", "ur must be of dimension %d", mm_);
    ensure_handle();

    /* extra inputs: r (1 value, shared or per instance) or the packed model A, B, Q, R, LB, UB */
    double *extra = NULL;
    int extra_stride = 0, extra_owned = 0;
#if N_EXTRA_ == 1
    {
        const size_t nr = mxGetNumberOfElements(prhs[3]);
        if (!mxIsDouble(prhs[3]) || !(nr == 1 || nr == (size_t)B))
            mexErrMsgIdAndTxt("Spcies:$FORM$:nrhs:r", "r must be a scalar (or one value per instance)");
        extra = mxGetPr(prhs[3]);
        extra_stride = (nr == (size_t)B && B > 1) ? 1 : 0;
    }
#elif N_EXTRA_ == 6
    {
        const size_t want[6] = {(size_t)nn_ * nn_, (size_t)nn_ * mm_, nn_, mm_, nn_ + mm_, nn_ + mm_};
        const char *names[6] = {"A", "B", "Q", "R", "LB", "UB"};
        size_t width = 0, per = 0;
        for (int i = 0; i < 6; i++) {
            const size_t ne = mxGetNumberOfElements(prhs[3 + i]);
            if (!mxIsDouble(prhs[3 + i]) || !(ne == want[i] || ne == want[i] * (size_t)B)) {
                char id[64];
                snprintf(id, sizeof(id), "Spcies:$FORM$:nrhs:%s", names[i]);
                mexErrMsgIdAndTxt(id, "%s must hold %d elements (or that many per instance)", names[i], (int)want[i]);
            }
            if (ne != want[i]) per = 1;
            width += want[i];
        }
        const size_t rows = per ? (size_t)B : 1;
        extra = (double *)mxMalloc(sizeof(double) * rows * width);
        extra_owned = 1;
        extra_stride = per ? (int)width : 0;
        for (size_t b = 0; b < rows; b++) {
            size_t at = 0;
            for (int i = 0; i < 6; i++) {  /* MATLAB hands A, B column-major: exactly the layout the engine expects */
                const size_t ne = mxGetNumberOfElements(prhs[3 + i]);
                const double *src = mxGetPr(prhs[3 + i]) + (ne == want[i] ? 0 : b * want[i]);
                memcpy(extra + b * width + at, src, sizeof(double) * want[i]);
                at += want[i];
            }
        }
    }
#endif

    int n_fields = 0, dims[8];
    const char *names[8];
    if (spcies_hip_get_sol_layout(g_handle, &n_fields, dims, names))
        mexErrMsgIdAndTxt("Spcies:$FORM$:HIP:layout", "%s", spcies_hip_last_error());
    plhs[0] = mxCreateDoubleMatrix(mm_, B, mxREAL);
    mxArray *k_d = mxCreateDoubleMatrix(1, B, mxREAL), *e_d = mxCreateDoubleMatrix(1, B, mxREAL);
    int *k = (int *)mxMalloc(sizeof(int) * B), *e = (int *)mxMalloc(sizeof(int) * B);
    const char *field_names[12];
    for (int i = 0; i < n_fields; i++) field_names[i] = names[i];
    field_names[n_fields] = "update_time"; field_names[n_fields + 1] = "solve_time";
    field_names[n_fields + 2] = "polish_time"; field_names[n_fields + 3] = "run_time";
    mxArray *sol = mxCreateStructMatrix(1, 1, n_fields + 4, field_names);
#ifdef DEBUG
    double *fields[8] = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, NULL};
    for (int i = 0; i < n_fields; i++) {
        mxArray *a = mxCreateDoubleMatrix(dims[i], B, mxREAL);
        fields[i] = mxGetPr(a);
        mxSetField(sol, 0, names[i], a);
    }
#endif
    spcies_hip_timing t;
#ifdef DEBUG
    double *const *fields_arg = fields;
#else
    double *const *fields_arg = NULL;
#endif
    int rc = g_multi ? spcies_hip_multi_solve_batch_ex(g_multi, mxGetPr(prhs[0]), mxGetPr(prhs[1]), mxGetPr(prhs[2]), per_instance, extra,
                                                       extra_stride, 0 /* the solver's own width */, B, mxGetPr(plhs[0]), k, e,
                                                       fields_arg, n_fields, &t)
                     : spcies_hip_solve_batch_ex(g_handle, mxGetPr(prhs[0]), mxGetPr(prhs[1]), mxGetPr(prhs[2]), per_instance, extra,
                                                 extra_stride, B, mxGetPr(plhs[0]), k, e, fields_arg, n_fields, &t);
    if (extra_owned) mxFree(extra);
    if (rc) mexErrMsgIdAndTxt("Spcies:$FORM$:HIP:solve", "%s", spcies_hip_last_error());
    for (long i = 0; i < B; i++) { mxGetPr(k_d)[i] = (double)k[i]; mxGetPr(e_d)[i] = (double)e[i]; }
    mxFree(k); mxFree(e);
    mxSetField(sol, 0, "update_time", mxCreateDoubleScalar(t.update_time));
    mxSetField(sol, 0, "solve_time", mxCreateDoubleScalar(t.solve_time));
    mxSetField(sol, 0, "polish_time", mxCreateDoubleScalar(t.polish_time));
    mxSetField(sol, 0, "run_time", mxCreateDoubleScalar(t.run_time));
    if (nlhs > 1) plhs[1] = k_d; else mxDestroyArray(k_d);
    if (nlhs > 2) plhs[2] = e_d; else mxDestroyArray(e_d);
    if (nlhs > 3) plhs[3] = sol; else mxDestroyArray(sol);
}
