// mex gateway of the HIP platform for laxMPC-ADMM (and, with $FORM$ = equMPC, equMPC-ADMM).
//
// Counterpart of the reference's formulations/+laxMPC/struct_laxMPC_ADMM_C_Matlab.c: same call,
//     [u, k, e_flag, sol] = $INSERT_NAME$(x0, xr, ur)
// same argument checks and error ids (:34-55), same output record (:109-166).  Extension: x0 may be
// n x B (one instance per column, MATLAB column-major == the engine's [B][n] layout); xr / ur are
// n x 1 / m x 1 (shared) or n x B / m x B.  Then u is m x B, k and e_flag are 1 x B, sol.z/v/lambda dim x B.
//
// The problem blob is written next to the mex by cons_laxMPC_ADMM_HIP.m ($INSERT_NAME$.spcb) and is
// loaded on the first call; the engine handle lives until the mex is cleared.
// Build (emitted as exec_me by cons_laxMPC_ADMM_HIP.m):
//     mex -silent $INSERT_NAME$.c -I<repo>/include -L<repo>/spcies_amd -lspcies_hip
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "mex.h"
#include "spcies_hip.h"

$INSERT_DEFINES$ /* nn_, mm_, nm_, NN_, dim_ and BLOB_PATH, as cons_laxMPC_ADMM_HIP.m prints them */

static spcies_hip_handle g_handle = NULL;

static void at_exit(void) {
    if (g_handle) spcies_hip_destroy(g_handle);
    g_handle = NULL;
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
    int rc = spcies_hip_create(blob, (size_t)bytes, 0, &g_handle);
    mxFree(blob);
    if (rc) mexErrMsgIdAndTxt("Spcies:$FORM$:HIP:create", "%s", spcies_hip_last_error());
    mexAtExit(at_exit);
}

void mexFunction(int nlhs, mxArray *plhs[], int nrhs, const mxArray *prhs[]) {
    if (nrhs != 3) mexErrMsgIdAndTxt("Spcies:$FORM$:nrhs:number", "Three inputs are required");
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

    plhs[0] = mxCreateDoubleMatrix(mm_, B, mxREAL);
    mxArray *k_d = mxCreateDoubleMatrix(1, B, mxREAL), *e_d = mxCreateDoubleMatrix(1, B, mxREAL);
    int *k = (int *)mxMalloc(sizeof(int) * B), *e = (int *)mxMalloc(sizeof(int) * B);
    const char *field_names[] = {"z", "v", "lambda", "update_time", "solve_time", "polish_time", "run_time"};
    mxArray *sol = mxCreateStructMatrix(1, 1, 7, field_names);
    double *z = NULL, *v = NULL, *lam = NULL;
#ifdef DEBUG
    mxArray *z_pt = mxCreateDoubleMatrix(dim_, B, mxREAL), *v_pt = mxCreateDoubleMatrix(dim_, B, mxREAL),
            *l_pt = mxCreateDoubleMatrix(dim_, B, mxREAL);
    z = mxGetPr(z_pt); v = mxGetPr(v_pt); lam = mxGetPr(l_pt);
    mxSetField(sol, 0, "z", z_pt); mxSetField(sol, 0, "v", v_pt); mxSetField(sol, 0, "lambda", l_pt);
#endif
    spcies_hip_timing t;
    int rc = spcies_hip_solve_batch(g_handle, mxGetPr(prhs[0]), mxGetPr(prhs[1]), mxGetPr(prhs[2]), per_instance, B,
                                    mxGetPr(plhs[0]), k, e, z, v, lam, &t);
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
