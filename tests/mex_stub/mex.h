/* TEST-ONLY stand-in for MATLAB's mex.h: just enough declarations to syntax-check OUR mex gateway
 * (matlab/formulations/+laxMPC/struct_laxMPC_ADMM_HIP_Matlab.c) with gcc -fsyntax-only.  MATLAB is
 * not available in this image; nothing is linked or run against this header. */
#ifndef MEX_STUB_H
#define MEX_STUB_H
#include <stddef.h>
typedef struct mxArray_tag mxArray;
typedef enum { mxREAL, mxCOMPLEX } mxComplexity;
void mexErrMsgIdAndTxt(const char *id, const char *fmt, ...);
int mexAtExit(void (*fn)(void));
int mxIsDouble(const mxArray *a);
size_t mxGetNumberOfElements(const mxArray *a);
double *mxGetPr(const mxArray *a);
void *mxMalloc(size_t n);
void mxFree(void *p);
mxArray *mxCreateDoubleMatrix(size_t m, size_t n, mxComplexity c);
mxArray *mxCreateDoubleScalar(double v);
mxArray *mxCreateStructMatrix(size_t m, size_t n, int nfields, const char **names);
void mxSetField(mxArray *s, size_t i, const char *name, mxArray *v);
void mxDestroyArray(mxArray *a);
#endif
