/* mex.h -- TEST INFRASTRUCTURE: the dozen MATLAB C-API entry points matlab/landing_solve_mex.c uses, implemented on plain
 * malloc'ed arrays so that the gateway can be compiled AND called by the CPU test-suite (no MATLAB in the image; the real
 * header ships with MATLAB).  Column-major data, dims[] as MATLAB reports them. */
#ifndef LANDING_TEST_MEX_H
#define LANDING_TEST_MEX_H
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef size_t mwSize;
typedef enum { mxREAL = 0 } mxComplexity;
typedef enum { mxDOUBLE_CLASS = 6, mxINT32_CLASS = 12 } mxClassID;
typedef struct mxArray_tag { mwSize ndim; mwSize dims[4]; void* data; mxClassID cls; } mxArray;
static mxArray* mx_new(mwSize ndim, const mwSize* dims, mxClassID c) {
  mxArray* a = (mxArray*)calloc(1, sizeof(mxArray)); size_t n = 1; mwSize i;
  a->ndim = ndim; a->cls = c;
  for (i = 0; i < ndim; ++i) { a->dims[i] = dims[i]; n *= dims[i]; }
  a->data = calloc(n ? n : 1, c == mxDOUBLE_CLASS ? 8 : 4);
  return a;
}
static mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity f) { mwSize d[2] = {m, n}; (void)f; return mx_new(2, d, mxDOUBLE_CLASS); }
static mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID c, mxComplexity f) { mwSize d[2] = {m, n}; (void)f; return mx_new(2, d, c); }
static const mwSize* mxGetDimensions(const mxArray* a) { return a->dims; }
static mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->ndim; }
static double* mxGetPr(const mxArray* a) { return (double*)a->data; }
static void* mxGetData(const mxArray* a) { return a->data; }
static char g_mex_err[512];
static void mexErrMsgTxt(const char* m) { snprintf(g_mex_err, sizeof(g_mex_err), "%s", m ? m : ""); fprintf(stderr, "mexErrMsgTxt: %s\n", g_mex_err); abort(); }
static int mexAtExit(void (*f)(void)) { (void)f; return 0; }
#endif
