/* mex.h -- TEST INFRASTRUCTURE: the MATLAB C-API entry points matlab/landing_solve_mex.c uses, implemented on plain
 * malloc'ed arrays so that the gateway can be compiled AND called by the CPU test-suite (no MATLAB in the image; the real
 * header ships with MATLAB).  Column-major data, dims[] as MATLAB reports them.  mexErrMsgTxt records the message and
 * longjmp()s back to the driver (MATLAB unwinds the mex call the same way). */
#ifndef LANDING_TEST_MEX_H
#define LANDING_TEST_MEX_H
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef size_t mwSize;
typedef enum { mxREAL = 0 } mxComplexity;
typedef enum { mxSTRUCT_CLASS = 2, mxLOGICAL_CLASS = 3, mxDOUBLE_CLASS = 6, mxSINGLE_CLASS = 7, mxINT32_CLASS = 12 } mxClassID;
#define MX_MAXFIELDS 40
typedef struct mxArray_tag {
  mwSize ndim; mwSize dims[4]; void* data; mxClassID cls;
  int nfields; const char* fname[MX_MAXFIELDS]; struct mxArray_tag* fval[MX_MAXFIELDS];     /* 1 x 1 struct arrays only */
} mxArray;
static size_t mx_elsize(mxClassID c) { return c == mxDOUBLE_CLASS ? 8 : (c == mxLOGICAL_CLASS ? 1 : 4); }
static mxArray* mx_new(mwSize ndim, const mwSize* dims, mxClassID c) {
  mxArray* a = (mxArray*)calloc(1, sizeof(mxArray)); size_t n = 1; mwSize i;
  a->ndim = ndim; a->cls = c;
  for (i = 0; i < ndim; ++i) { a->dims[i] = dims[i]; n *= dims[i]; }
  a->data = calloc(n ? n : 1, mx_elsize(c));
  return a;
}
static mxArray* mxCreateDoubleMatrix(mwSize m, mwSize n, mxComplexity f) { mwSize d[2] = {m, n}; (void)f; return mx_new(2, d, mxDOUBLE_CLASS); }
static mxArray* mxCreateNumericMatrix(mwSize m, mwSize n, mxClassID c, mxComplexity f) { mwSize d[2] = {m, n}; (void)f; return mx_new(2, d, c); }
static const mwSize* mxGetDimensions(const mxArray* a) { return a->dims; }
static mwSize mxGetNumberOfDimensions(const mxArray* a) { return a->ndim; }
static size_t mxGetNumberOfElements(const mxArray* a) { size_t n = 1; mwSize i; for (i = 0; i < a->ndim; ++i) n *= a->dims[i]; return n; }
static double* mxGetPr(const mxArray* a) { return (double*)a->data; }
static void* mxGetData(const mxArray* a) { return a->data; }
static int mxIsDouble(const mxArray* a) { return a->cls == mxDOUBLE_CLASS; }
static int mxIsLogical(const mxArray* a) { return a->cls == mxLOGICAL_CLASS; }
static int mxIsStruct(const mxArray* a) { return a->cls == mxSTRUCT_CLASS; }
static int mxIsComplex(const mxArray* a) { (void)a; return 0; }
static int mxIsSparse(const mxArray* a) { (void)a; return 0; }
static int mxIsEmpty(const mxArray* a) { return mxGetNumberOfElements(a) == 0; }
static double mxGetScalar(const mxArray* a) { return a->cls == mxDOUBLE_CLASS ? *(double*)a->data : (a->cls == mxLOGICAL_CLASS ? (double)*(unsigned char*)a->data : (double)*(int*)a->data); }
static mxArray* mxGetField(const mxArray* a, mwSize idx, const char* name) {
  int i; (void)idx;
  for (i = 0; i < a->nfields; ++i) if (strcmp(a->fname[i], name) == 0) return a->fval[i];
  return NULL;
}
static void* mxMalloc(size_t n) { return malloc(n ? n : 1); }
static void mxFree(void* p) { free(p); }
extern char g_mex_err[512];
extern jmp_buf g_mex_jmp;
static void mexErrMsgTxt(const char* m) { snprintf(g_mex_err, sizeof(g_mex_err), "%s", m ? m : ""); longjmp(g_mex_jmp, 1); }
static int mexAtExit(void (*f)(void)) { (void)f; return 0; }
#endif
