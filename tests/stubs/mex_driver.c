/* TEST INFRASTRUCTURE: calls matlab/landing_solve_mex.c's mexFunction on arrays handed over by ctypes (tests/test_args21_cpu.py,
 * tests/test_gpu_args21.py). */
#include "mex.h"
char g_mex_err[512];
jmp_buf g_mex_jmp;
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
const char* gateway_error(void) { return g_mex_err; }
/* data[i]: column-major buffer of argument i; ndim[i], dims[4*i..]: its MATLAB dimensions; cls[i]: 6 double, 7 single (to provoke
 * the class check).  Options: n_opt (name, value) scalar pairs + an optional device list; n_opt < 0 = no 22nd argument.
 * nlhs outputs are requested; those present are copied out (NULL pointers are skipped).  Returns 0, or 1 after mexErrMsgTxt. */
int call_gateway(const double* const* data, const int* ndim, const int* dims, const int* cls, int n_opt, const char* const* opt_name, const double* opt_val,
                 int n_dev, const double* devices, int nlhs, double* X, double* F, int* status, int* iters, double* kkt, double* lam, int nx, int ng, int B) {
  mxArray* in[22]; mxArray* out[6] = {0, 0, 0, 0, 0, 0}; int i, j, nrhs = 21;
  g_mex_err[0] = 0;
  for (i = 0; i < 21; ++i) {
    mwSize d[4]; size_t n = 1;
    for (j = 0; j < ndim[i]; ++j) { d[j] = (mwSize)dims[4 * i + j]; n *= d[j]; }
    in[i] = mx_new((mwSize)ndim[i], d, (mxClassID)cls[i]);
    if (cls[i] == mxDOUBLE_CLASS) memcpy(in[i]->data, data[i], n * sizeof(double));
  }
  if (n_opt >= 0) {
    mwSize one[2] = {1, 1};
    mxArray* s = mx_new(2, one, mxSTRUCT_CLASS);
    for (i = 0; i < n_opt; ++i) { s->fname[s->nfields] = opt_name[i]; s->fval[s->nfields] = mxCreateDoubleMatrix(1, 1, mxREAL); *mxGetPr(s->fval[s->nfields]) = opt_val[i]; s->nfields++; }
    if (n_dev > 0) { s->fname[s->nfields] = "devices"; s->fval[s->nfields] = mxCreateDoubleMatrix(1, (mwSize)n_dev, mxREAL); memcpy(mxGetPr(s->fval[s->nfields]), devices, sizeof(double) * n_dev); s->nfields++; }
    in[21] = s; nrhs = 22;
  }
  if (setjmp(g_mex_jmp)) return 1;
  mexFunction(nlhs, out, nrhs, (const mxArray**)in);
  if (X && out[0]) memcpy(X, out[0]->data, sizeof(double) * (size_t)nx * B);
  if (F && out[1]) memcpy(F, out[1]->data, sizeof(double) * B);
  if (status && out[2]) memcpy(status, out[2]->data, sizeof(int) * B);
  if (iters && out[3]) memcpy(iters, out[3]->data, sizeof(int) * B);
  if (kkt && out[4]) memcpy(kkt, out[4]->data, sizeof(double) * 3 * B);
  if (lam && out[5]) memcpy(lam, out[5]->data, sizeof(double) * (size_t)ng * B);
  for (i = 1; i < 6; ++i) if (i >= (nlhs > 1 ? nlhs : 1) && out[i]) return 2;      /* an output nobody asked for was created */
  return 0;
}
