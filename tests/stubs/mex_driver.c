/* TEST INFRASTRUCTURE: calls matlab/landing_solve_mex.c's mexFunction on arrays handed over by ctypes (tests/test_args21_cpu.py) */
#include "mex.h"
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]);
/* data[i]: column-major buffer of argument i; ndim[i], dims[4*i..]: its MATLAB dimensions.  Outputs are copied out. */
int call_gateway(const double* const* data, const int* ndim, const int* dims, double* X, double* F, int* status, int* iters, double* kkt, int nx, int B) {
  mxArray* in[21]; mxArray* out[5] = {0, 0, 0, 0, 0}; int i, j;
  for (i = 0; i < 21; ++i) {
    mwSize d[4]; size_t n = 1;
    for (j = 0; j < ndim[i]; ++j) { d[j] = (mwSize)dims[4 * i + j]; n *= d[j]; }
    in[i] = mx_new((mwSize)ndim[i], d, mxDOUBLE_CLASS);
    memcpy(in[i]->data, data[i], n * sizeof(double));
  }
  mexFunction(5, out, 21, (const mxArray**)in);
  memcpy(X, out[0]->data, sizeof(double) * (size_t)nx * B); memcpy(F, out[1]->data, sizeof(double) * B);
  memcpy(status, out[2]->data, sizeof(int) * B); memcpy(iters, out[3]->data, sizeof(int) * B); memcpy(kkt, out[4]->data, sizeof(double) * 3 * B);
  return 0;
}
