/* landing_solve_mex.c -- MATLAB gateway of the batched landing solver (mex -> C ABI -> HIP).  Drop-in for the call
 *   [res.x, res.f] = f_ipopt_SRBM(Xref, Uref, dt, q_min, ..., Ib, Ib_inv)
 * of the reference (generate_landingCtrller_IPOPT.m:323-327; landing_optimization.m:305-311;
 * generate_training_data_automated.m:130-136): the same 21 arguments in the same order, each with an optional trailing
 * batch dimension B (Xref 12x(N+1)xB, dt 1xNxB, 6-vectors 6xB, x0 nxxB, scalars 1xB, ...).
 *   [X, F, STATUS, ITERS, KKT] = landing_solve_mex(Xref, Uref, dt, ..., Ib_inv)
 * Build:  mex landing_solve_mex.c -I<repo>/include -L<repo>/landing-controller_amd -llanding_mi355x
 * All packing / solving lives in landing_solve_21 (include/landing_nlp.h); this file only maps mxArrays to pointers. */
#include "mex.h"
#include "landing_nlp.h"
static landing_ctx* ctx = NULL; static int ctxN = 0;
static void bye(void) { if (ctx) landing_destroy(ctx); ctx = NULL; }
void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  const double* a[21]; int i;
  if (nrhs != 21) mexErrMsgTxt("landing_solve_mex: 21 inputs (generate_landingCtrller_IPOPT.m:323-327)");
  const mwSize* d = mxGetDimensions(prhs[0]);
  const int N = (int)d[1] - 1, B = mxGetNumberOfDimensions(prhs[0]) > 2 ? (int)d[2] : 1;
  for (i = 0; i < 21; ++i) a[i] = mxGetPr(prhs[i]);
  if (!ctx || ctxN != N) { bye(); ctx = landing_create(N, 0, NULL); ctxN = N; mexAtExit(bye); }
  if (!ctx) mexErrMsgTxt(landing_last_error());
  plhs[0] = mxCreateDoubleMatrix((mwSize)landing_nx(N), B, mxREAL); plhs[1] = mxCreateDoubleMatrix(1, B, mxREAL);
  mxArray* st = mxCreateNumericMatrix(1, B, mxINT32_CLASS, mxREAL); mxArray* it = mxCreateNumericMatrix(1, B, mxINT32_CLASS, mxREAL);
  mxArray* kk = mxCreateDoubleMatrix(3, B, mxREAL);
  if (landing_solve_21(ctx, B, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15],
                       a[16], a[17], a[18], a[19], a[20], NULL, mxGetPr(plhs[0]), mxGetPr(plhs[1]), (int*)mxGetData(st),
                       (int*)mxGetData(it), mxGetPr(kk))) mexErrMsgTxt(landing_last_error());
  if (nlhs > 2) plhs[2] = st;
  if (nlhs > 3) plhs[3] = it;
  if (nlhs > 4) plhs[4] = kk;
}
