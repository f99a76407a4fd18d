/* landing_solve_mex.c -- MATLAB gateway of the batched landing solver (mex -> C ABI -> HIP).  Drop-in for the call
 *   [res.x, res.f] = f_ipopt_SRBM(Xref, Uref, dt, q_min, ..., Ib, Ib_inv)
 * of the reference (generate_landingCtrller_IPOPT.m:323-327; landing_optimization.m:305-311;
 * generate_training_data_automated.m:130-136), for B drop states at once:
 *   [X, F, STATUS, ITERS, KKT, LAM_G] = landing_solve_mex(Xref, Uref, dt, ..., Ib_inv [, opts])
 * The same 21 arguments in the same order.  The batch size B is the third dimension of Xref (12 x (N+1) x B); every other
 * argument holds either B members (trailing batch dimension: dt 1xNxB, 6-vectors 6xB, x0 nx x B, scalars 1xB, ...) or exactly
 * ONE member, which is then shared by the whole batch (what the reference's callers pass for mu, f_max, mass, q_min, ...).
 * Anything else -- a non-double array, a wrong element count -- is refused with an error message.
 * opts (optional struct): devices  vector of HIP device indices, the batch is sharded over them (default 0; an index may repeat)
 *                         warm     true = landing_solver_opts_warm (shifted / previous plan as x0) instead of the defaults
 *                         any scalar field of landing_solver_opts by name: tol, max_iter, mu_init, bound_push, bound_frac, ...
 * Outputs beyond the first are created only when asked for: F 1xB, STATUS / ITERS int32 1xB, KKT 3xB, LAM_G ng x B.
 * Build:  mex landing_solve_mex.c -I<repo>/include -L<repo>/landing-controller_amd -llanding_mi355x
 * All packing / sharding / solving lives in landing_solve_21_multi (include/landing_nlp.h).
 * Round 6: batches above 2048 members are streamed inside the library (landing_solve_batch_host -> landing_solve_stream_host: chunks of 1024 on two lanes of ONE context, uploads and
 * downloads under the solves; include/landing_nlp.h "streaming") -- the serial loop of generate_training_data_automated.m:38 becomes one call whatever the number of samples, with the
 * GPU kept busy across chunk boundaries (24.9 k instead of 19.3 k NLPs/s at N = 40) and a device workspace of 2 x 1024 members.  STATUS 4 = stalled (no certificate, no solution).
*/
#include <string.h>
#include "mex.h"
#include "landing_nlp.h"

static void bye(void) { landing_multi_release_cached(); }

static double opt_scalar(const mxArray* o, const char* name, double dflt) {
  const mxArray* f = o ? mxGetField(o, 0, name) : NULL;
  if (!f || mxIsEmpty(f)) return dflt;
  if (!mxIsDouble(f) && !mxIsLogical(f)) mexErrMsgTxt("landing_solve_mex: option fields must be double or logical scalars");
  return mxGetScalar(f);
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  static const char* names[21] = {"Xref", "Uref", "dt", "q_min", "q_max", "qd_min", "qd_max", "q_init", "qd_init", "q_term_min", "q_term_max",
                                  "qd_term_min", "qd_term_max", "QN", "x0", "mu", "l_leg_max", "f_max", "mass", "Ib", "Ib_inv"};
  const double* a[21]; double* tmp[21]; size_t per[21]; int i, b, N, B, ndev = 1, devs[64] = {0};
  char msg[256];
  static int registered = 0;
  if (nrhs != 21 && nrhs != 22) mexErrMsgTxt("landing_solve_mex: 21 inputs (generate_landingCtrller_IPOPT.m:323-327) and an optional options struct");
  if (nlhs > 6) mexErrMsgTxt("landing_solve_mex: at most 6 outputs [X, F, STATUS, ITERS, KKT, LAM_G]");
  for (i = 0; i < 21; ++i) if (!mxIsDouble(prhs[i]) || mxIsComplex(prhs[i]) || mxIsSparse(prhs[i])) {
    snprintf(msg, sizeof(msg), "landing_solve_mex: argument %d (%s) must be a full real double array", i + 1, names[i]); mexErrMsgTxt(msg); }
  {
    const mwSize* d = mxGetDimensions(prhs[0]); const mwSize nd = mxGetNumberOfDimensions(prhs[0]);
    if (nd < 2 || nd > 3 || d[0] != 12 || d[1] < 3) mexErrMsgTxt("landing_solve_mex: Xref must be 12 x (N+1) [x B]");
    N = (int)d[1] - 1; B = nd > 2 ? (int)d[2] : 1;
  }
  if (B < 1) mexErrMsgTxt("landing_solve_mex: empty batch");
  for (i = 0; i < 21; ++i) per[i] = 6;
  per[0] = 12 * (size_t)(N + 1); per[1] = 24 * (size_t)N; per[2] = (size_t)N; per[13] = 12; per[14] = (size_t)landing_nx(N);
  per[15] = per[16] = per[17] = per[18] = 1; per[19] = per[20] = 3;
  for (i = 0; i < 21; ++i) {
    const size_t n = mxGetNumberOfElements(prhs[i]);
    tmp[i] = NULL;
    if (n == per[i] * (size_t)B) a[i] = mxGetPr(prhs[i]);
    else if (n == per[i]) {        /* one member's worth: shared by the batch */
      tmp[i] = (double*)mxMalloc(per[i] * (size_t)B * sizeof(double));
      for (b = 0; b < B; ++b) memcpy(tmp[i] + (size_t)b * per[i], mxGetPr(prhs[i]), per[i] * sizeof(double));
      a[i] = tmp[i];
    } else {
      snprintf(msg, sizeof(msg), "landing_solve_mex: argument %d (%s) has %lu elements; expected %lu (one member) or %lu (B = %d members, N = %d)",
               i + 1, names[i], (unsigned long)n, (unsigned long)per[i], (unsigned long)(per[i] * (size_t)B), B, N);
      mexErrMsgTxt(msg);
    }
  }
  landing_solver_opts o;
  {
    const mxArray* os = nrhs == 22 ? prhs[21] : NULL;
    if (os && !mxIsEmpty(os) && !mxIsStruct(os)) mexErrMsgTxt("landing_solve_mex: the 22nd argument must be an options struct");
    if (os && mxIsEmpty(os)) os = NULL;
    if (opt_scalar(os, "warm", 0.0) != 0.0) landing_solver_opts_warm(&o); else landing_solver_opts_default(&o);
#define OPT_D(f) o.f = opt_scalar(os, #f, o.f)
#define OPT_I(f) o.f = (int)opt_scalar(os, #f, (double)o.f)
    OPT_D(tol); OPT_I(max_iter); OPT_D(mu_init); OPT_D(bound_push); OPT_D(bound_frac); OPT_D(kappa_eps); OPT_D(kappa_mu); OPT_D(theta_mu);
    OPT_I(max_resets); OPT_D(reset_du); OPT_I(restart_period); OPT_I(dispatch_order); OPT_D(delta_init); OPT_D(delta_inc_first); OPT_D(delta_inc);
    OPT_D(delta_dec); OPT_D(tau_min); OPT_D(alpha_fallback); OPT_D(reset_delta); OPT_I(clip_k); OPT_D(clip_until); OPT_D(theta_floor);
    OPT_I(fresh_restart); OPT_D(dual_step_cap); OPT_D(slack_corr); OPT_I(watchdog); OPT_D(barrier_smax); OPT_I(factor_fp32); OPT_I(feas_phase); OPT_D(feas_rho); OPT_D(feas_cert); OPT_D(delta_floor); OPT_I(jam_clip); OPT_I(stag_relief); OPT_I(feas_jam); OPT_I(feas_stat);
    OPT_D(feas_back); OPT_I(feas_max); OPT_D(feas_delta_dec); OPT_D(feas_ret_push); OPT_D(feas_ret_mu); OPT_I(feas_resume); OPT_D(feas_polish);
    {
      const mxArray* dv = os ? mxGetField(os, 0, "devices") : NULL;
      if (dv && !mxIsEmpty(dv)) {
        const size_t n = mxGetNumberOfElements(dv);
        if (!mxIsDouble(dv) || n > 64) mexErrMsgTxt("landing_solve_mex: opts.devices must be a double vector of at most 64 device indices");
        ndev = (int)n;
        for (i = 0; i < ndev; ++i) { const double v = mxGetPr(dv)[i]; if (v < 0 || v != (double)(int)v) mexErrMsgTxt("landing_solve_mex: opts.devices holds non-negative integers"); devs[i] = (int)v; }
      }
    }
  }
  if (!registered) { mexAtExit(bye); registered = 1; }
  {
    const mwSize nx = (mwSize)landing_nx(N), ng = (mwSize)landing_ng(N);
    mxArray* f = nlhs > 1 ? mxCreateDoubleMatrix(1, B, mxREAL) : NULL;
    mxArray* st = nlhs > 2 ? mxCreateNumericMatrix(1, B, mxINT32_CLASS, mxREAL) : NULL;
    mxArray* it = nlhs > 3 ? mxCreateNumericMatrix(1, B, mxINT32_CLASS, mxREAL) : NULL;
    mxArray* kk = nlhs > 4 ? mxCreateDoubleMatrix(3, B, mxREAL) : NULL;
    mxArray* lg = nlhs > 5 ? mxCreateDoubleMatrix(ng, B, mxREAL) : NULL;
    int rc;
    plhs[0] = mxCreateDoubleMatrix(nx, B, mxREAL);
    rc = landing_solve_21_multi(devs, ndev, N, B, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15],
                                a[16], a[17], a[18], a[19], a[20], &o, mxGetPr(plhs[0]), f ? mxGetPr(f) : NULL, lg ? mxGetPr(lg) : NULL,
                                st ? (int*)mxGetData(st) : NULL, it ? (int*)mxGetData(it) : NULL, kk ? mxGetPr(kk) : NULL);
    for (i = 0; i < 21; ++i) if (tmp[i]) mxFree(tmp[i]);
    if (rc) mexErrMsgTxt(landing_last_error());
    if (nlhs > 1) plhs[1] = f;
    if (nlhs > 2) plhs[2] = st;
    if (nlhs > 3) plhs[3] = it;
    if (nlhs > 4) plhs[4] = kk;
    if (nlhs > 5) plhs[5] = lg;
  }
}
