/* landing_refine_mex.c -- MATLAB gateway of the batched KINODYNAMIC REFINEMENT solve (mex -> C ABI -> HIP).  Drop-in for the call
 *   [res.x, res.f] = f_knitro(Xref, Uref, dt, q_min, q_max, qd_min, qd_max, q_init, qd_init, c_init, q_term_min, q_term_max, qd_term_min,
 *                             qd_term_max, QN, x0, jpos_min, jpos_max, kin_box, mu, l_leg_max, mass, Ib, Ib_inv)
 * of the reference (generate_solver/generate_landingCtrller_KNITRO.m:373-377; generate_data/generate_training_data_automated.m:150-156,169-175),
 * for B drop states at once:
 *   [X, F, STATUS, ITERS, KKT, LAM_G] = landing_refine_mex(Xref, Uref, dt, ..., Ib_inv [, opts])
 * The same 24 arguments in the same order; B is the third dimension of Xref (12 x (N+1) x B); every other argument holds B members (trailing batch
 * dimension) or exactly ONE member, shared by the batch.  dt, mu, mass, Ib, Ib_inv must be the same for all members (they are in every caller).
 * STATUS: 0 KKT point (<= tol, default 1e-6), 1 iteration limit, 2 numerical failure, 3 infeasible (a row over the fixed initial stance is
 * violated: landing_nlp.h).  opts (optional struct): device (HIP device index, default 0) and any scalar field of landing_solver_opts by name.
 * Build:  mex landing_refine_mex.c -I<repo>/include -L<repo>/landing-controller_amd -llanding_mi355x */
#include <string.h>
#include "mex.h"
#include "landing_nlp.h"

static double opt_scalar(const mxArray* o, const char* name, double dflt) {
  const mxArray* f = o ? mxGetField(o, 0, name) : NULL;
  if (!f || mxIsEmpty(f)) return dflt;
  if (!mxIsDouble(f) && !mxIsLogical(f)) mexErrMsgTxt("landing_refine_mex: option fields must be double or logical scalars");
  return mxGetScalar(f);
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  static const char* names[24] = {"Xref", "Uref", "dt", "q_min", "q_max", "qd_min", "qd_max", "q_init", "qd_init", "c_init", "q_term_min", "q_term_max",
                                  "qd_term_min", "qd_term_max", "QN", "x0", "jpos_min", "jpos_max", "kin_box", "mu", "l_leg_max", "mass", "Ib", "Ib_inv"};
  const double* a[24]; double* tmp[24]; size_t per[24]; int i, b, N, B, device;
  long long nx, ng;
  char msg[256];
  landing_solver_opts o; landing_kinodyn_form form; int own_opts = 0;
  const mxArray* os = nrhs == 25 ? prhs[24] : NULL;
  if (nrhs != 24 && nrhs != 25) mexErrMsgTxt("landing_refine_mex: 24 inputs (generate_landingCtrller_KNITRO.m:373-377) and an optional options struct");
  if (nlhs > 6) mexErrMsgTxt("landing_refine_mex: at most 6 outputs [X, F, STATUS, ITERS, KKT, LAM_G]");
  for (i = 0; i < 24; ++i) if (!mxIsDouble(prhs[i]) || mxIsComplex(prhs[i]) || mxIsSparse(prhs[i])) {
    snprintf(msg, sizeof(msg), "landing_refine_mex: argument %d (%s) must be a full real double array", i + 1, names[i]); mexErrMsgTxt(msg); }
  {
    const mwSize* d = mxGetDimensions(prhs[0]); const mwSize nd = mxGetNumberOfDimensions(prhs[0]);
    if (nd < 2 || nd > 3 || d[0] != 12 || d[1] < 3) mexErrMsgTxt("landing_refine_mex: Xref must be 12 x (N+1) [x B]");
    N = (int)d[1] - 1; B = nd > 2 ? (int)d[2] : 1;
  }
  if (B < 1 || landing_kinodyn_nlp_dims(N, &nx, &ng)) mexErrMsgTxt("landing_refine_mex: empty batch or unsupported horizon (2 <= N <= 64 intervals)");
  for (i = 0; i < 24; ++i) per[i] = 6;
  per[0] = 12 * (size_t)(N + 1); per[1] = 24 * (size_t)N; per[2] = (size_t)N; per[9] = 12; per[14] = 12; per[15] = (size_t)nx; per[16] = per[17] = 12; per[18] = 2;
  per[19] = per[20] = per[21] = 1; per[22] = per[23] = 3;
  for (i = 0; i < 24; ++i) {
    const size_t n = mxGetNumberOfElements(prhs[i]);
    tmp[i] = NULL;
    if (n == per[i] * (size_t)B) a[i] = mxGetPr(prhs[i]);
    else if (n == per[i]) {
      tmp[i] = (double*)mxMalloc(per[i] * (size_t)B * sizeof(double));
      for (b = 0; b < B; ++b) memcpy(tmp[i] + (size_t)b * per[i], mxGetPr(prhs[i]), per[i] * sizeof(double));
      a[i] = tmp[i];
    } else {
      snprintf(msg, sizeof(msg), "landing_refine_mex: argument %d (%s) has %lu elements; expected %lu (one member) or %lu (B = %d members, N = %d)",
               i + 1, names[i], (unsigned long)n, (unsigned long)per[i], (unsigned long)(per[i] * (size_t)B), B, N);
      mexErrMsgTxt(msg);
    }
  }
  if (os && !mxIsEmpty(os) && !mxIsStruct(os)) mexErrMsgTxt("landing_refine_mex: the 25th argument must be an options struct");
  if (os && mxIsEmpty(os)) os = NULL;
  if (opt_scalar(os, "warm", 0.0) != 0.0) { landing_kinodyn_solver_opts_warm(&o); own_opts = 1; }      /* the `_ws` re-solve from a previous solution (landing_optimization.m:395-435) */
  else landing_kinodyn_solver_opts_default(&o);
#define OPT_D(f) o.f = opt_scalar(os, #f, o.f)
#define OPT_I(f) o.f = (int)opt_scalar(os, #f, (double)o.f)
  OPT_D(tol); OPT_I(max_iter); OPT_D(mu_init); OPT_D(bound_push); OPT_D(bound_frac); OPT_D(kappa_eps); OPT_D(kappa_mu); OPT_D(theta_mu); OPT_I(max_resets);
  OPT_D(reset_du); OPT_I(restart_period); OPT_D(delta_init); OPT_D(delta_inc_first); OPT_D(delta_inc); OPT_D(delta_dec); OPT_D(tau_min); OPT_D(alpha_fallback);
  OPT_D(reset_delta); OPT_I(clip_k); OPT_D(clip_until); OPT_D(theta_floor); OPT_I(fresh_restart); OPT_D(dual_step_cap); OPT_D(slack_corr); OPT_I(watchdog);
  OPT_D(barrier_smax); OPT_D(delta_floor); OPT_I(jam_clip); OPT_I(stag_relief); OPT_I(feas_phase); OPT_I(feas_stat); OPT_D(feas_polish); OPT_I(kd_clone_after); OPT_I(kd_clone_max); OPT_I(kd_clone_iter);
  device = (int)opt_scalar(os, "device", 0.0);
  {      /* the library's defaults (incl. its retry ladder for the members the first pass leaves undecided) unless the caller set a solver option */
    static const char* solver_fields[] = {"tol", "max_iter", "mu_init", "bound_push", "bound_frac", "kappa_eps", "kappa_mu", "theta_mu", "max_resets", "reset_du",
      "restart_period", "delta_init", "delta_inc_first", "delta_inc", "delta_dec", "tau_min", "alpha_fallback", "reset_delta", "clip_k", "clip_until", "theta_floor",
      "fresh_restart", "dual_step_cap", "slack_corr", "watchdog", "barrier_smax", "delta_floor", "jam_clip", "stag_relief", "feas_phase", "feas_stat", "feas_polish", "kd_clone_after", "kd_clone_max", "kd_clone_iter"};
    size_t q;
    for (q = 0; q < sizeof(solver_fields) / sizeof(solver_fields[0]); ++q) if (os && mxGetField(os, 0, solver_fields[q])) own_opts = 1;
  }
  landing_kinodyn_form_knitro(&form);      /* literals of generate_landingCtrller_KNITRO.m; options.kin_box_y0 (0.10: landing_optimization.m) / kin_box_x0 override them */
  form.kin_box_y0 = opt_scalar(os, "kin_box_y0", form.kin_box_y0); form.kin_box_x0 = opt_scalar(os, "kin_box_x0", form.kin_box_x0);
  {
    mxArray* f = nlhs > 1 ? mxCreateDoubleMatrix(1, B, mxREAL) : NULL;
    mxArray* st = nlhs > 2 ? mxCreateNumericMatrix(1, B, mxINT32_CLASS, mxREAL) : NULL;
    mxArray* it = nlhs > 3 ? mxCreateNumericMatrix(1, B, mxINT32_CLASS, mxREAL) : NULL;
    mxArray* kk = nlhs > 4 ? mxCreateDoubleMatrix(3, B, mxREAL) : NULL;
    mxArray* lg = nlhs > 5 ? mxCreateDoubleMatrix((mwSize)ng, B, mxREAL) : NULL;
    int rc;
    plhs[0] = mxCreateDoubleMatrix((mwSize)nx, B, mxREAL);
    rc = landing_solve_kinodyn_24_on_form(device, N, B, &form, a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], a[8], a[9], a[10], a[11], a[12], a[13], a[14], a[15], a[16], a[17],
                                     a[18], a[19], a[20], a[21], a[22], a[23], own_opts ? &o : NULL, mxGetPr(plhs[0]), f ? mxGetPr(f) : NULL, lg ? mxGetPr(lg) : NULL,
                                     st ? (int*)mxGetData(st) : NULL, it ? (int*)mxGetData(it) : NULL, kk ? mxGetPr(kk) : NULL);
    for (i = 0; i < 24; ++i) if (tmp[i]) mxFree(tmp[i]);
    if (rc) mexErrMsgTxt(landing_last_error());
    if (nlhs > 1) plhs[1] = f;
    if (nlhs > 2) plhs[2] = st;
    if (nlhs > 3) plhs[3] = it;
    if (nlhs > 4) plhs[4] = kk;
    if (nlhs > 5) plhs[5] = lg;
  }
}
