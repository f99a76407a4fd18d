/*
 * landing_solver_cpu.c -- TEST INFRASTRUCTURE / CPU BASELINE, NOT PRODUCT CODE.
 *
 * Scalar fp64 CPU port of the interior-point algorithm the HIP solver implements (the reference's
 * solver layer -- CasADi Nlpsol('ipopt') + MA57, generate_landingCtrller_IPOPT.m:231-264,277,314 --
 * lives in third-party binaries that are absent from /root/reference: CasADi 3.5.5's libcasadi.so /
 * libcasadi_nlpsol_ipopt.so are listed in .MISSING_LARGE_BLOBS:9,14 and HSL MA57 is not redistributable;
 * so the solver layer of the oracle is a PORT of our algorithm, "parity unpinned" at the solution level
 * beyond the feasibility/objective goldens -- see DESIGN.md).  It follows IPOPT's published algorithm
 * (Waechter & Biegler 2006: slacks on every inequality row, fraction-to-the-boundary, filter line search,
 * monotone barrier update, inertia correction by delta_w) on the NLP functions of landing_oracle.c, and
 * solves the condensed KKT system with the same stage-wise Riccati recursion as the GPU kernel.
 *
 * Used by bench.py's cpu_baseline leg (timed on the host cores) and by tests as a cross-check of the GPU
 * solver's iteration counts.  OpenMP over batch members.
 */
#include "landing_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#ifdef _OPENMP
#include <omp.h>
#endif

typedef struct {
  double tol; int max_iter; double mu_init, bound_push, bound_frac, kappa_eps, kappa_mu, theta_mu;
  int max_resets; double reset_du;
  double delta_init, delta_inc_first, delta_inc, delta_dec, tau_min, alpha_fallback;
  int restart_period;
  double reset_delta;
  double barrier_smax;   /* s_max of the scaled optimality error in the barrier-subproblem test (include/landing_nlp.h); 0 = unscaled */
  int watchdog;          /* forced step to the boundary after this many successive iterations with step lengths <= 1/16 of it; 0 = off */
  double slack_corr;     /* slack correction at a rejected first trial point: slacks moved to g(x_trial), at most to (1 - slack_corr) of the way to a bound */
  double dual_step_cap;  /* a_du <= dual_step_cap * alpha (include/landing_nlp.h); 0 = independent dual step length                      */
  int fresh_restart;     /* bit mask of the restart rules of include/landing_nlp.h (default 9 = 1 | 8)                                   */
  double theta_floor;    /* constraint violations (1-norm theta) below theta_floor * tol count as equal in the filter tests              */
  int clip_k;            /* the step to the boundary is set by the clip_k-th most blocking slack; the more blocking ones stop at    */
  double clip_until;     /* (1 - tau) of their distance (include/landing_nlp.h); only while pr > clip_until                      */
  int feas_phase;        /* feasibility (restoration) phase, include/landing_nlp.h: a solve that would end as NUMERICAL / MAX_ITER continues on
                            the ELASTIC problem -- every inequality row may be violated by n, p >= 0 at the price feas_rho (n + p), no objective --
                            with the same primal-dual machinery; a feasible point restarts the solve from there, a KKT point of the elastic
                            problem with positive violation ends it as LANDING_INFEASIBLE (3): a certificate of local infeasibility           */
  double feas_rho;       /* price of a unit of violation (IPOPT's restoration phase: 1000)                                                   */
  double feas_cert;      /* l1 violation above which the elastic KKT point counts as a certificate                                           */
  double delta_floor;    /* first regularisation tried in an iteration of the terminal-cost form (include/landing_nlp.h)                     */
  int jam_clip, stag_relief;      /* include/landing_nlp.h */
  int feas_jam, feas_stat;        /* include/landing_nlp.h (round 5) */
  double feas_back; int feas_max; double feas_delta_dec, feas_ret_push, feas_ret_mu; int feas_resume; double feas_polish;      /* include/landing_nlp.h (round 6) */
  int max_soc;           /* LAB ONLY (the kernel has no counterpart; default 0 = the kernel's algorithm): second-order corrections in IPOPT's form (Waechter & Biegler 2006, A-5.5..5.9) */
} lo_solver_opts;

void lo_solver_opts_default(lo_solver_opts* o) {
  o->tol = 1e-6; o->max_iter = 3000; o->mu_init = 0.0 /* auto: 0.5 terminal-cost form, 0.1 with a running cost */; o->bound_push = 0.0 /* auto: 1.0 / 0.5 */; o->bound_frac = 0.1;
  o->kappa_eps = 0.0 /* auto: 120 terminal-cost form, 10 with a running cost */; o->kappa_mu = 0.2; o->theta_mu = 0.0 /* auto: 1.8 / 1.5 */; o->max_resets = 8; o->reset_du = 1e9;
  o->delta_init = 1e-4; o->delta_inc_first = 10.0; o->delta_inc = 4.0; o->delta_dec = 0.5; o->tau_min = 0.9; o->alpha_fallback = 1e-2; o->restart_period = 75; o->reset_delta = 1e5;
  o->barrier_smax = 1.0; o->watchdog = 3; o->slack_corr = 0.9; o->dual_step_cap = 1.0; o->fresh_restart = 9; o->theta_floor = 30.0; o->clip_k = 4; o->clip_until = 0.03;
  o->feas_phase = 1; o->feas_rho = 1000.0; o->feas_cert = 1e-4;
  o->delta_floor = 3e-4; o->jam_clip = 2; o->stag_relief = 3; o->feas_jam = 8; o->feas_stat = 25;
  o->feas_back = 0.2; o->feas_max = 3; o->feas_delta_dec = 0.1; o->feas_ret_push = 0.01; o->feas_ret_mu = 0.01; o->feas_resume = 1; o->feas_polish = 1e-8;
  o->max_soc = 0;
}

long long lo_soc_taken = 0, lo_soc_tried = 0;      /* LAB ONLY: corrected steps taken / correction solves, over the life of the process */
#define NW 48
static const int ROW2STATE[12] = {0, 1, 2, 3, 4, 5, 9, 10, 11, 6, 7, 8};
/* local variable (lo_stage_eval order X_k,c_k,f_k,X+,c+) -> w index (X,c,f,c+) or -1 */
static int loc2w(int loc) { if (loc < 36) return loc; if (loc < 48) return -1; return 36 + (loc - 48); }

typedef struct {
  int N; lo_int nx, ng; int feas;
  double *en, *ep, *wn, *wp, *den, *dep, *dwn, *dwp;      /* elastic variables of the feasibility phase and their steps */
  double *x, *xt, *dx, *g, *gt, *s, *ds, *zL, *zU, *dzL, *dzU, *y, *yn, *lb, *ub, *sig, *rho;
  double *Jst;   /* N x 104 x 60 */
  double *Hst;   /* N x 60 x 60  */
  double *M, *mvec, *Ah, *bv;         /* per stage: 48x48, 48, 12x36, 12 */
  double *K, *kap, *Px, *pvx;         /* per stage: 24x24, 24, 12x24, 12 (index N: terminal) */
} work_t;

static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }

static void eval_g(const lo_form* F, const double* x, const double* p, double* g) { lo_nlp_g(F, x, p, g); }

/* in-place LDL^T elimination of the n x n block with right-hand sides: solves A X = B (A spd), returns 0 if a
 * pivot is not positive.  A: n x n (ld lda), B: n x m (ld ldb) overwritten by X. */
static int spd_solve(double* A, int lda, int n, double* B, int ldb, int m) {
  int i, j, c;
  for (j = 0; j < n; ++j) {
    const double d = A[j * lda + j];
    if (!(d > 0.0) || !(d < 1e300)) return 0;
    for (i = j + 1; i < n; ++i) {
      const double l = A[i * lda + j] / d;
      if (l == 0.0) continue;
      for (c = j + 1; c < n; ++c) A[i * lda + c] -= l * A[j * lda + c];
      for (c = 0; c < m; ++c) B[i * ldb + c] -= l * B[j * ldb + c];
      A[i * lda + j] = l;
    }
  }
  for (j = n - 1; j >= 0; --j) {       /* back substitution with unit upper L^T and D */
    for (c = 0; c < m; ++c) {
      double v = B[j * ldb + c] / A[j * lda + j];
      for (i = j + 1; i < n; ++i) v -= A[i * lda + j] * B[i * ldb + c];
      B[j * ldb + c] = v;
    }
  }
  return 1;
}

/* backward Riccati sweep; returns 1 on success */
static int riccati_backward(const lo_form* F, const double* p, work_t* W, double delta, const lo_poff* o, double* sig0) {
  const int N = W->N;
  double P[24 * 24], pv[24], G[NW * NW], gam[NW], Y[24 * 36], q[24], Guu[24 * 24], R[24 * 25];
  int k, i, j, t;
  memset(P, 0, sizeof(P)); memset(pv, 0, sizeof(pv));
  for (i = 0; i < 12; ++i) {
    const int ra = i < 6 ? 12 + i : 24 + (i - 6), rb = i < 6 ? 18 + i : 30 + (i - 6);
    const double qn2 = W->feas ? 0.0 : 2.0 * p[o->QN + i];      /* the feasibility phase has no objective */
    P[i * 24 + i] = qn2 + W->sig[ra] + W->sig[rb] + delta;
    pv[i] = qn2 * (W->x[12 * N + i] - p[12 * N + i]) + W->rho[ra] + W->rho[rb];
  }
  for (i = 0; i < 12; ++i) { for (j = 0; j < 24; ++j) W->Px[(size_t)N * 288 + i * 24 + j] = P[i * 24 + j]; W->pvx[N * 12 + i] = pv[i]; }
  for (k = N - 1; k >= 0; --k) {
    const int last = (k == N - 1), nu = last ? 12 : 24, nsn = last ? 12 : 24, nw = 24 + nu;
    const double* Mk = W->M + (size_t)k * NW * NW; const double* Ah = W->Ah + (size_t)k * 432; const double* bv = W->bv + k * 12;
    memcpy(G, Mk, sizeof(G)); memcpy(gam, W->mvec + k * NW, sizeof(gam));
    for (i = 0; i < nw; ++i) G[i * NW + i] += delta;
    for (i = 0; i < nsn; ++i) {
      for (j = 0; j < 36; ++j) { double a = 0; for (t = 0; t < 12; ++t) a += P[i * 24 + t] * Ah[t * 36 + j]; Y[i * 36 + j] = a; }
      { double a = pv[i]; for (t = 0; t < 12; ++t) a += P[i * 24 + t] * bv[t]; q[i] = a; }
    }
    for (i = 0; i < 36; ++i) {
      for (j = 0; j < 36; ++j) { double a = 0; for (t = 0; t < 12; ++t) a += Ah[t * 36 + i] * Y[t * 36 + j]; G[i * NW + j] += a; }
      { double a = 0; for (t = 0; t < 12; ++t) a += Ah[t * 36 + i] * q[t]; gam[i] += a; }
    }
    if (!last) {
      for (i = 0; i < 12; ++i) {
        for (j = 0; j < 36; ++j) { G[(36 + i) * NW + j] += Y[(12 + i) * 36 + j]; G[j * NW + 36 + i] += Y[(12 + i) * 36 + j]; }
        for (j = 0; j < 12; ++j) G[(36 + i) * NW + 36 + j] += P[(12 + i) * 24 + 12 + j];
        gam[36 + i] += q[12 + i];
      }
    }
    /* K = Guu^-1 [Gus | gam_u] */
    for (i = 0; i < nu; ++i) {
      for (j = 0; j < nu; ++j) Guu[i * 24 + j] = G[(24 + i) * NW + 24 + j];
      for (j = 0; j < 24; ++j) R[i * 25 + j] = G[(24 + i) * NW + j];
      R[i * 25 + 24] = gam[24 + i];
    }
    if (!spd_solve(Guu, 24, nu, R, 25, 25)) return 0;
    for (i = 0; i < nu; ++i) { for (j = 0; j < 24; ++j) W->K[(size_t)k * 576 + i * 24 + j] = R[i * 25 + j]; W->kap[k * 24 + i] = R[i * 25 + 24]; }
    for (i = 0; i < 24; ++i) {
      for (j = 0; j < 24; ++j) { double a = G[i * NW + j]; for (t = 0; t < nu; ++t) a -= G[(24 + t) * NW + i] * R[t * 25 + j]; P[i * 24 + j] = a; }
      { double a = gam[i]; for (t = 0; t < nu; ++t) a -= G[(24 + t) * NW + i] * R[t * 25 + 24]; pv[i] = a; }
    }
    for (i = 0; i < 12; ++i) { for (j = 0; j < 24; ++j) W->Px[(size_t)k * 288 + i * 24 + j] = P[i * 24 + j]; W->pvx[k * 12 + i] = pv[i]; }
  }
  {  /* stage 0: X_0 fixed, c_0 free */
    double Pcc[144], rhs[12];
    for (i = 0; i < 12; ++i) sig0[i] = (i < 6 ? p[o->q_init + i] : p[o->qd_init + i - 6]) - W->x[i];
    for (i = 0; i < 12; ++i) {
      double a = pv[12 + i];
      for (j = 0; j < 12; ++j) { Pcc[i * 12 + j] = P[(12 + i) * 24 + 12 + j]; a += P[(12 + i) * 24 + j] * sig0[j]; }
      rhs[i] = a;
    }
    if (!spd_solve(Pcc, 12, 12, rhs, 1, 1)) return 0;
    for (i = 0; i < 12; ++i) sig0[12 + i] = -rhs[i];
  }
  return 1;
}

/* LAB ONLY (max_soc): the Newton step of the current iteration's KKT matrix (same Sigma, same delta: the sweep re-factorises the identical matrix, cost is of no
 * interest here) for ANOTHER primal residual c (rows 12..ng: g - s of an inequality row, g - lb of an equality row): dx, ds and the multipliers of the equality
 * rows.  Interior-point mode only.  Overwrites W->rho, W->mvec, W->bv, W->kap, W->pvx (none is read again in the iteration). */
static int soc_solve(const lo_form* F, const double* p, work_t* W, const lo_poff* o, double mu, double delta, const double* c, double* dx, double* ds, double* yn) {
  const int N = W->N; const lo_int ng = W->ng; lo_int r; int k, i; double sig[24], w[NW];
  for (r = 0; r < ng; ++r) {
    const double lb = W->lb[r], ub = W->ub[r]; double rh = 0;
    if (r >= 12 && lb != ub) {
      const double s = W->s[r];
      if (lb > -INFINITY) rh -= mu / (s - lb);
      if (ub < INFINITY) rh += mu / (ub - s);
      rh += W->sig[r] * c[r];
    }
    W->rho[r] = rh;
  }
  for (k = 0; k < N; ++k) {
    const int nr = lo_stage_rows(F, k), g0 = 36 + 104 * k; int q, a;
    const double* J = W->Jst + (size_t)k * 104 * 60; double* mk = W->mvec + k * NW;
    memset(mk, 0, sizeof(double) * NW);
    for (q = 12; q < nr; ++q) { const double rh = W->rho[g0 + q]; for (a = 0; a < 60; ++a) if (J[q * 60 + a] != 0.0 && loc2w(a) >= 0) mk[loc2w(a)] += rh * J[q * 60 + a]; }
    for (q = 0; q < 12; ++q) W->bv[k * 12 + ROW2STATE[q]] = -c[g0 + q];
    if (F->run_cost) { double gr[36]; for (a = 0; a < 36; ++a) gr[a] = 0.0; (void)lo_run_cost_stage(F, W->x, p, k, gr, gr + 12, gr + 24); for (a = 0; a < 36; ++a) mk[a] += gr[a]; }
  }
  if (!riccati_backward(F, p, W, delta, o, sig)) return 0;
  for (k = 0; k < N; ++k) {
    const int last = (k == N - 1), nu = last ? 12 : 24, nr = lo_stage_rows(F, k), g0 = 36 + 104 * k; int q, a, t;
    const double* J = W->Jst + (size_t)k * 104 * 60; double signext[24];
    for (a = 0; a < 24; ++a) w[a] = sig[a];
    for (a = 0; a < nu; ++a) { double v = W->kap[k * 24 + a]; for (t = 0; t < 24; ++t) v += W->K[(size_t)k * 576 + a * 24 + t] * sig[t]; w[24 + a] = -v; }
    for (a = nu; a < 24; ++a) w[24 + a] = 0.0;
    for (a = 0; a < 12; ++a) { dx[12 * k + a] = w[a]; dx[12 * (N + 1) + 24 * k + a] = w[12 + a]; dx[12 * (N + 1) + 24 * k + 12 + a] = w[24 + a]; }
    for (q = 12; q < nr; ++q) {
      double v = 0; for (a = 0; a < 60; ++a) { const int wa = loc2w(a); if (wa >= 0 && J[q * 60 + a] != 0.0) v += J[q * 60 + a] * w[wa]; }
      ds[g0 + q] = v + c[g0 + q];
    }
    for (a = 0; a < 12; ++a) { double v = W->bv[k * 12 + a]; for (t = 0; t < 36; ++t) v += W->Ah[(size_t)k * 432 + a * 36 + t] * w[t]; signext[a] = v; }
    for (a = 0; a < 12; ++a) signext[12 + a] = last ? 0.0 : w[36 + a];
    for (a = 0; a < 12; ++a) {
      double v = W->pvx[(k + 1) * 12 + a]; const int nn = last ? 12 : 24;
      for (t = 0; t < nn; ++t) v += W->Px[(size_t)(k + 1) * 288 + a * 24 + t] * signext[t];
      yn[g0 + (a < 6 ? a : (a < 9 ? a + 3 : a - 3))] = -v;
    }
    memcpy(sig, signext, sizeof(sig));
  }
  for (i = 0; i < 12; ++i) {
    const int ra = i < 6 ? 12 + i : 24 + (i - 6), rb = i < 6 ? 18 + i : 30 + (i - 6);
    dx[12 * N + i] = sig[i]; ds[ra] = sig[i] + c[ra]; ds[rb] = sig[i] + c[rb];
  }
  return 1;
}

static void init_slacks(work_t* W, const lo_solver_opts* op) {
  lo_int r;
  for (r = 0; r < W->ng; ++r) {
    const double lb = W->lb[r], ub = W->ub[r];
    double sv = 0, zl = 0, zu = 0;
    if (r >= 12 && lb != ub) {
      const int hL = lb > -INFINITY, hU = ub < INFINITY; double pl, pu;
      sv = W->g[r];
      if (hL && hU) { pl = fmin(op->bound_push * fmax(1.0, fabs(lb)), op->bound_frac * (ub - lb)); pu = fmin(op->bound_push * fmax(1.0, fabs(ub)), op->bound_frac * (ub - lb)); }
      else { pl = op->bound_push * fmax(1.0, hL ? fabs(lb) : 0.0); pu = op->bound_push * fmax(1.0, hU ? fabs(ub) : 0.0); }
      if (hL) sv = fmax(sv, lb + pl);
      if (hU) sv = fmin(sv, ub - pu);
      zl = hL ? 1.0 : 0.0; zu = hU ? 1.0 : 0.0;
    }
    W->s[r] = sv; W->zL[r] = zl; W->zU[r] = zu; W->y[r] = zu - zl;
  }
}

/* sorted insert into the four largest values seen so far (t[0] >= t[1] >= t[2] >= t[3]) */
static void top4_push(double t[4], double v) {
  int i;
  for (i = 0; i < 4; ++i) if (v > t[i]) { const double h = t[i]; t[i] = v; v = h; }
}
/* slack at step length alpha; with `clip` a slack does not pass (1 - tau) of its current distance to either bound (the
 * componentwise fraction-to-the-boundary rule, applied to the few slacks that are more blocking than the one that set alpha) */
static double slack_reset(double s, double g, double lb, double ub, double k) {
  const double lo = lb > -INFINITY ? lb + k * (s - lb) : -INFINITY, hi = ub < INFINITY ? ub - k * (ub - s) : INFINITY;
  return fmin(fmax(g, lo), hi);
}
static double slack_step(double s0, double ds, double alpha, double lb, double ub, int clip, double tau) {
  double s = s0 + alpha * ds;
  if (clip) {
    if (lb > -INFINITY) s = fmax(s, lb + (1.0 - tau) * (s0 - lb));
    if (ub < INFINITY) s = fmin(s, ub - (1.0 - tau) * (ub - s0));
  }
  return s;
}

/* slacks and multipliers of the interior-point iteration at the point a feasibility phase hands back (include/landing_nlp.h, feas_ret_push):
 * slacks pushed only feas_ret_push off their bounds, bound multipliers mu / distance -- what the phase gained in feasibility is kept */
static void init_slacks_return(work_t* W, const lo_solver_opts* op, double mu) {
  lo_solver_opts o2 = *op; lo_int r;
  o2.bound_push = op->feas_ret_push; o2.bound_frac = op->feas_ret_push;
  init_slacks(W, &o2);
  for (r = 12; r < W->ng; ++r) {
    const double lb = W->lb[r], ub = W->ub[r];
    if (lb == ub) continue;
    W->zL[r] = lb > -INFINITY ? fmin(fmax(mu / (W->s[r] - lb), 1e-8), 1e3) : 0.0;
    W->zU[r] = ub < INFINITY ? fmin(fmax(mu / (ub - W->s[r]), 1e-8), 1e3) : 0.0;
    W->y[r] = W->zU[r] - W->zL[r];
  }
}

/* one NLP; returns status (0 converged, 1 max_iter, 2 numerical, 3 certified locally infeasible, 4 stalled: include/landing_nlp.h) */
static int solve_one(const lo_form* F, const double* p, const double* x0, const lo_solver_opts* op, double* x_out,
                     double* lam_out, int* iters_out, double kkt_out[3], long long counters[2]) {
  const int N = F->N; const lo_int nx = lo_nx(N), ng = lo_ng(N);
  lo_poff o; work_t Wk, *W = &Wk; lo_int i, r; int k, it, status = 1, nfilt = 0, streak = 0, nreset = 0, last_reset_it = 0, ncrawl = 0, last_mu_it = 0, cutstreak = 0, force_step = 0, wd_count = 0;
  double mu = op->mu_init, delta_last = 0.0, th_max = 0.0, e_du = 0.0; int clip_k_cur = op->clip_k;
  double filt_th[64], filt_ph[64];
  double* gx;
  int feas = 0, feas_used = 0, fact_failed = 0, lim = op->max_iter; const double frho = op->feas_rho;
  /* round 6 (include/landing_nlp.h): up to feas_max entries into the phase, an entry that is not the last one returns as soon as the violation has
   * fallen to feas_back of its value at the entry; a phase that stalls hands its point back once (feas_resume); everything inside 3 max_iter iterations */
  int n_feas = 0, stalled = 0, polished = 0; double th_entry = 0.0; const int hard_lim = op->max_iter > 0 ? 3 * op->max_iter : 0;
  const double fdec = op->feas_delta_dec > 0.0 ? op->feas_delta_dec : op->delta_dec; const int adapt = op->feas_delta_dec > 0.0;
  double fdc = fdec;      /* ... adapted inside a phase: squared after an iteration whose first factorisation succeeded, square root (<= 0.7) after one that needed more */
  lo_param_offsets_form(F, &o);
  W->N = N; W->nx = nx; W->ng = ng; W->feas = 0;
  W->en = dalloc(ng); W->ep = dalloc(ng); W->wn = dalloc(ng); W->wp = dalloc(ng); W->den = dalloc(ng); W->dep = dalloc(ng); W->dwn = dalloc(ng); W->dwp = dalloc(ng);
  W->x = dalloc(nx); W->xt = dalloc(nx); W->dx = dalloc(nx); gx = dalloc(nx);
  W->g = dalloc(ng); W->gt = dalloc(ng); W->s = dalloc(ng); W->ds = dalloc(ng); W->zL = dalloc(ng); W->zU = dalloc(ng);
  W->dzL = dalloc(ng); W->dzU = dalloc(ng); W->y = dalloc(ng); W->yn = dalloc(ng); W->lb = dalloc(ng); W->ub = dalloc(ng);
  W->sig = dalloc(ng); W->rho = dalloc(ng);
  W->Jst = dalloc((size_t)N * 104 * 60); W->Hst = dalloc((size_t)N * 3600);
  W->M = dalloc((size_t)N * NW * NW); W->mvec = dalloc((size_t)N * NW); W->Ah = dalloc((size_t)N * 432); W->bv = dalloc((size_t)N * 12);
  W->K = dalloc((size_t)N * 576); W->kap = dalloc((size_t)N * 24); W->Px = dalloc((size_t)(N + 1) * 288); W->pvx = dalloc((size_t)(N + 1) * 12);
  memcpy(W->x, x0, sizeof(double) * nx);
  for (i = 0; i < 6; ++i) { W->x[i] = p[o.q_init + i]; W->x[6 + i] = p[o.qd_init + i]; }
  lo_bounds(F, p, W->lb, W->ub);
  eval_g(F, W->x, p, W->g);
  init_slacks(W, op);
  int stag = 0, full_prev = 0; double e_prev = 1e300; const int stag_k = op->stag_relief;      /* jam_clip / stag_relief: include/landing_nlp.h */
  int jamrun = 0; const int jam_k = op->jam_clip; const double jam_a = 0.02;
  int fjam = 0, fstat = 0; double v1_ref = 0.0;      /* feas_jam / feas_stat: include/landing_nlp.h */
  for (it = 0; it <= lim; ++it) {
    double du = 0, pr = 0, co = 0, tau, delta;
    int fact_ok = 0, attempt, clip_now; double use_reset = 0.0;
    double top[4]; const double th_floor = op->theta_floor * op->tol;
    double sig[24], w[NW], a_pr = 1.0, a_du = 1.0, th0 = 0, bar = 0, dphi = 0, f0 = 0, ph0, alpha;
    int accepted = 0, armijo = 0;
    /* derivatives per stage + gx = grad f + J^T y */
    memset(gx, 0, sizeof(double) * nx);
    for (i = 0; i < 12; ++i) {
      gx[12 * N + i] = (feas ? 0.0 : 2.0 * p[o.QN + i] * (W->x[12 * N + i] - p[12 * N + i])) + (i < 6 ? W->y[12 + i] + W->y[18 + i] : W->y[24 + i - 6] + W->y[30 + i - 6]);
    }
    for (k = 0; k < N; ++k) {
      const int nr = lo_stage_rows(F, k); int q, c;
      double lam[LO_NROW]; double* J = W->Jst + (size_t)k * 104 * 60;
      for (q = 0; q < LO_NROW; ++q) lam[q] = q < nr ? W->y[36 + 104 * k + q] : 0.0;
      lo_stage_eval(F, k, W->x, p, lam, NULL, J, W->Hst + (size_t)k * 3600);
      if (F->run_cost) {   /* running cost of the stage: gradient into gx, constant Hessian entries into the dense stage block */
        double* Hs = W->Hst + (size_t)k * 3600; const double dtk = p[o.dt + k]; int a, l2;
        double* gU = gx + 12 * (N + 1) + 24 * k;
        (void)lo_run_cost_stage(F, W->x, p, k, gx + 12 * k, gU, gU + 12);
        for (c = 0; c < 12; ++c) Hs[c * 60 + c] += 2.0 * dtk * lo_rc_weight(F, p, 0, c);
        for (l2 = 0; l2 < 4; ++l2) for (a = 0; a < 3; ++a) {
          const int ic = 12 + 3 * l2 + a, jf = 24 + 3 * l2 + a; const double hc = 2.0 * dtk * lo_rc_weight(F, p, 1, a);
          Hs[a * 60 + a] += hc; Hs[ic * 60 + ic] += hc; Hs[a * 60 + ic] -= hc; Hs[ic * 60 + a] -= hc;
          Hs[jf * 60 + jf] += 2.0 * dtk * lo_rc_weight(F, p, 2, a);
        }
      }
      for (q = 0; q < nr; ++q) for (c = 0; c < 60; ++c) if (J[q * 60 + c] != 0.0) {
        const lo_int gi = c < 12 ? 12 * k + c : (c < 36 ? 12 * (N + 1) + 24 * k + (c - 12) : (c < 48 ? 12 * (k + 1) + (c - 36) : 12 * (N + 1) + 24 * (k + 1) + (c - 48)));
        gx[gi] += lam[q] * J[q * 60 + c];
      }
    }
    for (i = 12; i < nx; ++i) du = fmax(du, fabs(gx[i]));
    for (r = 12; r < ng; ++r) {
      const double lb = W->lb[r], ub = W->ub[r], g = W->g[r];
      if (lb == ub) { pr = fmax(pr, fabs(g - lb)); continue; }
      pr = fmax(pr, fabs(g - W->s[r]));
      if (feas) {
        if (lb > -INFINITY) { co = fmax(co, (W->s[r] - lb + W->en[r]) * W->zL[r]); co = fmax(co, W->en[r] * W->wn[r]); du = fmax(du, fabs(W->zL[r] + W->wn[r] - frho)); }
        if (ub < INFINITY) { co = fmax(co, (ub + W->ep[r] - W->s[r]) * W->zU[r]); co = fmax(co, W->ep[r] * W->wp[r]); du = fmax(du, fabs(W->zU[r] + W->wp[r] - frho)); }
        continue;
      }
      if (lb > -INFINITY) co = fmax(co, (W->s[r] - lb) * W->zL[r]);
      if (ub < INFINITY) co = fmax(co, (ub - W->s[r]) * W->zU[r]);
    }
    e_du = du;
    if (stag_k > 0) {
      const double E = fmax(pr, du);
      if (!feas && mu <= op->tol / 10.0 * 1.0000001 && full_prev && E > 0.5 * e_prev) stag++; else stag = 0;
      e_prev = E;
    }
    if (getenv("LO_TRACE")) fprintf(stderr, "it %4d pr %9.2e du %9.2e co %9.2e mu %8.1e dlast %8.1e nreset %d nfilt %d\n", it, pr, du, co, mu, delta_last, nreset, nfilt);
    if (feas) {
      double vmax = 0.0, v1 = 0.0, theq = 0.0; int back = 0;
      for (r = 12; r < ng; ++r) {
        const double lb = W->lb[r], ub = W->ub[r], g = W->g[r], v = fmax(fmax(lb - g, g - ub), 0.0);
        if (lb == ub) { theq += fabs(g - lb); continue; }
        vmax = fmax(vmax, v); v1 += v;
      }
      if (getenv("LO_TRACE")) fprintf(stderr, "   feas: viol_inf %9.2e viol_1 %9.2e\n", vmax, v1);
      if (!(du < 1e300) || !(pr < 1e300) || !(co < 1e300)) { status = 2; break; }
      if (vmax <= 1e-9 && pr <= op->tol) back = 1;                       /* a feasible point: back to the interior-point solve from here */
      else if (fmax(du, fmax(pr, co)) <= op->tol) { if (v1 > op->feas_cert) { status = 3; break; } back = 1; }      /* KKT point of the elastic problem: certificate, or negligible violation */
      else if (op->feas_back > 0.0 && !feas_used && v1 + theq <= op->feas_back * th_entry) {      /* IPOPT leaves its restoration phase as soon as the violation has come down */
        back = 1; if (getenv("LO_TRACE")) fprintf(stderr, "   early return at it %d: theta %9.2e <= %g x %9.2e\n", it, v1 + theq, op->feas_back, th_entry);
      }
      else if (op->feas_stat > 0) {      /* stationary violation (include/landing_nlp.h): NOT a certificate */
        if (fstat < 0 || !(fabs(v1 - v1_ref) <= 0.05 * v1_ref)) { v1_ref = v1; fstat = 0; } else fstat++;
        if (fstat >= op->feas_stat && mu <= 1e-4 && pr <= 1e-3) {
          if (v1 <= op->feas_cert) back = 1;
          else if (op->feas_polish > 0.0 && !polished) {      /* once: the regularisation drops to feas_polish -- a stationary point is then a few Newton steps from the KKT point (status 3) */
            polished = 1; fstat = -1; delta_last = op->feas_polish / fdec; streak = 2; if (getenv("LO_TRACE")) fprintf(stderr, "   polish at it %d\n", it);
          }
          else if (op->feas_resume && !stalled) { stalled = 1; feas_used = 1; back = 1; if (getenv("LO_TRACE")) fprintf(stderr, "   stalled at it %d: the interior-point iteration resumes\n", it); }
          else { status = 4; break; }
        }
      }
      if (back) {
        feas = 0; W->feas = 0; lim = it + (op->max_iter > 1 ? op->max_iter : 1); if (lim > hard_lim) lim = hard_lim;
        if (op->feas_ret_push > 0.0) { mu = op->feas_ret_mu > 0.0 ? op->feas_ret_mu : op->mu_init; init_slacks_return(W, op, mu); } else { init_slacks(W, op); mu = op->mu_init; }
        fjam = 0; nfilt = 0; delta_last = 0.0; streak = 0; wd_count = 0; th_max = 0.0; nreset = 0; last_reset_it = it; ncrawl = 0; cutstreak = 0; force_step = 0;
        for (r = 12; r < ng; ++r) if (W->lb[r] == W->ub[r]) W->y[r] = 0.0;
        continue;
      }
      if (it >= lim) break;
      goto no_reset;
    }
    {
      int give_up = 0;
      if (fact_failed) { status = 2; fact_failed = 0; give_up = 1; }      /* no regularisation made the last step computable */
      else if (!(du < 1e300) || !(pr < 1e300) || !(co < 1e300)) { status = 2; give_up = 1; }
      else if (fmax(du, fmax(pr, co)) <= op->tol) { status = 0; break; }
      else if (it >= lim) give_up = 1;
      else if (du > op->reset_du && nreset >= op->max_resets && op->max_resets > 0) { status = 2; give_up = 1; }
      else if (op->feas_jam > 0 && fjam >= op->feas_jam && pr > 1e-3 && op->feas_phase) { give_up = 1; if (feas_used) stalled = 1;      /* (no entry left: the solve ends here, status 4) */ if (getenv("LO_TRACE")) fprintf(stderr, "   jammed line search at it %d\n", it); }
      if (give_up) {
        if (stalled) status = 4;      /* the point a stalled phase handed back did not lead anywhere either */
        if (!op->feas_phase || feas_used || op->max_iter < 1 || it >= hard_lim) break;
        /* feasibility phase: from the current point (from the caller's initial guess when the iterate is not finite) */
        fstat = -1;
        fdc = fdec;
        feas = 1; W->feas = 1; feas_used = (++n_feas >= op->feas_max); status = 1; fjam = 0; nfilt = 0; th_max = 0.0; delta_last = 0.0; streak = 0; lim = it + op->max_iter; if (lim > hard_lim) lim = hard_lim; cutstreak = 0; force_step = 0; wd_count = 0;
        { int bad = 0; for (i = 0; i < nx; ++i) if (!(fabs(W->x[i]) < 1e6)) bad = 1;
          if (bad) { memcpy(W->x, x0, sizeof(double) * nx); for (i = 0; i < 6; ++i) { W->x[i] = p[o.q_init + i]; W->x[6 + i] = p[o.qd_init + i]; } }
          eval_g(F, W->x, p, W->g); }
        mu = op->mu_init;
        th_entry = 0.0;      /* violation of the rows at the entry point (1-norm, equality rows included) */
        for (r = 12; r < ng; ++r) { const double lb = W->lb[r], ub = W->ub[r], g = W->g[r]; th_entry += (lb == ub) ? fabs(g - lb) : fmax(fmax(lb - g, g - ub), 0.0); }
        if (getenv("LO_TRACE")) fprintf(stderr, "   phase entry %d at it %d, theta %9.2e\n", n_feas, it, th_entry);
        for (r = 12; r < ng; ++r) {
          const double lb = W->lb[r], ub = W->ub[r], g = W->g[r];
          if (lb == ub) { W->y[r] = 0.0; continue; }
          /* slack on the row value; violation variables sized so that both distances start at a comfortable value */
          W->s[r] = g; W->zL[r] = W->zU[r] = W->wn[r] = W->wp[r] = W->en[r] = W->ep[r] = 0.0;
          if (lb > -INFINITY) { const double v = lb - g, n0 = fmax(v, 0.0) + fmax(1e-2, 0.1 * fabs(v)); W->en[r] = n0; W->zL[r] = fmin(mu / (g - lb + n0), 0.5 * frho); W->wn[r] = frho - W->zL[r]; }
          if (ub < INFINITY) { const double v = g - ub, p0 = fmax(v, 0.0) + fmax(1e-2, 0.1 * fabs(v)); W->ep[r] = p0; W->zU[r] = fmin(mu / (ub + p0 - g), 0.5 * frho); W->wp[r] = frho - W->zU[r]; }
          W->y[r] = W->zU[r] - W->zL[r];
        }
        continue;
      }
    }
    {
      const int stalled = op->restart_period > 0 && it - last_reset_it >= op->restart_period && mu >= op->mu_init && nreset < op->max_resets && ncrawl < ((op->fresh_restart & 4) ? 2 : 1);
      /* ... and a LATER barrier problem that is not solved 2 restart_period iterations after it began has wandered off (nothing else
       * catches that case: the dual infeasibility stays far below reset_du) -- restarted in place like a crawling iterate */
      const int lost = (op->fresh_restart & 8) && op->restart_period > 0 && mu < op->mu_init && pr > 1e-3 && nreset < op->max_resets &&
                       ((it - last_mu_it >= 2 * op->restart_period && it - last_reset_it >= op->restart_period) || wd_count >= 3);     /* ... or crawls on although the watchdog has fired three times */
      if (stalled) ncrawl++;
      if (!((du > op->reset_du && nreset < op->max_resets) || stalled || lost || (op->reset_delta > 0.0 && delta_last > op->reset_delta && nreset < op->max_resets))) goto no_reset;
      last_reset_it = it;
      nreset++;
      if (((op->fresh_restart & 2) && nreset == 2) || ((op->fresh_restart & 1) && nreset == 1 && !stalled && !lost)) {
        /* the restart in place did not help (second restart) or the iterate is jammed (multipliers blown up): back to the caller's
         * initial guess with another step rule -- the members that fail from it with clip_k = 4 solve with clip_k = 2 */
        memcpy(W->x, x0, sizeof(double) * nx);
        for (i = 0; i < 6; ++i) { W->x[i] = p[o.q_init + i]; W->x[6 + i] = p[o.qd_init + i]; }
        eval_g(F, W->x, p, W->g);
        clip_k_cur = clip_k_cur > 1 ? 2 : clip_k_cur; th_max = 0.0;
      }
      init_slacks(W, op); mu = op->mu_init; nfilt = 0; delta_last = 0.0; streak = 0; wd_count = 0;
      continue;
    }
    no_reset:;
    for (;;) {
      double cm = 0;
      for (r = 12; r < ng; ++r) {
        const double lb = W->lb[r], ub = W->ub[r];
        if (lb == ub) continue;
        if (feas) {
          if (lb > -INFINITY) { cm = fmax(cm, fabs((W->s[r] - lb + W->en[r]) * W->zL[r] - mu)); cm = fmax(cm, fabs(W->en[r] * W->wn[r] - mu)); }
          if (ub < INFINITY) { cm = fmax(cm, fabs((ub + W->ep[r] - W->s[r]) * W->zU[r] - mu)); cm = fmax(cm, fabs(W->ep[r] * W->wp[r] - mu)); }
          continue;
        }
        if (lb > -INFINITY) cm = fmax(cm, fabs((W->s[r] - lb) * W->zL[r] - mu));
        if (ub < INFINITY) cm = fmax(cm, fabs((ub - W->s[r]) * W->zU[r] - mu));
      }
      double sd = 1.0, sc = 1.0;
      if (op->barrier_smax > 0.0) {   /* IPOPT's scaling of the optimality error (s_d, s_c of Waechter & Biegler eq. 6) in the barrier-subproblem test */
        const double smax = op->barrier_smax; double ys = 0, zs = 0; long long nz = 0;
        for (r = 12; r < ng; ++r) { ys += fabs(W->y[r]); if (W->lb[r] != W->ub[r]) { if (W->lb[r] > -INFINITY) { zs += W->zL[r]; nz++; } if (W->ub[r] < INFINITY) { zs += W->zU[r]; nz++; } } }
        sd = fmax(smax, (ys + zs) / (double)(ng - 12 + nz)) / smax; sc = fmax(smax, zs / (double)nz) / smax;
      }
      if (fmax(du / sd, fmax(pr, cm / sc)) <= op->kappa_eps * mu && mu > op->tol / 10.0) { mu = fmax(op->tol / 10.0, fmin(op->kappa_mu * mu, pow(mu, op->theta_mu))); nfilt = 0; last_mu_it = it; wd_count = 0; }
      else break;
    }
    tau = fmax(op->tau_min, 1.0 - mu);
    for (r = 0; r < ng; ++r) {
      const double lb = W->lb[r], ub = W->ub[r]; double sg = 0, rh = 0;
      if (feas) {      /* elastic row: z + dz = (z + c) -/+ sigma ds after eliminating the violation variable and its multiplier */
        if (r >= 12 && lb != ub) {
          const double s = W->s[r];
          if (lb > -INFINITY) {
            const double n = W->en[r], a = s - lb + n, zl = W->zL[r], w = W->wn[r], D = a + zl * n / w;
            const double c = (mu - a * zl - zl * (mu - n * w + n * (zl + w - frho)) / w) / D;
            sg += zl / D; rh -= zl + c;
          }
          if (ub < INFINITY) {
            const double q = W->ep[r], b = ub + q - s, zu = W->zU[r], w = W->wp[r], D = b + zu * q / w;
            const double c = (mu - b * zu - zu * (mu - q * w + q * (zu + w - frho)) / w) / D;
            sg += zu / D; rh += zu + c;
          }
          rh += sg * (W->g[r] - s);
        }
      } else
      if (r >= 12 && lb != ub) {
        const double s = W->s[r];
        if (lb > -INFINITY) { const double d = s - lb; sg += W->zL[r] / d; rh -= mu / d; }
        if (ub < INFINITY) { const double d = ub - s; sg += W->zU[r] / d; rh += mu / d; }
        rh += sg * (W->g[r] - s);
      }
      W->sig[r] = sg; W->rho[r] = rh;
    }
    /* condensation per stage: M = H + Jd^T Sigma Jd (48x48), m = Jd^T rho, A^ = -J_dyn (state order), b */
    for (k = 0; k < N; ++k) {
      const int nr = lo_stage_rows(F, k), g0 = 36 + 104 * k; int q, a, b;
      const double* J = W->Jst + (size_t)k * 104 * 60; const double* H = W->Hst + (size_t)k * 3600;
      double* Mk = W->M + (size_t)k * NW * NW; double* mk = W->mvec + k * NW; double* Ah = W->Ah + (size_t)k * 432;
      memset(mk, 0, sizeof(double) * NW); memset(Ah, 0, sizeof(double) * 432);
      for (a = 0; a < 60; ++a) { const int wa = loc2w(a); if (wa < 0) continue; for (b = 0; b < 60; ++b) { const int wb = loc2w(b); if (wb >= 0) Mk[wa * NW + wb] = H[a * 60 + b]; } }
      for (q = 12; q < nr; ++q) {
        int idx[16], n = 0; double val[16]; const double sg = W->sig[g0 + q], rh = W->rho[g0 + q];
        for (a = 0; a < 60; ++a) if (J[q * 60 + a] != 0.0 && loc2w(a) >= 0) { idx[n] = loc2w(a); val[n] = J[q * 60 + a]; ++n; }
        for (a = 0; a < n; ++a) { mk[idx[a]] += rh * val[a]; for (b = 0; b < n; ++b) Mk[idx[a] * NW + idx[b]] += sg * val[a] * val[b]; }
      }
      for (q = 0; q < 12; ++q) { for (a = 0; a < 36; ++a) Ah[ROW2STATE[q] * 36 + a] = -J[q * 60 + a]; W->bv[k * 12 + ROW2STATE[q]] = -W->g[g0 + q]; }
      if (F->run_cost) {   /* objective gradient of the stage variables enters the stage right-hand side (w order: X, c, f) */
        double gr[36]; for (a = 0; a < 36; ++a) gr[a] = 0.0;
        (void)lo_run_cost_stage(F, W->x, p, k, gr, gr + 12, gr + 24);
        for (a = 0; a < 36; ++a) mk[a] += gr[a];
      }
    }
    /* factorisation with inertia correction (same schedule as the HIP kernel) */
    delta = (streak >= 2 && delta_last > 0.0) ? fmax(1e-20, delta_last * (feas ? fdc : op->delta_dec)) : 0.0;
    if (!F->run_cost && !feas) {
      double fl = op->delta_floor;
      if (stag_k > 0 && stag >= stag_k) { int e; for (e = stag - stag_k; e >= 0; --e) fl *= 0.1; if (fl < 1e-12) fl = 0.0; }
      delta = fmax(delta, fl);
    }
    for (attempt = 0; attempt < 60 && !fact_ok; ++attempt) {
      if (attempt > 0) {
        if (delta == 0.0) delta = (delta_last == 0.0) ? op->delta_init : fmax(1e-20, delta_last * (feas ? fdc : op->delta_dec));
        else if (feas && adapt && attempt == 1 && delta < delta_last) delta = delta_last;      /* the regularisation of the last iteration is the best guess of what this one needs */
        else delta *= (delta_last == 0.0 ? op->delta_inc_first : op->delta_inc);
        if (delta > 1e40) break;
      }
      counters[0]++;
      fact_ok = riccati_backward(F, p, W, delta, &o, sig);
    }
    if (!fact_ok) { status = stalled ? 4 : 2; if (op->feas_phase && !feas_used && !feas) { fact_failed = 1; continue; } break; }
    if (feas && adapt) fdc = attempt <= 1 ? fmax(fdec, fdc * fdc) : fmin(0.7, sqrt(fdc));      /* (attempt counts the factorisations of this iteration) */
    if (delta > 0.0) { delta_last = delta; streak++; } else streak = 0;
    if (streak > 8) streak = (feas && adapt) ? 2 : 0;      /* (the elastic problem has no objective: delta_w = 0 is not probed again inside the phase) */
    /* forward sweep */
    for (k = 0; k < N; ++k) {
      const int last = (k == N - 1), nu = last ? 12 : 24, nr = lo_stage_rows(F, k), g0 = 36 + 104 * k; int q, a, t;
      const double* J = W->Jst + (size_t)k * 104 * 60; double signext[24];
      for (a = 0; a < 24; ++a) w[a] = sig[a];
      for (a = 0; a < nu; ++a) { double v = W->kap[k * 24 + a]; for (t = 0; t < 24; ++t) v += W->K[(size_t)k * 576 + a * 24 + t] * sig[t]; w[24 + a] = -v; }
      for (a = nu; a < 24; ++a) w[24 + a] = 0.0;
      for (a = 0; a < 12; ++a) { W->dx[12 * k + a] = w[a]; W->dx[12 * (N + 1) + 24 * k + a] = w[12 + a]; W->dx[12 * (N + 1) + 24 * k + 12 + a] = w[24 + a]; }
      for (q = 12; q < nr; ++q) {
        double v = 0; for (a = 0; a < 60; ++a) { const int wa = loc2w(a); if (wa >= 0 && J[q * 60 + a] != 0.0) v += J[q * 60 + a] * w[wa]; }
        W->ds[g0 + q] = v + (W->g[g0 + q] - W->s[g0 + q]);
      }
      for (a = 0; a < 12; ++a) { double v = W->bv[k * 12 + a]; for (t = 0; t < 36; ++t) v += W->Ah[(size_t)k * 432 + a * 36 + t] * w[t]; signext[a] = v; }
      for (a = 0; a < 12; ++a) signext[12 + a] = last ? 0.0 : w[36 + a];
      for (a = 0; a < 12; ++a) {
        double v = W->pvx[(k + 1) * 12 + a]; const int nn = last ? 12 : 24;
        for (t = 0; t < nn; ++t) v += W->Px[(size_t)(k + 1) * 288 + a * 24 + t] * signext[t];
        W->yn[g0 + (a < 6 ? a : (a < 9 ? a + 3 : a - 3))] = -v;
      }
      memcpy(sig, signext, sizeof(sig));
    }
    for (i = 0; i < 12; ++i) {
      const int ra = i < 6 ? 12 + i : 24 + (i - 6), rb = i < 6 ? 18 + i : 30 + (i - 6);
      W->dx[12 * N + i] = sig[i];
      W->ds[ra] = sig[i] + (W->g[ra] - W->s[ra]); W->ds[rb] = sig[i] + (W->g[rb] - W->s[rb]);
    }
    /* dual steps, step bounds, merit data.  clip_now: the primal step length comes from the clip_k-th largest ratio
     * |ds| / distance (top[] holds the four largest); the slacks with a larger ratio are clipped in slack_step(). */
    clip_now = !feas && clip_k_cur > 1 && (pr > op->clip_until || (jam_k > 0 && jamrun >= jam_k));
    top[0] = top[1] = top[2] = top[3] = 0.0;
    for (r = 12; r < ng; ++r) {
      const double lb = W->lb[r], ub = W->ub[r], g = W->g[r]; double s, ds, yn;
      if (lb == ub) { th0 += fabs(g - lb); continue; }
      s = W->s[r]; ds = W->ds[r]; th0 += fabs(g - s); yn = W->sig[r] * ds;
      if (feas) {      /* steps of the eliminated variables, step bounds (a, n, b, p and their multipliers stay positive), merit data */
        W->dzL[r] = W->dzU[r] = W->den[r] = W->dep[r] = W->dwn[r] = W->dwp[r] = 0.0;
        if (lb > -INFINITY) {
          const double n = W->en[r], a = s - lb + n, zl = W->zL[r], w = W->wn[r], D = a + zl * n / w, rn = zl + w - frho;
          const double dz = (mu - a * zl - zl * (mu - n * w + n * rn) / w - zl * ds) / D, dw = -dz - rn, dn = (mu - n * w - n * dw) / w, da = ds + dn;
          W->dzL[r] = dz; W->dwn[r] = dw; W->den[r] = dn;
          if (da < 0.0) a_pr = fmin(a_pr, -tau * a / da);
          if (dn < 0.0) a_pr = fmin(a_pr, -tau * n / dn);
          if (dz < 0.0) a_du = fmin(a_du, -tau * zl / dz);
          if (dw < 0.0) a_du = fmin(a_du, -tau * w / dw);
          bar -= log(a) + log(n); dphi += frho * dn - mu * (da / a + dn / n); f0 += frho * n;
        }
        if (ub < INFINITY) {
          const double q = W->ep[r], b = ub + q - s, zu = W->zU[r], w = W->wp[r], D = b + zu * q / w, rp = zu + w - frho;
          const double dz = (mu - b * zu - zu * (mu - q * w + q * rp) / w + zu * ds) / D, dw = -dz - rp, dq = (mu - q * w - q * dw) / w, db = dq - ds;
          W->dzU[r] = dz; W->dwp[r] = dw; W->dep[r] = dq;
          if (db < 0.0) a_pr = fmin(a_pr, -tau * b / db);
          if (dq < 0.0) a_pr = fmin(a_pr, -tau * q / dq);
          if (dz < 0.0) a_du = fmin(a_du, -tau * zu / dz);
          if (dw < 0.0) a_du = fmin(a_du, -tau * w / dw);
          bar -= log(b) + log(q); dphi += frho * dq - mu * (db / b + dq / q); f0 += frho * q;
        }
        continue;
      }
      if (lb > -INFINITY) {
        const double d = s - lb, zl = W->zL[r], dz = mu / d - zl - zl / d * ds;
        W->dzL[r] = dz; yn -= mu / d;
        if (ds < 0.0) { a_pr = fmin(a_pr, -tau * d / ds); top4_push(top, -ds / d); }
        if (dz < 0.0) a_du = fmin(a_du, -tau * zl / dz);
        bar -= log(d); dphi -= mu * ds / d;
      } else W->dzL[r] = 0.0;
      if (ub < INFINITY) {
        const double d = ub - s, zu = W->zU[r], dz = mu / d - zu + zu / d * ds;
        W->dzU[r] = dz; yn += mu / d;
        if (ds > 0.0) { a_pr = fmin(a_pr, tau * d / ds); top4_push(top, ds / d); }
        if (dz < 0.0) a_du = fmin(a_du, -tau * zu / dz);
        bar -= log(d); dphi += mu * ds / d;
      } else W->dzU[r] = 0.0;
      W->yn[r] = yn;
    }
    if (!feas) for (i = 0; i < 12; ++i) { const double d = W->x[12 * N + i] - p[12 * N + i], qn = p[o.QN + i]; f0 += qn * d * d; dphi += 2.0 * qn * d * W->dx[12 * N + i]; }
    if (F->run_cost && !feas) for (k = 0; k < N; ++k) {
      double gX[12] = {0}, gc[12] = {0}, gf[12] = {0}; const double* dX = W->dx + 12 * k; const double* dU = W->dx + 12 * (N + 1) + 24 * k;
      f0 += lo_run_cost_stage(F, W->x, p, k, gX, gc, gf);
      for (i = 0; i < 12; ++i) dphi += gX[i] * dX[i] + gc[i] * dU[i] + gf[i] * dU[12 + i];
    }
    ph0 = f0 + mu * bar;
    if (th_max == 0.0) th_max = 1e4 * fmax(1.0, th0);
    if (clip_now) { const double rk = top[(clip_k_cur > 4 ? 4 : clip_k_cur) - 1]; a_pr = rk > tau ? tau / rk : 1.0; }
    alpha = a_pr;
    while (alpha > 1e-10) {
      double tht = 0, bt = 0, ft = 0, pht; int ok_f, e, switching;
      counters[1]++;
      for (i = 0; i < nx; ++i) W->xt[i] = W->x[i] + alpha * W->dx[i];
      eval_g(F, W->xt, p, W->gt);
      for (r = 12; r < ng; ++r) {
        const double lb = W->lb[r], ub = W->ub[r], g = W->gt[r]; double s;
        if (lb == ub) { tht += fabs(g - lb); continue; }
        if (feas) {
          s = W->s[r] + alpha * W->ds[r]; tht += fabs(g - s);
          if (lb > -INFINITY) { const double n = W->en[r] + alpha * W->den[r]; bt -= log(s - lb + n) + log(n); ft += frho * n; }
          if (ub < INFINITY) { const double q = W->ep[r] + alpha * W->dep[r]; bt -= log(ub + q - s) + log(q); ft += frho * q; }
          continue;
        }
        s = slack_step(W->s[r], W->ds[r], alpha, lb, ub, clip_now, tau); tht += fabs(g - s);
        if (lb > -INFINITY) bt -= log(s - lb);
        if (ub < INFINITY) bt -= log(ub - s);
      }
      if (!feas) for (i = 0; i < 12; ++i) { const double d = W->xt[12 * N + i] - p[12 * N + i]; ft += p[o.QN + i] * d * d; }
      if (F->run_cost && !feas) for (k = 0; k < N; ++k) ft += lo_run_cost_stage(F, W->xt, p, k, NULL, NULL, NULL);
      pht = ft + mu * bt;
      ok_f = (tht <= th_max) && (pht < 1e300) && (pht > -1e300) && (tht < 1e300);
      for (e = 0; e < nfilt && ok_f; ++e) if (tht >= fmax(filt_th[e], th_floor) && pht >= filt_ph[e]) ok_f = 0;
      switching = (dphi < 0.0) && (th0 <= 1e-4) && (alpha * pow(-dphi, 2.3) > pow(th0, 1.1));
      if (ok_f) {
        /* th_floor: constraint violations below the convergence tolerance count as equal (trial points that stay below it are
         * never rejected for their theta): at the last barrier problems theta sits at 1e-7 while the dual infeasibility still needs
         * full Newton steps, and the relative-decrease test alone cuts those to 1/64 (212 instead of 99 iterations on one member
         * of the bench batches) */
        if (switching) { if (pht <= ph0 + 1e-8 * alpha * dphi) { accepted = 1; armijo = 1; } }
        else if (tht <= fmax((1.0 - 1e-5) * th0, th_floor) || pht <= ph0 - 1e-8 * th0) accepted = 1;
      }
      if (force_step && ok_f) { accepted = 1; nfilt = 0; break; }      /* watchdog: the step to the boundary is taken without the sufficient-decrease / switching tests (it passed theta_max and the filter entries: ok_f) */
      if (accepted) break;
      if (!feas && op->max_soc > 0 && alpha == a_pr && tht >= th0) {   /* LAB ONLY: second-order corrections (IPOPT A-5.5..5.9) at the rejected first trial point */
        double* cs = dalloc(ng); double* dx2 = dalloc(nx); double* ds2 = dalloc(ng); double* yn2 = dalloc(ng); double* xt2 = dalloc(nx); double* gt2 = dalloc(ng);
        double th_soc = th0; int ps, took = 0;
        for (r = 12; r < ng; ++r) {
          const double lb = W->lb[r], ub = W->ub[r];
          if (lb == ub) cs[r] = alpha * (W->g[r] - lb) + (W->gt[r] - lb);
          else cs[r] = alpha * (W->g[r] - W->s[r]) + (W->gt[r] - slack_step(W->s[r], W->ds[r], alpha, lb, ub, clip_now, tau));
        }
        for (ps = 0; ps < op->max_soc && !took; ++ps) {
          double a2 = 1.0, tht2 = 0, bt2 = 0, ft2 = 0, pht2; int okf2;
#pragma omp atomic
          lo_soc_tried++;
          counters[0]++;
          if (!soc_solve(F, p, W, &o, mu, delta, cs, dx2, ds2, yn2)) break;
          for (r = 12; r < ng; ++r) {
            const double lb = W->lb[r], ub = W->ub[r];
            if (lb == ub) continue;
            if (lb > -INFINITY && ds2[r] < 0.0) a2 = fmin(a2, -tau * (W->s[r] - lb) / ds2[r]);
            if (ub < INFINITY && ds2[r] > 0.0) a2 = fmin(a2, tau * (ub - W->s[r]) / ds2[r]);
          }
          counters[1]++;
          for (i = 0; i < nx; ++i) xt2[i] = W->x[i] + a2 * dx2[i];
          eval_g(F, xt2, p, gt2);
          for (r = 12; r < ng; ++r) {
            const double lb = W->lb[r], ub = W->ub[r], g = gt2[r]; double s2;
            if (lb == ub) { tht2 += fabs(g - lb); continue; }
            s2 = W->s[r] + a2 * ds2[r]; tht2 += fabs(g - s2);
            if (lb > -INFINITY) bt2 -= log(s2 - lb);
            if (ub < INFINITY) bt2 -= log(ub - s2);
          }
          for (i = 0; i < 12; ++i) { const double d = xt2[12 * N + i] - p[12 * N + i]; ft2 += p[o.QN + i] * d * d; }
          if (F->run_cost) for (k = 0; k < N; ++k) ft2 += lo_run_cost_stage(F, xt2, p, k, NULL, NULL, NULL);
          pht2 = ft2 + mu * bt2;
          okf2 = (tht2 <= th_max) && (pht2 < 1e300) && (pht2 > -1e300) && (tht2 < 1e300);
          for (e = 0; e < nfilt && okf2; ++e) if (tht2 >= fmax(filt_th[e], th_floor) && pht2 >= filt_ph[e]) okf2 = 0;
          if (okf2) {
            if (switching) { if (pht2 <= ph0 + 1e-8 * alpha * dphi) { took = 1; armijo = 1; } }
            else if (tht2 <= fmax((1.0 - 1e-5) * th0, th_floor) || pht2 <= ph0 - 1e-8 * th0) took = 1;
          }
          if (took) {      /* the corrected step replaces the search direction: primal step, slack steps, multipliers of the equality rows; bound multipliers from the slack steps */
            memcpy(W->dx, dx2, sizeof(double) * nx); memcpy(W->xt, xt2, sizeof(double) * nx); memcpy(W->gt, gt2, sizeof(double) * ng);
            a_du = 1.0;
            for (r = 12; r < ng; ++r) {
              const double lb = W->lb[r], ub = W->ub[r]; double yv;
              W->ds[r] = ds2[r];
              if (lb == ub) { W->yn[r] = yn2[r]; continue; }
              yv = W->sig[r] * ds2[r];
              if (lb > -INFINITY) { const double d = W->s[r] - lb, zl = W->zL[r], dz = mu / d - zl - zl / d * ds2[r]; W->dzL[r] = dz; yv -= mu / d; if (dz < 0.0) a_du = fmin(a_du, -tau * zl / dz); }
              if (ub < INFINITY) { const double d = ub - W->s[r], zu = W->zU[r], dz = mu / d - zu + zu / d * ds2[r]; W->dzU[r] = dz; yv += mu / d; if (dz < 0.0) a_du = fmin(a_du, -tau * zu / dz); }
              W->yn[r] = yv;
            }
            alpha = a2; clip_now = 0;
#pragma omp atomic
            lo_soc_taken++;
            break;
          }
          if (tht2 > 0.99 * th_soc) break;      /* kappa_soc: the correction did not reduce the violation */
          th_soc = tht2;
          for (r = 12; r < ng; ++r) {
            const double lb = W->lb[r], ub = W->ub[r];
            cs[r] = a2 * cs[r] + ((lb == ub) ? gt2[r] - lb : gt2[r] - (W->s[r] + a2 * ds2[r]));
          }
        }
        free(cs); free(dx2); free(ds2); free(yn2); free(xt2); free(gt2);
        if (took) { accepted = 1; break; }
      }
      if (!feas && op->slack_corr > 0.0 && alpha == a_pr && tht >= th0) {   /* slack correction at the rejected first trial point (include/landing_nlp.h): no new solve */
        const double kk = op->slack_corr; double tht2 = 0, bt2 = 0, pht2; int okf2;
        for (r = 12; r < ng; ++r) {
          const double lb = W->lb[r], ub = W->ub[r], g = W->gt[r]; double s2;
          if (lb == ub) { tht2 += fabs(g - lb); continue; }
          s2 = slack_reset(slack_step(W->s[r], W->ds[r], alpha, lb, ub, clip_now, tau), g, lb, ub, kk); tht2 += fabs(g - s2);
          if (lb > -INFINITY) bt2 -= log(s2 - lb);
          if (ub < INFINITY) bt2 -= log(ub - s2);
        }
        pht2 = ft + mu * bt2;
        okf2 = (tht2 <= th_max) && (pht2 < 1e300) && (pht2 > -1e300);
        for (e = 0; e < nfilt && okf2; ++e) if (tht2 >= fmax(filt_th[e], th_floor) && pht2 >= filt_ph[e]) okf2 = 0;
        if (okf2 && (tht2 <= fmax((1.0 - 1e-5) * th0, th_floor) || pht2 <= ph0 - 1e-8 * th0)) { accepted = 1; use_reset = kk; break; }
      }
      alpha *= 0.5;
    }
    /* watchdog (cf. IPOPT's watchdog_shortened_iter_trigger): after `watchdog` successive iterations whose accepted step length is at most
     * 1/16 of the step to the boundary the next iteration takes that step unconditionally and restarts the filter */
    force_step = 0;
    if (op->watchdog > 0 && !feas) {
      if (accepted && alpha <= 0.0625 * a_pr) { if (++cutstreak >= op->watchdog) { force_step = 1; cutstreak = 0; wd_count++; } }
      else cutstreak = 0;
    }
    if (!accepted) {
      nfilt = 0; alpha = fmin(a_pr, op->alpha_fallback);
      for (i = 0; i < nx; ++i) W->xt[i] = W->x[i] + alpha * W->dx[i];
      eval_g(F, W->xt, p, W->gt);
    } else if (!armijo) {
      if (nfilt == 64) { memmove(filt_th, filt_th + 1, 63 * sizeof(double)); memmove(filt_ph, filt_ph + 1, 63 * sizeof(double)); nfilt = 63; }
      filt_th[nfilt] = (1.0 - 1e-5) * th0; filt_ph[nfilt] = ph0 - 1e-8 * th0; nfilt++;
    }
    if (getenv("LO_TRACE")) fprintf(stderr, "      alpha %9.2e a_pr %9.2e a_du %9.2e delta %8.1e acc %d armijo %d th0 %9.2e dphi %9.2e clip %d\n", alpha, a_pr, a_du, delta, accepted, armijo, th0, dphi, clip_now);
    if (op->dual_step_cap > 0.0) a_du = fmin(a_du, op->dual_step_cap * alpha);      /* the multipliers do not run ahead of a blocked primal step */
    if (jam_k > 0) { if (!clip_now && a_pr < jam_a) jamrun++; else jamrun = 0; }
    if (op->feas_jam > 0) { if (!feas && alpha < 1e-2) fjam++; else fjam = fjam > 2 ? fjam - 2 : 0; }
    full_prev = accepted && alpha == 1.0 && a_du == 1.0 && attempt <= 1;
    memcpy(W->x, W->xt, sizeof(double) * nx);
    for (r = 0; r < ng; ++r) {
      const double lb = W->lb[r], ub = W->ub[r]; double s, zl = 0, zu = 0;
      W->g[r] = W->gt[r];
      if (r < 12) continue;
      if (lb == ub) { W->y[r] += alpha * (W->yn[r] - W->y[r]); continue; }
      if (feas) {
        s = W->s[r] + alpha * W->ds[r];
        if (lb > -INFINITY) {
          const double n = W->en[r] + alpha * W->den[r], a = s - lb + n;
          zl = W->zL[r] + a_du * W->dzL[r]; zl = fmin(fmax(zl, mu / (1e10 * a)), 1e10 * mu / a);
          W->wn[r] = fmin(fmax(W->wn[r] + a_du * W->dwn[r], mu / (1e10 * n)), 1e10 * mu / n); W->en[r] = n;
        }
        if (ub < INFINITY) {
          const double q = W->ep[r] + alpha * W->dep[r], b = ub + q - s;
          zu = W->zU[r] + a_du * W->dzU[r]; zu = fmin(fmax(zu, mu / (1e10 * b)), 1e10 * mu / b);
          W->wp[r] = fmin(fmax(W->wp[r] + a_du * W->dwp[r], mu / (1e10 * q)), 1e10 * mu / q); W->ep[r] = q;
        }
        W->s[r] = s; W->zL[r] = zl; W->zU[r] = zu; W->y[r] = zu - zl;
        continue;
      }
      s = slack_step(W->s[r], W->ds[r], alpha, lb, ub, clip_now, tau);
      if (use_reset > 0.0) s = slack_reset(s, W->gt[r], lb, ub, use_reset);
      if (lb > -INFINITY) { const double d = s - lb; zl = W->zL[r] + a_du * W->dzL[r]; zl = fmin(fmax(zl, mu / (1e10 * d)), 1e10 * mu / d); }
      if (ub < INFINITY) { const double d = ub - s; zu = W->zU[r] + a_du * W->dzU[r]; zu = fmin(fmax(zu, mu / (1e10 * d)), 1e10 * mu / d); }
      W->s[r] = s; W->zL[r] = zl; W->zU[r] = zu; W->y[r] = zu - zl;
    }
  }
  for (i = 0; i < 12; ++i) W->y[i] = -gx[i];
  memcpy(x_out, W->x, sizeof(double) * nx);
  if (lam_out) memcpy(lam_out, W->y, sizeof(double) * ng);
  if (iters_out) *iters_out = it;
  if (kkt_out) { lo_kkt(F, W->x, p, W->y, kkt_out); (void)e_du; }
  free(W->x); free(W->xt); free(W->dx); free(gx); free(W->g); free(W->gt); free(W->s); free(W->ds); free(W->zL); free(W->zU);
  free(W->dzL); free(W->dzU); free(W->y); free(W->yn); free(W->lb); free(W->ub); free(W->sig); free(W->rho); free(W->Jst); free(W->Hst);
  free(W->en); free(W->ep); free(W->wn); free(W->wp); free(W->den); free(W->dep); free(W->dwn); free(W->dwp);
  free(W->M); free(W->mvec); free(W->Ah); free(W->bv); free(W->K); free(W->kap); free(W->Px); free(W->pvx);
  return status;
}

/* batch driver: B members, `threads` OpenMP threads (<=0: all).  counters[0..1] = total factorisations, trial points. */
int lo_solve_batch(const lo_form* F, int B, const double* p, const double* x0, const lo_solver_opts* opts, int threads,
                   double* x, double* lam_g, int* status, int* iters, double* kkt, long long* counters) {
  const lo_int nx = lo_nx(F->N), ng = lo_ng(F->N), np = lo_np_form(F);
  lo_solver_opts o; long long c0 = 0, c1 = 0; int b;
  if (opts) o = *opts; else lo_solver_opts_default(&o);
  if (!(o.kappa_eps > 0.0)) o.kappa_eps = F->run_cost ? 10.0 : 120.0;     /* automatic choice by formulation (include/landing_nlp.h) */
  if (!(o.theta_mu > 0.0)) o.theta_mu = F->run_cost ? 1.5 : 1.8;
  if (!(o.mu_init > 0.0)) o.mu_init = F->run_cost ? 0.1 : 0.5;
  if (!(o.bound_push > 0.0)) o.bound_push = F->run_cost ? 0.5 : 1.0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : c0, c1)
  for (b = 0; b < B; ++b) {
    long long cc[2] = {0, 0};
    status[b] = solve_one(F, p + (size_t)b * np, x0 + (size_t)b * nx, &o, x + (size_t)b * nx, lam_g ? lam_g + (size_t)b * ng : NULL,
                          iters ? iters + b : NULL, kkt ? kkt + 3 * (size_t)b : NULL, cc);
    c0 += cc[0]; c1 += cc[1];
  }
  if (counters) { counters[0] = c0; counters[1] = c1; }
  return 0;
}

/* Full derivative sweeps of a batch (g, grad f, Jacobian and Hessian nonzeros of every member; SURVEY 8d unit of work for
 * the function layer), `reps` times, OpenMP over members: CPU timing leg of bench.py.  Outputs go to per-thread scratch. */
int lo_sweep_batch(const lo_form* F, int B, const double* x, const double* p, const double* lam_g, int reps, int threads) {
  const lo_int nx = lo_nx(F->N), ng = lo_ng(F->N), np = lo_np_form(F), nj = lo_nnz_jac(F->N), nh = lo_nnz_hess(F->N);
  int bad = 0;
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel reduction(+ : bad)
  {
    double* g = (double*)malloc(sizeof(double) * (size_t)(ng + nx + nj + nh));
    double f;
    int r, b;
    if (!g) bad = 1;
    else {
      double *gf = g + ng, *J = gf + nx, *H = J + nj;
      for (r = 0; r < reps; ++r) {
#pragma omp for schedule(static) nowait
        for (b = 0; b < B; ++b) {
          const double* xb = x + (size_t)b * nx; const double* pb = p + (size_t)b * np;
          lo_nlp_grad_f(F, xb, pb, &f, gf);
          lo_nlp_jac_g(F, xb, pb, g, J);
          lo_nlp_hess_l(F, xb, pb, 1.0, lam_g + (size_t)b * ng, H);
          if (!(f == f)) bad += 1;
        }
      }
      free(g);
    }
  }
  return bad;
}
