/*
 * ipm_lab.c -- DEVELOPMENT LAB (test infrastructure, not product, not the cpu_baseline): a copy of
 * oracle/landing_solver_cpu.c with experimental switches read from the environment, used to measure what an
 * algorithmic change does to the iteration-count distribution of the bench batch before it is written into
 * the HIP kernel.  Build / run: tests/dev/ipm_lab.py.
 *   LAB_THMIN=1     theta_min = 1e-4 max(1, theta(x0))   (IPOPT; the port uses 1e-4)
 *   LAB_SOC=n       up to n second-order corrections per iteration (IPOPT max_soc; reference: 4)
 *   LAB_STICKY=1    stage-sticky regularisation: a failed stage elimination is retried in place with a larger
 *                   delta_w, which then stays for the remaining stages of the sweep (no restart of the sweep)
 *   LAB_STALL=mode  0: port's rule; 1: no crawl restart when the barrier problem is about to be solved
 *   LAB_PROBE=1     Mehrotra probing for mu (adaptive), monotone fallback
 */
#include "../../oracle/landing_oracle.h"
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
} lo_solver_opts;

void lo_solver_opts_default(lo_solver_opts* o) {
  o->tol = 1e-6; o->max_iter = 3000; o->mu_init = 0.1; o->bound_push = 0.5; o->bound_frac = 0.1;
  o->kappa_eps = 10.0; o->kappa_mu = 0.2; o->theta_mu = 1.5; o->max_resets = 8; o->reset_du = 1e9;
  o->delta_init = 1e-4; o->delta_inc_first = 10.0; o->delta_inc = 4.0; o->delta_dec = 1.0 / 3.0; o->tau_min = 0.9; o->alpha_fallback = 1e-2; o->restart_period = 80; o->reset_delta = 1e5;
}

#define NW 48
static const int ROW2STATE[12] = {0, 1, 2, 3, 4, 5, 9, 10, 11, 6, 7, 8};
/* local variable (lo_stage_eval order X_k,c_k,f_k,X+,c+) -> w index (X,c,f,c+) or -1 */
static int loc2w(int loc) { if (loc < 36) return loc; if (loc < 48) return -1; return 36 + (loc - 48); }

typedef struct {
  int N; lo_int nx, ng;
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
static _Thread_local double t_minpiv, t_failpiv;
static int spd_solve(double* A, int lda, int n, double* B, int ldb, int m) {
  int i, j, c;
  for (j = 0; j < n; ++j) {
    const double d = A[j * lda + j];
    if (!(d > 0.0) || !(d < 1e300)) { t_failpiv = d; return 0; }
    if (d < t_minpiv) t_minpiv = d;
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

/* LAB_FP32: the stage elimination in single precision (BASELINE configs[4] names an fp32 matrix-core KKT factor): G, gamma, the
 * cost-to-go P, p and the gains are rounded to float, the products T^T P T accumulate in float, the LDL^T runs in float.  Everything
 * outside the factorisation (residuals, right-hand sides, forward sweep) stays fp64: inexact Newton with exact residuals. */
static int lab_fp32(void);
#define F32(v) ((double)(float)(v))
static int spd_solve_f(double* A, int lda, int n, double* B, int ldb, int m) {
  int i, j, c;
  for (j = 0; j < n; ++j) {
    const float d = (float)A[j * lda + j];
    if (!(d > 0.0f) || !(d < 1e30f)) { t_failpiv = d; return 0; }
    if (d < t_minpiv) t_minpiv = d;
    for (i = j + 1; i < n; ++i) {
      const float l = (float)A[i * lda + j] / d;
      if (l == 0.0f) continue;
      for (c = j + 1; c < n; ++c) A[i * lda + c] = (float)A[i * lda + c] - l * (float)A[j * lda + c];
      for (c = 0; c < m; ++c) B[i * ldb + c] = (float)B[i * ldb + c] - l * (float)B[j * ldb + c];
      A[i * lda + j] = l;
    }
  }
  for (j = n - 1; j >= 0; --j) {
    for (c = 0; c < m; ++c) {
      float v = (float)B[j * ldb + c] / (float)A[j * lda + j];
      for (i = j + 1; i < n; ++i) v -= (float)A[i * lda + j] * (float)B[i * ldb + c];
      B[j * ldb + c] = v;
    }
  }
  return 1;
}
/* backward Riccati sweep; returns 1 on success */
typedef struct { int sticky; double delta_last, delta_init, inc_first, inc, dec; long long nstage; int fail_stage; } reg_t;
static double next_delta(double delta, const reg_t* R) {
  if (delta == 0.0) return (R->delta_last == 0.0) ? R->delta_init : fmax(1e-20, R->delta_last * R->dec);
  return delta * (R->delta_last == 0.0 ? R->inc_first : R->inc);
}
/* *pdelta: in = regularisation to start with, out = the largest one used (sticky mode) */
static int riccati_backward(const lo_form* F, const double* p, work_t* W, double* pdelta, const lo_poff* o, double* sig0, reg_t* RG) {
  double delta = *pdelta;
  const int N = W->N;
  double P[24 * 24], pv[24], G[NW * NW], gam[NW], Y[24 * 36], q[24], Guu[24 * 24], R[24 * 25];
  int k, i, j, t;
  memset(P, 0, sizeof(P)); memset(pv, 0, sizeof(pv));
  for (i = 0; i < 12; ++i) {
    const int ra = i < 6 ? 12 + i : 24 + (i - 6), rb = i < 6 ? 18 + i : 30 + (i - 6);
    const double qn2 = 2.0 * p[o->QN + i];
    P[i * 24 + i] = qn2 + W->sig[ra] + W->sig[rb] + delta;
    pv[i] = qn2 * (W->x[12 * N + i] - p[12 * N + i]) + W->rho[ra] + W->rho[rb];
  }
  for (i = 0; i < 12; ++i) { for (j = 0; j < 24; ++j) W->Px[(size_t)N * 288 + i * 24 + j] = P[i * 24 + j]; W->pvx[N * 12 + i] = pv[i]; }
  for (k = N - 1; k >= 0; --k) {
    const int last = (k == N - 1), nu = last ? 12 : 24, nsn = last ? 12 : 24, nw = 24 + nu;
    const double* Mk = W->M + (size_t)k * NW * NW; const double* Ah = W->Ah + (size_t)k * 432; const double* bv = W->bv + k * 12;
    retry_stage:
    RG->nstage++;
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
    if (lab_fp32()) { for (i = 0; i < NW * NW; ++i) G[i] = F32(G[i]); for (i = 0; i < NW; ++i) gam[i] = F32(gam[i]); }
    /* K = Guu^-1 [Gus | gam_u] */
    for (i = 0; i < nu; ++i) {
      for (j = 0; j < nu; ++j) Guu[i * 24 + j] = G[(24 + i) * NW + 24 + j];
      for (j = 0; j < 24; ++j) R[i * 25 + j] = G[(24 + i) * NW + j];
      R[i * 25 + 24] = gam[24 + i];
    }
    if (!(lab_fp32() ? spd_solve_f(Guu, 24, nu, R, 25, 25) : spd_solve(Guu, 24, nu, R, 25, 25))) {
      RG->fail_stage = k;
      if (!RG->sticky) return 0;
      delta = next_delta(delta, RG);
      if (delta > 1e40) return 0;
      goto retry_stage;
    }
    for (i = 0; i < nu; ++i) { for (j = 0; j < 24; ++j) W->K[(size_t)k * 576 + i * 24 + j] = R[i * 25 + j]; W->kap[k * 24 + i] = R[i * 25 + 24]; }
    for (i = 0; i < 24; ++i) {
      for (j = 0; j < 24; ++j) { double a = G[i * NW + j]; for (t = 0; t < nu; ++t) a -= G[(24 + t) * NW + i] * R[t * 25 + j]; P[i * 24 + j] = a; }
      { double a = gam[i]; for (t = 0; t < nu; ++t) a -= G[(24 + t) * NW + i] * R[t * 25 + 24]; pv[i] = a; }
    }
    if (lab_fp32()) { for (i = 0; i < 576; ++i) P[i] = F32(P[i]); for (i = 0; i < 24; ++i) pv[i] = F32(pv[i]); }
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
    {
      double extra = 0.0;
      for (;;) {
        double Pc2[144], r2[12];
        memcpy(Pc2, Pcc, sizeof(Pc2)); memcpy(r2, rhs, sizeof(r2));
        for (i = 0; i < 12; ++i) Pc2[i * 12 + i] += extra;
        RG->nstage++;
        if (spd_solve(Pc2, 12, 12, r2, 1, 1)) { memcpy(rhs, r2, sizeof(rhs)); break; }
        RG->fail_stage = -1;
        if (!RG->sticky) return 0;
        { const double nd = next_delta(delta, RG); extra += nd - delta; delta = nd; }
        if (delta > 1e40) return 0;
      }
    }
    for (i = 0; i < 12; ++i) sig0[12 + i] = -rhs[i];
  }
  *pdelta = delta;
  return 1;
}

static _Thread_local const double* t_lam0 = NULL;      /* warm-start multipliers (CasADi sign) or NULL */
static void init_slacks(work_t* W, const lo_solver_opts* op0) {
  lo_int r; lo_solver_opts opl = *op0; const lo_solver_opts* op = &opl;
  const char* e_;
  if ((e_ = getenv("LAB_BPUSH"))) opl.bound_push = atof(e_);
  if ((e_ = getenv("LAB_BFRAC"))) opl.bound_frac = atof(e_);
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
      if ((e_ = getenv("LAB_ZINIT")) && atoi(e_)) { const double m0 = getenv("LAB_MUINIT") ? atof(getenv("LAB_MUINIT")) : op->mu_init;
        if (hL) zl = fmin(1.0, m0 / (sv - lb)); if (hU) zu = fmin(1.0, m0 / (ub - sv)); if (atoi(e_) == 2) { if (hL) zl = m0 / (sv - lb); if (hU) zu = m0 / (ub - sv); } }
    }
    if (t_lam0 && r >= 12) {
      const double fl = getenv("LAB_ZFLOOR") ? atof(getenv("LAB_ZFLOOR")) : 1e-3;
      if (lb == ub) { W->s[r] = sv; W->zL[r] = 0; W->zU[r] = 0; W->y[r] = t_lam0[r]; continue; }
      if (lb > -INFINITY) zl = fmax(-t_lam0[r], fl); if (ub < INFINITY) zu = fmax(t_lam0[r], fl);
    }
    W->s[r] = sv; W->zL[r] = zl; W->zU[r] = zu; W->y[r] = zu - zl;
  }
}


typedef struct { int thmin_rel, max_soc, sticky, stall, probe, trace, clip; double clip_tau, mu_init, tau_min, bpush, bfrac, mehro_lo, mehro_hi, piv_jump, piv_keep; int zinit, mehro, zcomp, crawl2; double thcap, thfloor, crawl2_frac; double stall_frac; int restart_period; double kappa_eps; int adapt; double sig_max, mono_fact; int adapt_glob; int clipk; double clipk_until; int full0; double full0_alpha; int clipkd; int fp32; int scaled; int stallany; } lab_t;
static lab_t LAB;
#define LAB_THFLOOR (getenv("LAB_THETAFLOOR") ? atof(getenv("LAB_THETAFLOOR")) * 1e-6 : 0.0)
static int lab_fp32(void) { return LAB.fp32; }
static void lab_init(void) {
  const char* e;
  memset(&LAB, 0, sizeof(LAB));
  if ((e = getenv("LAB_THMIN"))) LAB.thmin_rel = atoi(e);
  if ((e = getenv("LAB_SOC"))) LAB.max_soc = atoi(e);
  if ((e = getenv("LAB_STICKY"))) LAB.sticky = atoi(e);
  if ((e = getenv("LAB_STALL"))) LAB.stall = atoi(e);
  if ((e = getenv("LAB_PROBE"))) LAB.probe = atoi(e);
  if ((e = getenv("LAB_RESTART"))) LAB.restart_period = atoi(e);
  if ((e = getenv("LAB_KEPS"))) LAB.kappa_eps = atof(e);
  if ((e = getenv("LAB_CLIP"))) LAB.clip = atoi(e);
  LAB.clip_tau = 0.9; if ((e = getenv("LAB_CLIPTAU"))) LAB.clip_tau = atof(e);
  if ((e = getenv("LAB_MUINIT"))) LAB.mu_init = atof(e);
  if ((e = getenv("LAB_TAUMIN"))) LAB.tau_min = atof(e);
  if ((e = getenv("LAB_BPUSH"))) LAB.bpush = atof(e);
  if ((e = getenv("LAB_BFRAC"))) LAB.bfrac = atof(e);
  if ((e = getenv("LAB_ZINIT"))) LAB.zinit = atoi(e);
  if ((e = getenv("LAB_MEHRO"))) LAB.mehro = atoi(e);
  LAB.mehro_lo = -1e300; LAB.mehro_hi = 1e300;
  if ((e = getenv("LAB_MEHRO_LO"))) LAB.mehro_lo = atof(e);
  if ((e = getenv("LAB_MEHRO_HI"))) LAB.mehro_hi = atof(e);
  if ((e = getenv("LAB_PIVJUMP"))) LAB.piv_jump = atof(e);
  if ((e = getenv("LAB_PIVKEEP"))) LAB.piv_keep = atof(e);
  if ((e = getenv("LAB_ZCOMP"))) LAB.zcomp = atoi(e);
  if ((e = getenv("LAB_THCAP"))) LAB.thcap = atof(e);
  LAB.thfloor = 1e-3; if ((e = getenv("LAB_THFLOOR"))) LAB.thfloor = atof(e);
  if ((e = getenv("LAB_CRAWL2"))) LAB.crawl2 = atoi(e);
  LAB.crawl2_frac = 0.125; if ((e = getenv("LAB_CRAWL2_FRAC"))) LAB.crawl2_frac = atof(e);
  if ((e = getenv("LAB_CLIPK"))) LAB.clipk = atoi(e);
  LAB.clipk_until = 0.0; if ((e = getenv("LAB_CLIPK_UNTIL"))) LAB.clipk_until = atof(e);
  if ((e = getenv("LAB_SCALED"))) LAB.scaled = atoi(e);
  if ((e = getenv("LAB_STALLANY"))) LAB.stallany = atoi(e);
  if ((e = getenv("LAB_FP32"))) LAB.fp32 = atoi(e);
  if ((e = getenv("LAB_CLIPKD"))) LAB.clipkd = atoi(e);
  if ((e = getenv("LAB_FULL0"))) LAB.full0 = atoi(e);
  LAB.full0_alpha = 1.0; if ((e = getenv("LAB_FULL0_ALPHA"))) LAB.full0_alpha = atof(e);
  if ((e = getenv("LAB_ADAPT"))) LAB.adapt = atoi(e);
  LAB.sig_max = 100.0; if ((e = getenv("LAB_SIGMAX"))) LAB.sig_max = atof(e);
  LAB.mono_fact = 0.8; if ((e = getenv("LAB_MONOFACT"))) LAB.mono_fact = atof(e);
  LAB.adapt_glob = 1; if ((e = getenv("LAB_ADAPTGLOB"))) LAB.adapt_glob = atoi(e);
  LAB.trace = getenv("LO_TRACE") != NULL;
}

/* right-hand-side vectors for constraint residual cres (rows >= 12): rho, gamma_k, b_k */
static void build_vectors(const lo_form* F, const double* p, work_t* W, const double* cres, const double* rbar) {
  const int N = W->N; int k; lo_int r;
  for (r = 0; r < W->ng; ++r) W->rho[r] = (r >= 12 && W->lb[r] != W->ub[r]) ? rbar[r] + W->sig[r] * cres[r] : 0.0;
  for (k = 0; k < N; ++k) {
    const int nr = lo_stage_rows(F, k), g0 = 36 + 104 * k; int q, a;
    const double* J = W->Jst + (size_t)k * 104 * 60; double* mk = W->mvec + k * NW;
    memset(mk, 0, sizeof(double) * NW);
    for (q = 12; q < nr; ++q) { const double rh = W->rho[g0 + q]; for (a = 0; a < 60; ++a) { const int wa = loc2w(a); if (wa >= 0 && J[q * 60 + a] != 0.0) mk[wa] += rh * J[q * 60 + a]; } }
    for (q = 0; q < 12; ++q) W->bv[k * 12 + ROW2STATE[q]] = -cres[g0 + q];
    if (F->run_cost) { double gr[36]; for (a = 0; a < 36; ++a) gr[a] = 0.0; (void)lo_run_cost_stage(F, W->x, p, k, gr, gr + 12, gr + 24); for (a = 0; a < 36; ++a) mk[a] += gr[a]; }
  }
}

/* forward sweep from sig0: dx, ds (inequality rows), yn (dynamics rows) */
static void forward_sweep(const lo_form* F, work_t* W, const double* sig0, const double* cres) {
  const int N = W->N; int k; lo_int i; double sig[24], w[NW];
  memcpy(sig, sig0, sizeof(sig));
  for (k = 0; k < N; ++k) {
    const int last = (k == N - 1), nu = last ? 12 : 24, nr = lo_stage_rows(F, k), g0 = 36 + 104 * k; int q, a, t;
    const double* J = W->Jst + (size_t)k * 104 * 60; double signext[24];
    for (a = 0; a < 24; ++a) w[a] = sig[a];
    for (a = 0; a < nu; ++a) { double v = W->kap[k * 24 + a]; for (t = 0; t < 24; ++t) v += W->K[(size_t)k * 576 + a * 24 + t] * sig[t]; w[24 + a] = -v; }
    for (a = nu; a < 24; ++a) w[24 + a] = 0.0;
    for (a = 0; a < 12; ++a) { W->dx[12 * k + a] = w[a]; W->dx[12 * (N + 1) + 24 * k + a] = w[12 + a]; W->dx[12 * (N + 1) + 24 * k + 12 + a] = w[24 + a]; }
    for (q = 12; q < nr; ++q) {
      double v = 0; for (a = 0; a < 60; ++a) { const int wa = loc2w(a); if (wa >= 0 && J[q * 60 + a] != 0.0) v += J[q * 60 + a] * w[wa]; }
      W->ds[g0 + q] = v + cres[g0 + q];
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
    W->ds[ra] = sig[i] + cres[ra]; W->ds[rb] = sig[i] + cres[rb];
  }
}

/* dual steps + fraction-to-the-boundary bounds for the current ds; mu_c = centering parameter used in dz */
static lo_int g_block_row = -1;
static const double *g_muL = NULL, *g_muU = NULL;     /* per-row centering targets (corrector), NULL = scalar mu */
static _Thread_local int t_clip_now = 0, t_clipk_cur = -1;
static void dual_steps(work_t* W, double mu_s, double tau, double* a_pr, double* a_du) {
  lo_int r; double ap = 1.0, ad = 1.0; double small[64]; int ns = 0, kk = (t_clipk_cur >= 0 ? t_clipk_cur : LAB.clipk) > 64 ? 64 : (t_clipk_cur >= 0 ? t_clipk_cur : LAB.clipk); g_block_row = -1;
  double smalld[64]; int nsd = 0, kd = LAB.clipkd > 64 ? 64 : LAB.clipkd;
#define PUSH_RATIOD(v) do { if (kd > 0) { double v_ = (v); int q_; if (nsd < kd) { smalld[nsd++] = v_; } else { int im = 0; for (q_ = 1; q_ < kd; ++q_) if (smalld[q_] > smalld[im]) im = q_; if (v_ < smalld[im]) smalld[im] = v_; } } } while (0)
#define PUSH_RATIO(v) do { if (kk > 0) { double v_ = (v); int q_; if (ns < kk) { small[ns++] = v_; } else { int im = 0; for (q_ = 1; q_ < kk; ++q_) if (small[q_] > small[im]) im = q_; if (v_ < small[im]) small[im] = v_; } } } while (0)
  for (r = 12; r < W->ng; ++r) {
    const double lb = W->lb[r], ub = W->ub[r]; double s, ds, yn;
    if (lb == ub) continue;
    s = W->s[r]; ds = W->ds[r]; yn = W->sig[r] * ds;
    if (lb > -INFINITY) {
      const double mu_c = g_muL ? g_muL[r] : mu_s;
      const double d = s - lb, zl = W->zL[r], dz = mu_c / d - zl - zl / d * ds;
      W->dzL[r] = dz; yn -= mu_c / d;
      if (ds < 0.0) PUSH_RATIO(-tau * d / ds);
      if (ds < 0.0 && -tau * d / ds < ap) { ap = -tau * d / ds; g_block_row = r; }
      if (dz < 0.0) { ad = fmin(ad, -tau * zl / dz); PUSH_RATIOD(-tau * zl / dz); }
    } else W->dzL[r] = 0.0;
    if (ub < INFINITY) {
      const double mu_c = g_muU ? g_muU[r] : mu_s;
      const double d = ub - s, zu = W->zU[r], dz = mu_c / d - zu + zu / d * ds;
      W->dzU[r] = dz; yn += mu_c / d;
      if (ds > 0.0) PUSH_RATIO(tau * d / ds);
      if (ds > 0.0 && tau * d / ds < ap) { ap = tau * d / ds; g_block_row = r; }
      if (dz < 0.0) { ad = fmin(ad, -tau * zu / dz); PUSH_RATIOD(-tau * zu / dz); }
    } else W->dzU[r] = 0.0;
    W->yn[r] = yn;
  }
  if (kd > 0 && t_clip_now && nsd == kd) { int q_, im = 0; for (q_ = 1; q_ < kd; ++q_) if (smalld[q_] > smalld[im]) im = q_; ad = fmin(1.0, smalld[im]); }
  if (kk > 0 && t_clip_now && ns == kk) { int q_, im = 0; for (q_ = 1; q_ < kk; ++q_) if (small[q_] > small[im]) im = q_; ap = fmin(1.0, small[im]); }
  *a_pr = ap; *a_du = ad;
}

/* slack of row r at step alpha; LAB.clip: the slack stops at (1 - tau) of its current distance to the bound instead of
 * limiting the step of every other variable (componentwise fraction-to-the-boundary rule) */
static double slack_at(const work_t* W, lo_int r, double alpha) {
  double s = W->s[r] + alpha * W->ds[r];
  if (LAB.clip || t_clip_now) {
    const double lb = W->lb[r], ub = W->ub[r], t = 1.0 - LAB.clip_tau;
    if (lb > -INFINITY) s = fmax(s, lb + t * (W->s[r] - lb));
    if (ub < INFINITY) s = fmin(s, ub - t * (ub - W->s[r]));
  }
  return s;
}
/* theta and barrier objective at the trial point x + alpha dx, s + alpha ds (gt receives g) */
static void trial_point(const lo_form* F, const double* p, work_t* W, const lo_poff* o, double alpha, double mu, double* tht, double* pht) {
  const int N = W->N; lo_int i, r; int k; double th = 0, bt = 0, ft = 0;
  for (i = 0; i < W->nx; ++i) W->xt[i] = W->x[i] + alpha * W->dx[i];
  eval_g(F, W->xt, p, W->gt);
  for (r = 12; r < W->ng; ++r) {
    const double lb = W->lb[r], ub = W->ub[r], g = W->gt[r]; double s;
    if (lb == ub) { th += fabs(g - lb); continue; }
    s = slack_at(W, r, alpha); th += fabs(g - s);
    if (lb > -INFINITY) bt -= log(s - lb);
    if (ub < INFINITY) bt -= log(ub - s);
  }
  for (i = 0; i < 12; ++i) { const double d = W->xt[12 * N + i] - p[12 * N + i]; ft += p[o->QN + i] * d * d; }
  if (F->run_cost) for (k = 0; k < N; ++k) ft += lo_run_cost_stage(F, W->xt, p, k, NULL, NULL, NULL);
  *tht = th; *pht = ft + mu * bt;
}

/* one NLP; returns status (0 converged, 1 max_iter, 2 numerical) */
static int solve_one(const lo_form* F, const double* p, const double* x0, const lo_solver_opts* op, double* x_out,
                     double* lam_out, int* iters_out, double kkt_out[3], long long counters[5]) {
  const int N = F->N; const lo_int nx = lo_nx(N), ng = lo_ng(N);
  lo_poff o; work_t Wk, *W = &Wk; lo_int i, r; int k, it, status = 1, nfilt = 0, streak = 0, nreset = 0, last_reset_it = 0, ncrawl = 0;
  double mu = LAB.mu_init > 0 ? LAB.mu_init : op->mu_init, delta_last = 0.0, th_max = 0.0, e_du = 0.0;
  double filt_th[64], filt_ph[64], th_min = 1e-4, delta_used = 0.0, minpiv_last = 1e300;
  double *gx, *cres, *rbar, *csoc, *dx0, *ds0, *yn0, *dzL0, *dzU0, *rbar2, *muL, *muU; long long ncorr = 0;
  int last_mu_it = 0, first_mu_it = -1;
  int afree = 1, nref = 0; double refs[4], mu_max_ad = -1.0; long long nfixed = 0;
  reg_t RG; long long nsoc_total = 0, soc_acc = 0; int cutstreak = 0; double thhist[32]; int nth = 0;
  const double keps = LAB.kappa_eps > 0 ? LAB.kappa_eps : op->kappa_eps;
  const int rperiod = LAB.restart_period > 0 ? LAB.restart_period : op->restart_period;
  memset(&RG, 0, sizeof(RG)); RG.fail_stage = 99; t_clipk_cur = -1;
  lo_param_offsets(N, &o);
  W->N = N; W->nx = nx; W->ng = ng;
  W->x = dalloc(nx); W->xt = dalloc(nx); W->dx = dalloc(nx); gx = dalloc(nx);
  rbar2 = dalloc(ng); muL = dalloc(ng); muU = dalloc(ng);
  cres = dalloc(ng); rbar = dalloc(ng); csoc = dalloc(ng); dx0 = dalloc(nx); ds0 = dalloc(ng); yn0 = dalloc(ng); dzL0 = dalloc(ng); dzU0 = dalloc(ng);
  W->g = dalloc(ng); W->gt = dalloc(ng); W->s = dalloc(ng); W->ds = dalloc(ng); W->zL = dalloc(ng); W->zU = dalloc(ng);
  W->dzL = dalloc(ng); W->dzU = dalloc(ng); W->y = dalloc(ng); W->yn = dalloc(ng); W->lb = dalloc(ng); W->ub = dalloc(ng);
  W->sig = dalloc(ng); W->rho = dalloc(ng);
  W->Jst = dalloc((size_t)N * 104 * 60); W->Hst = dalloc((size_t)N * 3600);
  W->M = dalloc((size_t)N * NW * NW); W->mvec = dalloc((size_t)N * NW); W->Ah = dalloc((size_t)N * 432); W->bv = dalloc((size_t)N * 12);
  W->K = dalloc((size_t)N * 576); W->kap = dalloc((size_t)N * 24); W->Px = dalloc((size_t)(N + 1) * 288); W->pvx = dalloc((size_t)(N + 1) * 12);
  memcpy(W->x, x0, sizeof(double) * nx);
  for (i = 0; i < 6; ++i) { W->x[i] = p[o.q_init + i]; W->x[6 + i] = p[o.qd_init + i]; }
  if (getenv("LAB_F0")) {   /* gravity-compensating forces where the guess has none */
    const double fz = atof(getenv("LAB_F0")) * p[o.mass] * 9.81 / 4.0; int l;
    for (k = 0; k < N; ++k) for (l = 0; l < 4; ++l) { double* f = W->x + 12 * (N + 1) + 24 * k + 12 + 3 * l; if (f[0] == 0.0 && f[1] == 0.0 && f[2] == 0.0) f[2] = fz; }
  }
  lo_bounds(F, p, W->lb, W->ub);
  eval_g(F, W->x, p, W->g);
  t_lam0 = (getenv("LAB_LAMWS") && lam_out) ? lam_out : NULL;
  init_slacks(W, op);
  t_lam0 = NULL;
  for (it = 0; it <= op->max_iter; ++it) {
    double du = 0, pr = 0, co = 0, tau, delta;
    int fact_ok = 0, attempt;
    double sig[24], a_pr = 1.0, a_du = 1.0, th0 = 0, bar = 0, dphi = 0, f0 = 0, ph0, alpha;
    int accepted = 0, armijo = 0;
    /* derivatives per stage + gx = grad f + J^T y */
    memset(gx, 0, sizeof(double) * nx);
    for (i = 0; i < 12; ++i) {
      gx[12 * N + i] = 2.0 * p[o.QN + i] * (W->x[12 * N + i] - p[12 * N + i]) + (i < 6 ? W->y[12 + i] + W->y[18 + i] : W->y[24 + i - 6] + W->y[30 + i - 6]);
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
        for (c = 0; c < 12; ++c) Hs[c * 60 + c] += 2.0 * dtk * F->QX[c];
        for (l2 = 0; l2 < 4; ++l2) for (a = 0; a < 3; ++a) {
          const int ic = 12 + 3 * l2 + a, jf = 24 + 3 * l2 + a; const double hc = 2.0 * dtk * F->Qc[a];
          Hs[a * 60 + a] += hc; Hs[ic * 60 + ic] += hc; Hs[a * 60 + ic] -= hc; Hs[ic * 60 + a] -= hc;
          Hs[jf * 60 + jf] += 2.0 * dtk * F->Qf[a];
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
      if (lb > -INFINITY) co = fmax(co, (W->s[r] - lb) * W->zL[r]);
      if (ub < INFINITY) co = fmax(co, (ub - W->s[r]) * W->zU[r]);
    }
    e_du = du;
    if (getenv("LO_TRACE")) fprintf(stderr, "it %4d pr %9.2e du %9.2e co %9.2e mu %8.1e dlast %8.1e nreset %d nfilt %d\n", it, pr, du, co, mu, delta_last, nreset, nfilt);
    if (!(du < 1e300) || !(pr < 1e300) || !(co < 1e300)) { status = 2; break; }
    if (fmax(du, fmax(pr, co)) <= op->tol) { status = 0; break; }
    if (it == op->max_iter) break;
    if (du > op->reset_du && nreset >= op->max_resets && op->max_resets > 0) { status = 2; break; }
    {
      int stalled = rperiod > 0 && it - last_reset_it >= rperiod && mu >= (LAB.mu_init > 0 ? LAB.mu_init : op->mu_init) && nreset < op->max_resets && ncrawl < (getenv("LAB_CRAWLMAX") ? atoi(getenv("LAB_CRAWLMAX")) : 1);
      if (stalled && LAB.stall == 1 && pr <= 1e-2 && du <= 1e2 * keps * mu) stalled = 0;   /* the barrier problem is about to be solved */
      if (LAB.stall == 2) {   /* progress-based crawl test: theta must have dropped by the factor stall_frac over the last `win` iterations */
        const int win = getenv("LAB_WIN") ? atoi(getenv("LAB_WIN")) : 20; const double fr = getenv("LAB_FRAC") ? atof(getenv("LAB_FRAC")) : 0.5;
        const int t0 = getenv("LAB_T0") ? atoi(getenv("LAB_T0")) : 40;
        stalled = 0;
        if (it - last_reset_it >= t0 && nth >= win + 1 && mu >= (LAB.mu_init > 0 ? LAB.mu_init : op->mu_init) && nreset < op->max_resets && ncrawl < 1 &&
            thhist[(nth - 1) & 31] > fr * thhist[(nth - 1 - win) & 31]) stalled = 1;
        if (it - last_reset_it >= 100 && mu >= (LAB.mu_init > 0 ? LAB.mu_init : op->mu_init) && nreset < op->max_resets && ncrawl < 1) stalled = 1;
      }
      if (LAB.stallany > 0 && mu < (LAB.mu_init > 0 ? LAB.mu_init : op->mu_init) && it - last_mu_it >= LAB.stallany && it - last_reset_it >= LAB.stallany && nreset < op->max_resets) stalled = 1;
      if (LAB.crawl2 > 0 && cutstreak >= LAB.crawl2 && nreset < op->max_resets && ncrawl < 1 && it - last_reset_it >= 20) stalled = 1;
      if (stalled) ncrawl++;
      if (!((du > op->reset_du && nreset < op->max_resets) || stalled || (op->reset_delta > 0.0 && delta_last > op->reset_delta && nreset < op->max_resets))) goto no_reset;
      last_reset_it = it;
      cutstreak = 0; nth = 0;
      nreset++;
      if ((getenv("LAB_FRESH") && nreset == atoi(getenv("LAB_FRESH"))) || (getenv("LAB_FRESHJAM") && !stalled && nreset <= atoi(getenv("LAB_FRESHJAM")))) {   /* n-th restart: back to the caller's initial guess with another step rule */
        memcpy(W->x, x0, sizeof(double) * nx);
        for (i = 0; i < 6; ++i) { W->x[i] = p[o.q_init + i]; W->x[6 + i] = p[o.qd_init + i]; }
        eval_g(F, W->x, p, W->g);
        t_clipk_cur = getenv("LAB_FRESHK") ? atoi(getenv("LAB_FRESHK")) : 2;
        th_max = 0.0;
      }
      init_slacks(W, op); mu = LAB.mu_init > 0 ? LAB.mu_init : op->mu_init; nfilt = 0; delta_last = 0.0; streak = 0;
      if (getenv("LAB_ALTCLIP")) { static const int seq[3] = {4, 2, 1}; t_clipk_cur = seq[nreset % 3]; }   /* another step rule after every restart */
      continue;
    }
    no_reset:;
    if (LAB.adapt) {   /* IPOPT's adaptive barrier strategy, kkt-error globalisation (IpAdaptiveMuUpdate.cpp) */
      double du1 = 0, pr1 = 0, co1 = 0, E; long long nc = 0; int suff = (nref == 0), e2;
      for (i = 12; i < nx; ++i) du1 += fabs(gx[i]);
      for (r = 12; r < ng; ++r) {
        const double lb = W->lb[r], ub = W->ub[r];
        if (lb == ub) { pr1 += fabs(W->g[r] - lb); continue; }
        pr1 += fabs(W->g[r] - W->s[r]);
        if (lb > -INFINITY) { co1 += (W->s[r] - lb) * W->zL[r]; nc++; }
        if (ub < INFINITY) { co1 += (ub - W->s[r]) * W->zU[r]; nc++; }
      }
      E = du1 / (double)(nx - 12) + pr1 / (double)(ng - 12) + co1 / (double)nc;
      for (e2 = 0; e2 < nref; ++e2) if (E <= 0.9999 * refs[e2]) suff = 1;
      if (!LAB.adapt_glob) suff = 1;
      if (suff) {
        if (!afree) { afree = 1; }
        if (nref == 4) { memmove(refs, refs + 1, 3 * sizeof(double)); nref = 3; }
        refs[nref++] = E;
      } else if (afree) {
        afree = 0; mu = fmax(op->tol / 10.0, LAB.mono_fact * co1 / (double)nc); nfilt = 0;
      }
      if (!afree) nfixed++;
    }
    if (!LAB.adapt || !afree) for (;;) {
      double cm = 0;
      for (r = 12; r < ng; ++r) {
        const double lb = W->lb[r], ub = W->ub[r];
        if (lb == ub) continue;
        if (lb > -INFINITY) cm = fmax(cm, fabs((W->s[r] - lb) * W->zL[r] - mu));
        if (ub < INFINITY) cm = fmax(cm, fabs((ub - W->s[r]) * W->zU[r] - mu));
      }
      double sd = 1.0, sc = 1.0;
      if (LAB.scaled) {   /* IPOPT's scaling of the optimality error (eq. 5 of Waechter & Biegler): s_d = max(s_max, (|y|_1 + |z|_1) / (m + n)) / s_max, s_c likewise */
        double ys = 0, zs = 0; long long nz = 0;
        for (r = 12; r < ng; ++r) { ys += fabs(W->y[r]); if (W->lb[r] != W->ub[r]) { if (W->lb[r] > -INFINITY) { zs += W->zL[r]; nz++; } if (W->ub[r] < INFINITY) { zs += W->zU[r]; nz++; } } }
        sd = fmax(100.0, (ys + zs) / (double)(ng - 12 + nz)) / 100.0; sc = fmax(100.0, zs / (double)nz) / 100.0;
        if (LAB.scaled == 2) { sd = fmax(1.0, (ys + zs) / (double)(ng - 12 + nz)); sc = fmax(1.0, zs / (double)nz); }
      }
      if (fmax(du / sd, fmax(pr, cm / sc)) <= keps * mu && mu > op->tol / 10.0) { mu = fmax(op->tol / 10.0, fmin(op->kappa_mu * mu, pow(mu, op->theta_mu))); nfilt = 0; last_mu_it = it; if (first_mu_it < 0) first_mu_it = it; }
      else break;
    }
    tau = fmax(LAB.tau_min > 0 ? LAB.tau_min : op->tau_min, 1.0 - mu);
    for (r = 0; r < ng; ++r) {
      const double lb = W->lb[r], ub = W->ub[r]; double sg = 0, rh = 0, cr = 0;
      if (r >= 12) {
        if (lb != ub) {
          const double s = W->s[r];
          if (lb > -INFINITY) { const double d = s - lb; sg += W->zL[r] / d; rh -= mu / d; }
          if (ub < INFINITY) { const double d = ub - s; sg += W->zU[r] / d; rh += mu / d; }
          cr = W->g[r] - s;
        } else cr = W->g[r] - lb;
      }
      W->sig[r] = sg; rbar[r] = rh; cres[r] = cr;
    }
    /* condensation per stage (matrices): M = H + Jd^T Sigma Jd (48x48), A^ = -J_dyn (state order) */
    for (k = 0; k < N; ++k) {
      const int nr = lo_stage_rows(F, k), g0 = 36 + 104 * k; int q, a, b;
      const double* J = W->Jst + (size_t)k * 104 * 60; const double* H = W->Hst + (size_t)k * 3600;
      double* Mk = W->M + (size_t)k * NW * NW; double* Ah = W->Ah + (size_t)k * 432;
      memset(Ah, 0, sizeof(double) * 432);
      for (a = 0; a < 60; ++a) { const int wa = loc2w(a); if (wa < 0) continue; for (b = 0; b < 60; ++b) { const int wb = loc2w(b); if (wb >= 0) Mk[wa * NW + wb] = H[a * 60 + b]; } }
      for (q = 12; q < nr; ++q) {
        int idx[16], n = 0; double val[16]; const double sg = W->sig[g0 + q];
        for (a = 0; a < 60; ++a) if (J[q * 60 + a] != 0.0 && loc2w(a) >= 0) { idx[n] = loc2w(a); val[n] = J[q * 60 + a]; ++n; }
        for (a = 0; a < n; ++a) for (b = 0; b < n; ++b) Mk[idx[a] * NW + idx[b]] += sg * val[a] * val[b];
      }
      for (q = 0; q < 12; ++q) for (a = 0; a < 36; ++a) Ah[ROW2STATE[q] * 36 + a] = -J[q * 60 + a];
    }
    build_vectors(F, p, W, cres, rbar);
    /* factorisation with inertia correction */
    delta = (streak >= (getenv("LAB_STREAKMIN") ? atoi(getenv("LAB_STREAKMIN")) : 2) && delta_last > 0.0) ? fmax(1e-20, delta_last * op->delta_dec) : 0.0;
    if (LAB.piv_keep > 0 && delta_last > 0.0 && streak >= 1) {   /* below delta_last - minpiv the last matrix was certainly indefinite */
      const double lbd = delta_last - LAB.piv_keep * minpiv_last;
      if (lbd > delta) delta = lbd;
    }
    RG.sticky = LAB.sticky; RG.delta_last = delta_last; RG.delta_init = op->delta_init; RG.inc_first = op->delta_inc_first; RG.inc = op->delta_inc; RG.dec = op->delta_dec;
    for (attempt = 0; attempt < 60 && !fact_ok; ++attempt) {
      if (attempt > 0) {
        const double dn = next_delta(delta, &RG);
        delta = (LAB.piv_jump > 0 && t_failpiv <= 0.0 && delta + LAB.piv_jump * -t_failpiv > dn) ? delta + LAB.piv_jump * -t_failpiv : dn;
        if (delta > 1e40) break;
      }
      t_minpiv = 1e300; t_failpiv = 0.0;
      counters[0]++;
      { double d_io = delta; fact_ok = riccati_backward(F, p, W, &d_io, &o, sig, &RG); if (fact_ok) { delta_used = delta; delta = d_io; } }
    }
    if (!fact_ok) { status = 2; break; }
    if (getenv("LAB_ATT")) fprintf(stderr, "ATT %d streak %d dlast %.3e first %.3e final %.3e attempts %d\n", it, streak, delta_last, (streak >= 2 && delta_last > 0.0) ? fmax(1e-20, delta_last * op->delta_dec) : 0.0, delta, attempt);
    minpiv_last = t_minpiv;
    if (delta > 0.0) { delta_last = delta; streak++; } else streak = 0;
    if (streak > 8) streak = 0;
    if (LAB.adapt && afree) {
      double avg = 0, minc = 1e300, sigma, d_io = delta_used, sg2[24]; long long nc = 0;
      for (r = 12; r < ng; ++r) {
        const double lb = W->lb[r], ub = W->ub[r];
        if (lb == ub) continue;
        if (lb > -INFINITY) { const double c = (W->s[r] - lb) * W->zL[r]; avg += c; if (c < minc) minc = c; nc++; }
        if (ub < INFINITY) { const double c = (ub - W->s[r]) * W->zU[r]; avg += c; if (c < minc) minc = c; nc++; }
      }
      avg /= (double)nc;
      if (mu_max_ad < 0.0) mu_max_ad = 1e3 * avg;
      if (LAB.adapt == 1) {   /* LOQO rule */
        const double xi = minc / avg; sigma = 0.1 * pow(fmin(0.05 * (1.0 - xi) / xi, 2.0), 3.0);
      } else {                /* Mehrotra probing: affine-scaling step with the factorisation at hand */
        double ap, ad, maff = 0;
        for (r = 0; r < ng; ++r) rbar2[r] = 0.0;
        build_vectors(F, p, W, cres, rbar2);
        RG.sticky = 1;
        if (!riccati_backward(F, p, W, &d_io, &o, sg2, &RG)) { status = 2; break; }
        RG.sticky = LAB.sticky; ncorr++;
        forward_sweep(F, W, sg2, cres);
        dual_steps(W, 0.0, 1.0, &ap, &ad);
        for (r = 12; r < ng; ++r) {
          const double lb = W->lb[r], ub = W->ub[r];
          if (lb == ub) continue;
          if (lb > -INFINITY) maff += (W->s[r] + ap * W->ds[r] - lb) * (W->zL[r] + ad * W->dzL[r]);
          if (ub < INFINITY) maff += (ub - W->s[r] - ap * W->ds[r]) * (W->zU[r] + ad * W->dzU[r]);
        }
        maff /= (double)nc;
        sigma = fmin(pow(maff / avg, 3.0), LAB.sig_max);
      }
      mu = fmin(fmax(sigma * avg, op->tol / 10.0), mu_max_ad);
      tau = fmax(LAB.tau_min > 0 ? LAB.tau_min : op->tau_min, 1.0 - mu);
      for (r = 12; r < ng; ++r) {
        const double lb = W->lb[r], ub = W->ub[r]; double rh = 0;
        if (lb != ub) { if (lb > -INFINITY) rh -= mu / (W->s[r] - lb); if (ub < INFINITY) rh += mu / (ub - W->s[r]); }
        rbar[r] = rh;
      }
      build_vectors(F, p, W, cres, rbar);
      RG.sticky = 1; d_io = delta_used;
      if (!riccati_backward(F, p, W, &d_io, &o, sig, &RG)) { status = 2; break; }
      RG.sticky = LAB.sticky; ncorr++;
      nfilt = 0;
      if (LAB.trace) fprintf(stderr, "      adaptive: avg %9.2e sigma %9.2e mu %9.2e\n", avg, sigma, mu);
    }
    forward_sweep(F, W, sig, cres);
    if (it < LAB.full0) {   /* warm-up: full Newton step in x, slacks / multipliers re-initialised at the new point */
      for (i = 0; i < nx; ++i) W->x[i] += LAB.full0_alpha * W->dx[i];
      eval_g(F, W->x, p, W->g);
      for (r = 12; r < ng; ++r) if (W->lb[r] == W->ub[r]) W->y[r] += LAB.full0_alpha * (W->yn[r] - W->y[r]);
      init_slacks(W, op); nfilt = 0; delta_last = 0.0; streak = 0;
      continue;
    }
    /* dual steps, step bounds, merit data */
    t_clip_now = (t_clipk_cur >= 0 ? t_clipk_cur : LAB.clipk) > 1 && pr > LAB.clipk_until;
    dual_steps(W, mu, tau, &a_pr, &a_du);
    if (LAB.mehro && (LAB.mehro == 1 || a_pr < 0.5)) {   /* corrector: second-order complementarity term of the predictor step */
      double sg2[24]; double d_io = delta_used; const double sc = (LAB.mehro == 3) ? a_pr * a_du : 1.0;
      for (r = 12; r < ng; ++r) {
        const double lb = W->lb[r], ub = W->ub[r]; double rb = 0.0;
        muL[r] = mu; muU[r] = mu;
        if (lb == ub) { rbar2[r] = 0.0; continue; }
        if (lb > -INFINITY) { const double d = W->s[r] - lb; muL[r] = fmin(fmax(mu - sc * W->ds[r] * W->dzL[r], LAB.mehro_lo * mu), LAB.mehro_hi * mu); rb -= muL[r] / d; }
        if (ub < INFINITY) { const double d = ub - W->s[r]; muU[r] = fmin(fmax(mu + sc * W->ds[r] * W->dzU[r], LAB.mehro_lo * mu), LAB.mehro_hi * mu); rb += muU[r] / d; }
        rbar2[r] = rb;
      }
      build_vectors(F, p, W, cres, rbar2);
      RG.sticky = 1;
      if (riccati_backward(F, p, W, &d_io, &o, sg2, &RG)) {
        forward_sweep(F, W, sg2, cres);
        g_muL = muL; g_muU = muU;
        dual_steps(W, mu, tau, &a_pr, &a_du);
        g_muL = g_muU = NULL;
        ncorr++;
      }
      RG.sticky = LAB.sticky;
    }
    for (r = 12; r < ng; ++r) {
      const double lb = W->lb[r], ub = W->ub[r], g = W->g[r]; double s, ds;
      if (lb == ub) { th0 += fabs(g - lb); continue; }
      s = W->s[r]; ds = W->ds[r]; th0 += fabs(g - s);
      if (lb > -INFINITY) { const double d = s - lb; bar -= log(d); dphi -= mu * ds / d; }
      if (ub < INFINITY) { const double d = ub - s; bar -= log(d); dphi += mu * ds / d; }
    }
    for (i = 0; i < 12; ++i) { const double d = W->x[12 * N + i] - p[12 * N + i], qn = p[o.QN + i]; f0 += qn * d * d; dphi += 2.0 * qn * d * W->dx[12 * N + i]; }
    if (F->run_cost) for (k = 0; k < N; ++k) {
      double gX[12] = {0}, gc[12] = {0}, gf[12] = {0}; const double* dX = W->dx + 12 * k; const double* dU = W->dx + 12 * (N + 1) + 24 * k;
      f0 += lo_run_cost_stage(F, W->x, p, k, gX, gc, gf);
      for (i = 0; i < 12; ++i) dphi += gX[i] * dX[i] + gc[i] * dU[i] + gf[i] * dU[12 + i];
    }
    ph0 = f0 + mu * bar;
    if (th_max == 0.0) { th_max = 1e4 * fmax(1.0, th0); th_min = LAB.thmin_rel ? 1e-4 * fmax(1.0, th0) : 1e-4; }
    alpha = LAB.clip ? 1.0 : a_pr;
    {
      int first = 1, nsoc_done = 0;
      while (alpha > 1e-10) {
        double tht, pht; int ok_f, e, switching;
        counters[1]++;
        trial_point(F, p, W, &o, alpha, mu, &tht, &pht);
        ok_f = (tht <= th_max) && (pht < 1e300) && (pht > -1e300) && (tht < 1e300);
        if (LAB.thcap > 0 && tht > fmax(LAB.thcap * th0, LAB.thfloor)) ok_f = 0;
        for (e = 0; e < nfilt && ok_f; ++e) if (tht >= fmax(filt_th[e], LAB_THFLOOR) && pht >= filt_ph[e]) ok_f = 0;
        switching = (dphi < 0.0) && (th0 <= th_min) && (alpha * pow(-dphi, 2.3) > pow(th0, 1.1));
        if (ok_f) {
          if (switching) { if (pht <= ph0 + 1e-8 * alpha * dphi) { accepted = 1; armijo = 1; } }
          else if (tht <= fmax((1.0 - 1e-5) * th0, LAB_THFLOOR) || pht <= ph0 - 1e-8 * th0) accepted = 1;
        }
        if (accepted) break;
        /* second-order correction (Waechter & Biegler, sec. 2.4): only at the first trial point, only when theta went up */
        if (first && LAB.max_soc > 0 && tht >= th0) {
          double th_prev = tht, a_soc = alpha, a_du_soc = a_du; int ps;
          memcpy(dx0, W->dx, sizeof(double) * nx); memcpy(ds0, W->ds, sizeof(double) * ng); memcpy(yn0, W->yn, sizeof(double) * ng);
          memcpy(dzL0, W->dzL, sizeof(double) * ng); memcpy(dzU0, W->dzU, sizeof(double) * ng);
          for (r = 0; r < ng; ++r) csoc[r] = cres[r];
          for (ps = 0; ps < LAB.max_soc && !accepted; ++ps) {
            double tht2, pht2, sg2[24]; double d_io = delta_used; int okf2;
            /* c_soc = alpha_soc * c_soc + c(trial) */
            for (r = 12; r < ng; ++r) {
              const double lb = W->lb[r], ub = W->ub[r];
              const double ct = (lb == ub) ? W->gt[r] - lb : W->gt[r] - slack_at(W, r, a_soc);
              csoc[r] = a_soc * csoc[r] + ct;
            }
            build_vectors(F, p, W, csoc, rbar);
            RG.sticky = 1;     /* same matrix: the regularisation that worked is reused (sticky retries should not trigger) */
            if (!riccati_backward(F, p, W, &d_io, &o, sg2, &RG)) break;
            nsoc_total++; nsoc_done++;
            forward_sweep(F, W, sg2, csoc);
            dual_steps(W, mu, tau, &a_soc, &a_du_soc);
            counters[1]++;
            trial_point(F, p, W, &o, a_soc, mu, &tht2, &pht2);
            okf2 = (tht2 <= th_max) && (pht2 < 1e300) && (pht2 > -1e300) && (tht2 < 1e300);
            for (e = 0; e < nfilt && okf2; ++e) if (tht2 >= fmax(filt_th[e], LAB_THFLOOR) && pht2 >= filt_ph[e]) okf2 = 0;
            if (okf2) {
              if (switching) { if (pht2 <= ph0 + 1e-8 * alpha * dphi) { accepted = 1; armijo = 1; } }
              else if (tht2 <= fmax((1.0 - 1e-5) * th0, LAB_THFLOOR) || pht2 <= ph0 - 1e-8 * th0) accepted = 1;
            }
            if (accepted) { alpha = a_soc; a_du = a_du_soc; soc_acc++; break; }
            if (tht2 > 0.99 * th_prev) break;
            th_prev = tht2;
          }
          if (!accepted) {   /* back to the original step */
            memcpy(W->dx, dx0, sizeof(double) * nx); memcpy(W->ds, ds0, sizeof(double) * ng); memcpy(W->yn, yn0, sizeof(double) * ng);
            memcpy(W->dzL, dzL0, sizeof(double) * ng); memcpy(W->dzU, dzU0, sizeof(double) * ng);
            build_vectors(F, p, W, cres, rbar);
          }
          if (accepted) break;
        }
        first = 0;
        alpha *= 0.5;
      }
      (void)nsoc_done;
    }
    if (!accepted) {
      double tht, pht;
      nfilt = 0; alpha = fmin(a_pr, op->alpha_fallback);
      if (LAB.clip) alpha = op->alpha_fallback;
      trial_point(F, p, W, &o, alpha, mu, &tht, &pht);
    } else if (!armijo) {
      if (nfilt == 64) { memmove(filt_th, filt_th + 1, 63 * sizeof(double)); memmove(filt_ph, filt_ph + 1, 63 * sizeof(double)); nfilt = 63; }
      filt_th[nfilt] = (1.0 - 1e-5) * th0; filt_ph[nfilt] = ph0 - 1e-8 * th0; nfilt++;
    }
    if (accepted && alpha <= LAB.crawl2_frac * a_pr) cutstreak++; else cutstreak = 0;
    thhist[nth & 31] = th0; nth++;
    if (LAB.trace) fprintf(stderr, "      alpha %9.2e a_pr %9.2e a_du %9.2e delta %8.1e acc %d armijo %d th0 %9.2e dphi %9.2e failstage %d block row %d (stage %d type %d) s-dist %g\n", alpha, a_pr, a_du, delta, accepted, armijo, th0, dphi, RG.fail_stage, (int)g_block_row, g_block_row >= 36 ? (int)((g_block_row - 36) / 104) : -1, g_block_row >= 36 ? (int)((g_block_row - 36) % 104) : (int)g_block_row, g_block_row >= 0 ? fmin(W->s[g_block_row] - W->lb[g_block_row], W->ub[g_block_row] - W->s[g_block_row]) : 0.0);
    RG.fail_stage = 99;
    if (getenv("LAB_DUALCAP")) a_du = fmin(a_du, atof(getenv("LAB_DUALCAP")) * alpha);
    memcpy(W->x, W->xt, sizeof(double) * nx);
    for (r = 0; r < ng; ++r) {
      const double lb = W->lb[r], ub = W->ub[r]; double s, zl = 0, zu = 0;
      W->g[r] = W->gt[r];
      if (r < 12) continue;
      if (lb == ub) { W->y[r] += alpha * (W->yn[r] - W->y[r]); continue; }
      s = slack_at(W, r, alpha);
      if (LAB.zcomp) {   /* componentwise dual step: full Newton step per multiplier, each with its own fraction-to-the-boundary clip */
        const double az = LAB.zcomp == 2 ? alpha : 1.0;
        if (lb > -INFINITY) { const double d = s - lb; zl = fmax(W->zL[r] + az * W->dzL[r], (1.0 - tau) * W->zL[r]); zl = fmin(fmax(zl, mu / (1e10 * d)), 1e10 * mu / d); }
        if (ub < INFINITY) { const double d = ub - s; zu = fmax(W->zU[r] + az * W->dzU[r], (1.0 - tau) * W->zU[r]); zu = fmin(fmax(zu, mu / (1e10 * d)), 1e10 * mu / d); }
        W->s[r] = s; W->zL[r] = zl; W->zU[r] = zu; W->y[r] = zu - zl;
        continue;
      }
      if (lb > -INFINITY) { const double d = s - lb; zl = W->zL[r] + a_du * W->dzL[r]; if (LAB.clipkd && t_clip_now) zl = fmax(zl, (1.0 - tau) * W->zL[r]); zl = fmin(fmax(zl, mu / (1e10 * d)), 1e10 * mu / d); }
      if (ub < INFINITY) { const double d = ub - s; zu = W->zU[r] + a_du * W->dzU[r]; if (LAB.clipkd && t_clip_now) zu = fmax(zu, (1.0 - tau) * W->zU[r]); zu = fmin(fmax(zu, mu / (1e10 * d)), 1e10 * mu / d); }
      W->s[r] = s; W->zL[r] = zl; W->zU[r] = zu; W->y[r] = zu - zl;
    }
  }
  counters[2] += RG.nstage; counters[3] += nsoc_total + ncorr; counters[4] += soc_acc;
  for (i = 0; i < 12; ++i) W->y[i] = -gx[i];
  memcpy(x_out, W->x, sizeof(double) * nx);
  if (lam_out) memcpy(lam_out, W->y, sizeof(double) * ng);
  if (iters_out) *iters_out = getenv("LAB_ENCODE_T1") ? it + 1000 * (first_mu_it < 0 ? it : first_mu_it) : it;
  if (kkt_out) { lo_kkt(F, W->x, p, W->y, kkt_out); (void)e_du; }
  free(W->x); free(W->xt); free(W->dx); free(gx); free(W->g); free(W->gt); free(W->s); free(W->ds); free(W->zL); free(W->zU);
  free(W->dzL); free(W->dzU); free(W->y); free(W->yn); free(W->lb); free(W->ub); free(W->sig); free(W->rho); free(W->Jst); free(W->Hst);
  free(W->M); free(W->mvec); free(W->Ah); free(W->bv); free(W->K); free(W->kap); free(W->Px); free(W->pvx);
  return status;
}

/* batch driver: B members, `threads` OpenMP threads (<=0: all).  counters[0..1] = total factorisations, trial points. */
int lo_solve_batch(const lo_form* F, int B, const double* p, const double* x0, const lo_solver_opts* opts, int threads,
                   double* x, double* lam_g, int* status, int* iters, double* kkt, long long* counters) {
  const lo_int nx = lo_nx(F->N), ng = lo_ng(F->N), np = lo_np(F->N);
  lo_solver_opts o; long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0; int b;
  lab_init();
  if (opts) o = *opts; else lo_solver_opts_default(&o);
#ifdef _OPENMP
  if (threads > 0) omp_set_num_threads(threads);
#else
  (void)threads;
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : c0, c1, c2, c3, c4)
  for (b = 0; b < B; ++b) {
    long long cc[5] = {0, 0, 0, 0, 0};
    status[b] = solve_one(F, p + (size_t)b * np, x0 + (size_t)b * nx, &o, x + (size_t)b * nx, lam_g ? lam_g + (size_t)b * ng : NULL,
                          iters ? iters + b : NULL, kkt ? kkt + 3 * (size_t)b : NULL, cc);
    c0 += cc[0]; c1 += cc[1]; c2 += cc[2]; c3 += cc[3]; c4 += cc[4];
  }
  if (counters) { counters[0] = c0; counters[1] = c1; counters[2] = c2; counters[3] = c3; counters[4] = c4; }
  return 0;
}

/* Full derivative sweeps of a batch (g, grad f, Jacobian and Hessian nonzeros of every member; SURVEY 8d unit of work for
 * the function layer), `reps` times, OpenMP over members: CPU timing leg of bench.py.  Outputs go to per-thread scratch. */
int lo_sweep_batch(const lo_form* F, int B, const double* x, const double* p, const double* lam_g, int reps, int threads) {
  const lo_int nx = lo_nx(F->N), ng = lo_ng(F->N), np = lo_np(F->N), nj = lo_nnz_jac(F->N), nh = lo_nnz_hess(F->N);
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
