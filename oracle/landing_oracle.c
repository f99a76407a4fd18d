/*
 * landing_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.  See landing_oracle.h.
 *
 * Every function cites the reference lines it restates ("ref:" = path under
 * /root/reference/optimizations/landing/, gen = generate_solver/generate_landingCtrller_IPOPT.m).
 */
#include "landing_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* sizes, layouts                                                                        */
/* ------------------------------------------------------------------------------------ */
lo_int lo_nx(int N) { return 36 * (lo_int)N + 12; }
lo_int lo_ng(int N) { return 104 * (lo_int)N + 12; }
lo_int lo_np(int N) { return 13 * (lo_int)N + 94; }
lo_int lo_nnz_jac(int N) { return 36 + 385 * (lo_int)(N - 1) + 313; }
lo_int lo_nnz_hess(int N) { return 177 * (lo_int)N + 12 * (lo_int)(N - 1) + 12; }

void lo_form_default(lo_form* F, int N) {
  F->N = N;
  F->kin_box[0] = 0.15; F->kin_box[1] = 0.15; F->kin_box[2] = 0.30; /* gen:149-151 */
  F->kin_z_off = 0.05;                                                /* gen:155 */
  F->comp_eps = 1e-3;                                                 /* gen:140 */
  F->slip_eps = 1e-2;                                                 /* gen:143-144 */
  F->run_cost = 0;
  { int i; static const double ph[12] = {0.19, -0.1, -0.2, 0.19, 0.1, -0.2, -0.19, -0.1, -0.2, -0.19, 0.1, -0.2};   /* CCC :76-79 */
    for (i = 0; i < 12; i++) { F->QX[i] = 0.0; F->p_hip[i] = ph[i]; }
    for (i = 0; i < 3; i++) { F->Qc[i] = 0.0; F->Qf[i] = 0.0; F->f_ref[i] = 0.0; } }
}

static double rcQX(const lo_form* F, const lo_poff* o, const double* p, int i);
static double rcQc(const lo_form* F, const lo_poff* o, const double* p, int a);
static double rcQf(const lo_form* F, const lo_poff* o, const double* p, int a);
static double rcfref(const lo_form* F, const lo_poff* o, const double* p, int k, int l, int a);
double lo_run_cost_stage(const lo_form* F, const double* x, const double* p, int k, double* gX, double* gc, double* gf) {
  /* CCC :81-89 */
  const int N = F->N; lo_poff o; int i, l, a; double s = 0.0, dt;
  const double* X = x + 12 * k; const double* U = x + 12 * (N + 1) + 24 * k;
  lo_param_offsets_form(F, &o);
  dt = p[o.dt + k];
  for (i = 0; i < 12; i++) {
    const double e = X[i] - p[o.Xref + 12 * k + i], q = rcQX(F, &o, p, i);
    s += q * e * e;
    if (gX) gX[i] += 2.0 * dt * q * e;
  }
  for (l = 0; l < 4; l++) for (a = 0; a < 3; a++) {
    const double r = X[a] + F->p_hip[3 * l + a] - U[3 * l + a], u = U[12 + 3 * l + a] - rcfref(F, &o, p, k, l, a);
    const double qc = rcQc(F, &o, p, a), qf = rcQf(F, &o, p, a);
    s += qc * r * r + qf * u * u;
    if (gX) gX[a] += 2.0 * dt * qc * r;
    if (gc) gc[3 * l + a] -= 2.0 * dt * qc * r;
    if (gf) gf[3 * l + a] += 2.0 * dt * qf * u;
  }
  return dt * s;
}

/* form-aware offsets: run_cost == 2 is the N=41 script's own parameter vector (generate_quadruped_SRBM_CCC.m:49-71, Opti's order of
 * the active parameters; c_init is declared but unused and dropped): Xref | Uref | dt | bounds 60 | QX | QN | Qc | Qf | mu .. | Ib | Ib_inv */
void lo_param_offsets_form(const lo_form* F, lo_poff* o) {
  const int N = F->N; int b;
  lo_param_offsets(N, o);
  o->Uref = o->QX = o->Qc = o->Qf = -1;
  if (F->run_cost != 2) return;
  o->Uref = 12 * (N + 1);
  o->dt = o->Uref + 24 * N;
  b = o->dt + N;
  o->q_min = b; o->q_max = b + 6; o->qd_min = b + 12; o->qd_max = b + 18;
  o->q_init = b + 24; o->qd_init = b + 30;
  o->q_term_min = b + 36; o->q_term_max = b + 42; o->qd_term_min = b + 48; o->qd_term_max = b + 54;
  o->QX = b + 60; o->QN = b + 72; o->Qc = b + 84; o->Qf = b + 87;
  o->mu = b + 90; o->l_leg_max = b + 91; o->f_max = b + 92; o->mass = b + 93; o->Ib = b + 94; o->Ib_inv = b + 97; o->np = b + 100;
}
double lo_rc_weight(const lo_form* F, const double* p, int which, int i) {      /* which: 0 QX[i], 1 Qc[i], 2 Qf[i] -- from the form or from p (run_cost 2) */
  lo_poff o;
  if (F->run_cost != 2) return which == 0 ? F->QX[i] : (which == 1 ? F->Qc[i] : F->Qf[i]);
  lo_param_offsets_form(F, &o);
  return p[(which == 0 ? o.QX : (which == 1 ? o.Qc : o.Qf)) + i];
}
lo_int lo_np_form(const lo_form* F) { lo_poff o; lo_param_offsets_form(F, &o); return o.np; }
/* weights / force reference of the running cost: constants of the form (run_cost 1) or entries of p (run_cost 2) */
static double rcQX(const lo_form* F, const lo_poff* o, const double* p, int i) { return F->run_cost == 2 ? p[o->QX + i] : F->QX[i]; }
static double rcQc(const lo_form* F, const lo_poff* o, const double* p, int a) { return F->run_cost == 2 ? p[o->Qc + a] : F->Qc[a]; }
static double rcQf(const lo_form* F, const lo_poff* o, const double* p, int a) { return F->run_cost == 2 ? p[o->Qf + a] : F->Qf[a]; }
static double rcfref(const lo_form* F, const lo_poff* o, const double* p, int k, int l, int a) { return F->run_cost == 2 ? p[o->Uref + 24 * k + 12 + 3 * l + a] : F->f_ref[a]; }

void lo_param_offsets(int N, lo_poff* o) { /* gen:51-75, order of opti.parameter() calls */
  int b;
  o->Xref = 0;
  o->dt = 12 * (N + 1);
  b = o->dt + N;
  o->q_min = b; o->q_max = b + 6; o->qd_min = b + 12; o->qd_max = b + 18;
  o->q_init = b + 24; o->qd_init = b + 30;
  o->q_term_min = b + 36; o->q_term_max = b + 42; o->qd_term_min = b + 48; o->qd_term_max = b + 54;
  o->QN = b + 60; o->mu = b + 72; o->l_leg_max = b + 73; o->f_max = b + 74; o->mass = b + 75;
  o->Ib = b + 76; o->Ib_inv = b + 79; o->np = b + 82;
  o->Uref = o->QX = o->Qc = o->Qf = -1;
}

/* hipSrbmLocation, ref: utilities_general/dynamics-utilities/get_robot_params.m:90-91 */
static const double HIP[4][3] = {{0.19, -0.1, 0.0}, {0.19, 0.1, 0.0}, {-0.19, -0.1, 0.0}, {-0.19, 0.1, 0.0}};
static const double GRAV[3] = {0.0, 0.0, -9.81}; /* get_robot_model.m:140 */
static const double FRIC = 0.71;                 /* gen:160-163 */

/* local variable indices of a stage */
enum { LP = 0, LE = 3, LW = 6, LV = 9, LC = 12, LF = 24, LXN = 36, LCN = 48 };

int lo_stage_rows(const lo_form* F, int k) { return (k == F->N - 1) ? 80 : 104; }

/* local -> global variable index (x = [X(:);U(:)], gen:41-47) */
static lo_int loc2glob(int N, int k, int loc) {
  if (loc < 12) return 12 * (lo_int)k + loc;
  if (loc < 36) return 12 * (lo_int)(N + 1) + 24 * (lo_int)k + (loc - 12);
  if (loc < 48) return 12 * (lo_int)(k + 1) + (loc - 36);
  return 12 * (lo_int)(N + 1) + 24 * (lo_int)(k + 1) + (loc - 48);
}

/* row layout inside a stage */
typedef struct { int leg0, leg_stride, kin, fric, box, slip; } rowmap;
static rowmap stage_rowmap(int last) {
  rowmap m;
  m.leg0 = 16;
  if (!last) { m.leg_stride = 12; m.slip = 1; m.kin = 8; m.fric = 64; m.box = 80; }
  else       { m.leg_stride = 6;  m.slip = 0; m.kin = 2; m.fric = 40; m.box = 56; }
  return m;
}

/* ------------------------------------------------------------------------------------ */
/* small 3x3 helpers                                                                     */
/* ------------------------------------------------------------------------------------ */
typedef double m3[3][3];
static void mm(const m3 a, const m3 b, m3 c) {
  int i, j, k;
  for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) { double s = 0; for (k = 0; k < 3; k++) s += a[i][k] * b[k][j]; c[i][j] = s; }
}
static void mv(const m3 a, const double* v, double* o) {
  int i; for (i = 0; i < 3; i++) o[i] = a[i][0] * v[0] + a[i][1] * v[1] + a[i][2] * v[2];
}
static void mtv(const m3 a, const double* v, double* o) { /* a^T v */
  int i; for (i = 0; i < 3; i++) o[i] = a[0][i] * v[0] + a[1][i] * v[1] + a[2][i] * v[2];
}
static void cross(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static double eps3(int i, int j, int k) { /* Levi-Civita */
  if (i == j || j == k || i == k) return 0.0;
  return ((j - i + 3) % 3 == 1) ? 1.0 : -1.0;
}

/* Rotation R = rz(psi)' ry(theta)' rx(phi)'  (rpyToRotMat.m:2 with rx.m/ry.m/rz.m) and its
 * first/second derivatives w.r.t. e=(phi,theta,psi).  d1[a], d2[a][b]. */
typedef struct { m3 R; m3 d1[3]; m3 d2[3][3]; } rotset;
static void rot_elem(int axis, double ang, m3 r0, m3 r1, m3 r2) {
  double c = cos(ang), s = sin(ang);
  memset(r0, 0, sizeof(m3)); memset(r1, 0, sizeof(m3)); memset(r2, 0, sizeof(m3));
  if (axis == 0) {       /* rx(phi)' = [1 0 0; 0 c -s; 0 s c] */
    r0[0][0] = 1; r0[1][1] = c; r0[1][2] = -s; r0[2][1] = s; r0[2][2] = c;
    r1[1][1] = -s; r1[1][2] = -c; r1[2][1] = c; r1[2][2] = -s;
    r2[1][1] = -c; r2[1][2] = s; r2[2][1] = -s; r2[2][2] = -c;
  } else if (axis == 1) { /* ry(theta)' = [c 0 s; 0 1 0; -s 0 c] */
    r0[0][0] = c; r0[0][2] = s; r0[1][1] = 1; r0[2][0] = -s; r0[2][2] = c;
    r1[0][0] = -s; r1[0][2] = c; r1[2][0] = -c; r1[2][2] = -s;
    r2[0][0] = -c; r2[0][2] = -s; r2[2][0] = s; r2[2][2] = -c;
  } else {                /* rz(psi)' = [c -s 0; s c 0; 0 0 1] */
    r0[0][0] = c; r0[0][1] = -s; r0[1][0] = s; r0[1][1] = c; r0[2][2] = 1;
    r1[0][0] = -s; r1[0][1] = -c; r1[1][0] = c; r1[1][1] = -s;
    r2[0][0] = -c; r2[0][1] = s; r2[1][0] = -s; r2[1][1] = -c;
  }
}
static void rot3(const m3 z, const m3 y, const m3 x, m3 out) { m3 t; mm(z, y, t); mm(t, x, out); }
static void rotset_eval(const double* e, rotset* S, int second) {
  m3 X[3], Y[3], Z[3];
  rot_elem(0, e[0], X[0], X[1], X[2]);
  rot_elem(1, e[1], Y[0], Y[1], Y[2]);
  rot_elem(2, e[2], Z[0], Z[1], Z[2]);
  rot3(Z[0], Y[0], X[0], S->R);
  rot3(Z[0], Y[0], X[1], S->d1[0]);
  rot3(Z[0], Y[1], X[0], S->d1[1]);
  rot3(Z[1], Y[0], X[0], S->d1[2]);
  if (second) {
    rot3(Z[0], Y[0], X[2], S->d2[0][0]);
    rot3(Z[0], Y[2], X[0], S->d2[1][1]);
    rot3(Z[2], Y[0], X[0], S->d2[2][2]);
    rot3(Z[0], Y[1], X[1], S->d2[0][1]);
    rot3(Z[1], Y[0], X[1], S->d2[0][2]);
    rot3(Z[1], Y[1], X[0], S->d2[1][2]);
    memcpy(S->d2[1][0], S->d2[0][1], sizeof(m3));
    memcpy(S->d2[2][0], S->d2[0][2], sizeof(m3));
    memcpy(S->d2[2][1], S->d2[1][2], sizeof(m3));
  }
}
/* Binv(rpy), Binv.m:13-17, with derivatives w.r.t. e (phi-derivatives vanish). */
typedef struct { m3 B; m3 d1[3]; m3 d2[3][3]; } binvset;
static void binv_eval(const double* e, binvset* S) {
  double th = e[1], ps = e[2];
  double c = cos(ps), s = sin(ps), ct = cos(th), t = tan(th), sec = 1.0 / ct, sec2 = sec * sec;
  int a, b;
  memset(S, 0, sizeof(*S));
  S->B[0][0] = c / ct; S->B[0][1] = s / ct;
  S->B[1][0] = -s;     S->B[1][1] = c;
  S->B[2][0] = c * t;  S->B[2][1] = s * t; S->B[2][2] = 1.0;
  /* d/dtheta */
  S->d1[1][0][0] = c * sec * t; S->d1[1][0][1] = s * sec * t;
  S->d1[1][2][0] = c * sec2;    S->d1[1][2][1] = s * sec2;
  /* d/dpsi */
  S->d1[2][0][0] = -s * sec; S->d1[2][0][1] = c * sec;
  S->d1[2][1][0] = -c;       S->d1[2][1][1] = -s;
  S->d1[2][2][0] = -s * t;   S->d1[2][2][1] = c * t;
  /* second */
  S->d2[1][1][0][0] = c * sec * (t * t + sec2); S->d2[1][1][0][1] = s * sec * (t * t + sec2);
  S->d2[1][1][2][0] = c * 2 * sec2 * t;         S->d2[1][1][2][1] = s * 2 * sec2 * t;
  S->d2[1][2][0][0] = -s * sec * t; S->d2[1][2][0][1] = c * sec * t;
  S->d2[1][2][2][0] = -s * sec2;    S->d2[1][2][2][1] = c * sec2;
  S->d2[2][2][0][0] = -c * sec; S->d2[2][2][0][1] = -s * sec;
  S->d2[2][2][1][0] = s;        S->d2[2][2][1][1] = -c;
  S->d2[2][2][2][0] = -c * t;   S->d2[2][2][2][1] = -s * t;
  for (a = 0; a < 3; a++) for (b = 0; b < a; b++) memcpy(S->d2[a][b], S->d2[b][a], sizeof(m3));
}

/* ------------------------------------------------------------------------------------ */
/* one stage: residual rows, dense Jacobian, dense Hessian of lam^T g_k                  */
/* ------------------------------------------------------------------------------------ */
#define JJ(r, c) J[(r) * LO_NLOC + (c)]
static void hadd(double* H, int a, int b, double v) {
  H[a * LO_NLOC + b] += v;
  if (a != b) H[b * LO_NLOC + a] += v;
}

void lo_stage_eval(const lo_form* F, int k, const double* x, const double* p,
                   const double* lam, double* g, double* J, double* H) {
  const int N = F->N, last = (k == N - 1);
  const rowmap rm = stage_rowmap(last);
  lo_poff o;
  double z[LO_NLOC];
  int i, j, a, b, l;
  double dt, mu, mass, Ib[3], Ibi[3];
  rotset RS; binvset BS;
  double y[3], edot[3], fsum[3], rdd[3], tau_w[3], tau_b[3], nn[3], omd[3];
  double r[4][3];
  const double *pp, *e, *w, *v;
  const int need2 = (H != NULL && lam != NULL);

  lo_param_offsets_form(F, &o);
  dt = p[o.dt + k]; mu = p[o.mu]; mass = p[o.mass];
  for (i = 0; i < 3; i++) { Ib[i] = p[o.Ib + i]; Ibi[i] = p[o.Ib_inv + i]; }
  for (i = 0; i < LO_NLOC; i++) {
    if (last && i >= LCN) z[i] = 0.0; else z[i] = x[loc2glob(N, k, i)];
  }
  pp = z + LP; e = z + LE; w = z + LW; v = z + LV;
  if (J) memset(J, 0, sizeof(double) * LO_NROW * LO_NLOC);
  if (need2) memset(H, 0, sizeof(double) * LO_NLOC * LO_NLOC);

  rotset_eval(e, &RS, 1);
  binv_eval(e, &BS);

  /* ---- dynamics, gen:114-130 ---- */
  mv(RS.R, w, y);               /* R_body_to_world*qdk(1:3) */
  mv(BS.B, y, edot);            /* Binv(rpyk)*(...)   gen:128 */
  fsum[0] = fsum[1] = fsum[2] = 0; tau_w[0] = tau_w[1] = tau_w[2] = 0;
  for (l = 0; l < 4; l++) {
    double t[3];
    for (i = 0; i < 3; i++) { r[l][i] = z[LC + 3 * l + i] - pp[i]; fsum[i] += z[LF + 3 * l + i]; }
    cross(r[l], z + LF + 3 * l, t);             /* cross(ck-qk(1:3),fk) gen:120-123 */
    for (i = 0; i < 3; i++) tau_w[i] += t[i];
  }
  for (i = 0; i < 3; i++) rdd[i] = fsum[i] / mass + GRAV[i];   /* gen:118 */
  mtv(RS.R, tau_w, tau_b);                                       /* R_world_to_body*... */
  { double Iw[3] = {Ib[0] * w[0], Ib[1] * w[1], Ib[2] * w[2]}; cross(w, Iw, nn); } /* gen:124 */
  for (i = 0; i < 3; i++) omd[i] = Ibi[i] * (tau_b[i] - nn[i]);  /* gen:119 */

  if (g) {
    for (i = 0; i < 3; i++) {
      g[0 + i] = z[LXN + LP + i] - pp[i] - v[i] * dt;          /* gen:127 */
      g[3 + i] = z[LXN + LE + i] - e[i] - edot[i] * dt;        /* gen:128 */
      g[6 + i] = z[LXN + LV + i] - v[i] - rdd[i] * dt;         /* gen:129 */
      g[9 + i] = z[LXN + LW + i] - w[i] - omd[i] * dt;         /* gen:130 */
    }
  }
  if (J) {
    m3 T; mm(BS.B, RS.R, T);
    for (i = 0; i < 3; i++) {
      JJ(0 + i, LXN + LP + i) = 1; JJ(0 + i, LP + i) = -1; JJ(0 + i, LV + i) = -dt;
      JJ(3 + i, LXN + LE + i) = 1; JJ(3 + i, LE + i) += -1;
      for (j = 0; j < 3; j++) JJ(3 + i, LW + j) = -dt * T[i][j];
      JJ(6 + i, LXN + LV + i) = 1; JJ(6 + i, LV + i) = -1;
      for (l = 0; l < 4; l++) JJ(6 + i, LF + 3 * l + i) = -dt / mass;
      JJ(9 + i, LXN + LW + i) = 1; JJ(9 + i, LW + i) += -1;
    }
    for (a = 0; a < 3; a++) {   /* d edot / d e_a = dB_a y + B (dR_a w) */
      double t1[3], t2[3], t3[3];
      mv(BS.d1[a], y, t1); mv(RS.d1[a], w, t2); mv(BS.B, t2, t3);
      for (i = 0; i < 3; i++) JJ(3 + i, LE + a) += -dt * (t1[i] + t3[i]);
      mtv(RS.d1[a], tau_w, t1);  /* d tau_b / d e_a */
      for (i = 0; i < 3; i++) JJ(9 + i, LE + a) += -dt * Ibi[i] * t1[i];
    }
    /* d n / d w */
    {
      double dn[3][3] = {{0, (Ib[2] - Ib[1]) * w[2], (Ib[2] - Ib[1]) * w[1]},
                         {(Ib[0] - Ib[2]) * w[2], 0, (Ib[0] - Ib[2]) * w[0]},
                         {(Ib[1] - Ib[0]) * w[1], (Ib[1] - Ib[0]) * w[0], 0}};
      for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) JJ(9 + i, LW + j) += dt * Ibi[i] * dn[i][j];
    }
    /* d tau_w/d c_l = -S(f_l) ; d/d p = +sum S(f_l) ; d/d f_l = S(r_l) ;  S(a) b = a x b */
    for (l = 0; l < 4; l++) {
      const double* f = z + LF + 3 * l;
      for (j = 0; j < 3; j++) {
        double dc[3], df[3], ej[3] = {0, 0, 0}, tb[3];
        ej[j] = 1;
        cross(ej, f, dc);        /* d(r x f)/d r_j = e_j x f */
        cross(r[l], ej, df);     /* d(r x f)/d f_j = r x e_j */
        mtv(RS.R, dc, tb);
        for (i = 0; i < 3; i++) { JJ(9 + i, LC + 3 * l + j) += -dt * Ibi[i] * tb[i]; JJ(9 + i, LP + j) += dt * Ibi[i] * tb[i]; }
        mtv(RS.R, df, tb);
        for (i = 0; i < 3; i++) JJ(9 + i, LF + 3 * l + j) += -dt * Ibi[i] * tb[i];
      }
    }
  }
  if (need2) {
    const double* le = lam + 3; const double* lw = lam + 9;
    double mu_b[3] = {lw[0] * Ibi[0], lw[1] * Ibi[1], lw[2] * Ibi[2]};
    double mvec[3], Ra_mu[3][3];
    mv(RS.R, mu_b, mvec);
    for (a = 0; a < 3; a++) mv(RS.d1[a], mu_b, Ra_mu[a]);
    /* rpy rows: -dt * le^T T(e) w */
    for (a = 0; a < 3; a++) {
      m3 t1, t2, Ta;
      mm(BS.d1[a], RS.R, t1); mm(BS.B, RS.d1[a], t2);
      for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) Ta[i][j] = t1[i][j] + t2[i][j];
      for (j = 0; j < 3; j++) {
        double s = 0; for (i = 0; i < 3; i++) s += le[i] * Ta[i][j];
        hadd(H, LE + a, LW + j, -dt * s);
      }
      for (b = a; b < 3; b++) {
        m3 u1, u2, u3, u4; double tw[3], s = 0;
        mm(BS.d2[a][b], RS.R, u1); mm(BS.d1[a], RS.d1[b], u2); mm(BS.d1[b], RS.d1[a], u3); mm(BS.B, RS.d2[a][b], u4);
        for (i = 0; i < 3; i++) { tw[i] = 0; for (j = 0; j < 3; j++) tw[i] += (u1[i][j] + u2[i][j] + u3[i][j] + u4[i][j]) * w[j]; }
        for (i = 0; i < 3; i++) s += le[i] * tw[i];
        hadd(H, LE + a, LE + b, -dt * s);
      }
    }
    /* omega rows: -dt*( m(e).tau_w - mu_b.n ) */
    for (a = 0; a < 3; a++) for (b = a; b < 3; b++) {
      double t[3]; mv(RS.d2[a][b], mu_b, t);
      hadd(H, LE + a, LE + b, -dt * dot3(t, tau_w));
    }
    for (l = 0; l < 4; l++) {
      const double* f = z + LF + 3 * l;
      for (a = 0; a < 3; a++) {
        double t[3];
        cross(f, Ra_mu[a], t);            /* d/dr of (Ra mu).(r x f) = f x (Ra mu) */
        for (i = 0; i < 3; i++) { hadd(H, LE + a, LC + 3 * l + i, -dt * t[i]); hadd(H, LE + a, LP + i, dt * t[i]); }
        cross(Ra_mu[a], r[l], t);         /* d/df = (Ra mu) x r */
        for (i = 0; i < 3; i++) hadd(H, LE + a, LF + 3 * l + i, -dt * t[i]);
      }
      for (i = 0; i < 3; i++) for (j = 0; j < 3; j++) {
        double s = 0; int q;
        if (i == j) continue;
        for (q = 0; q < 3; q++) s += eps3(i, j, q) * mvec[q];
        hadd(H, LC + 3 * l + i, LF + 3 * l + j, -dt * s);
        hadd(H, LP + i, LF + 3 * l + j, dt * s);
      }
    }
    hadd(H, LW + 1, LW + 2, dt * mu_b[0] * (Ib[2] - Ib[1]));
    hadd(H, LW + 0, LW + 2, dt * mu_b[1] * (Ib[0] - Ib[2]));
    hadd(H, LW + 0, LW + 1, dt * mu_b[2] * (Ib[1] - Ib[0]));
  }

  /* ---- non-negative GRF rows gen:133 ---- */
  for (l = 0; l < 4; l++) {
    if (g) g[12 + l] = z[LF + 3 * l + 2];
    if (J) JJ(12 + l, LF + 3 * l + 2) = 1;
  }
  /* ---- per-leg contact / kinematic rows gen:137-157 ---- */
  for (l = 0; l < 4; l++) {
    const int rb = rm.leg0 + rm.leg_stride * l;
    const int ic = LC + 3 * l, iff = LF + 3 * l, icn = LCN + 3 * l;
    const double fz = z[iff + 2], cz = z[ic + 2];
    double Rh[3], prel[3], Rah[3][3];
    if (g) { g[rb + 0] = cz; g[rb + 1] = fz * cz; }           /* gen:139-140 */
    if (J) { JJ(rb + 0, ic + 2) = 1; JJ(rb + 1, iff + 2) = cz; JJ(rb + 1, ic + 2) = fz; }
    if (need2) hadd(H, iff + 2, ic + 2, lam[rb + 1]);
    if (rm.slip) {                                               /* gen:143-144 */
      for (i = 0; i < 3; i++) {
        double d = z[icn + i] - z[ic + i];
        if (g) { g[rb + 2 + i] = fz * d; g[rb + 5 + i] = fz * d; }
        if (J) {
          int q;
          for (q = 0; q < 2; q++) {
            JJ(rb + 2 + 3 * q + i, iff + 2) = d; JJ(rb + 2 + 3 * q + i, icn + i) = fz; JJ(rb + 2 + 3 * q + i, ic + i) = -fz;
          }
        }
        if (need2) {
          double ls = lam[rb + 2 + i] + lam[rb + 5 + i];
          hadd(H, iff + 2, icn + i, ls); hadd(H, iff + 2, ic + i, -ls);
        }
      }
    }
    mv(RS.R, HIP[l], Rh);                                        /* gen:147-148 */
    for (i = 0; i < 3; i++) prel[i] = z[ic + i] - (pp[i] + Rh[i]);
    for (a = 0; a < 3; a++) mv(RS.d1[a], HIP[l], Rah[a]);
    {
      const int rk = rb + rm.kin;
      if (g) {
        g[rk + 0] = prel[0]; g[rk + 1] = prel[1]; g[rk + 2] = prel[2] + F->kin_z_off;  /* gen:153-155 */
        g[rk + 3] = dot3(prel, prel);                                                   /* gen:156 */
      }
      if (J) {
        for (i = 0; i < 3; i++) {
          JJ(rk + i, ic + i) = 1; JJ(rk + i, LP + i) = -1;
          for (a = 0; a < 3; a++) JJ(rk + i, LE + a) = -Rah[a][i];
          JJ(rk + 3, ic + i) = 2 * prel[i]; JJ(rk + 3, LP + i) = -2 * prel[i];
        }
        for (a = 0; a < 3; a++) JJ(rk + 3, LE + a) = -2 * dot3(prel, Rah[a]);
      }
      if (need2) {
        const double lL = lam[rk + 3];
        for (a = 0; a < 3; a++) for (b = a; b < 3; b++) {
          double t[3]; mv(RS.d2[a][b], HIP[l], t);
          hadd(H, LE + a, LE + b, -(lam[rk] * t[0] + lam[rk + 1] * t[1] + lam[rk + 2] * t[2]));
          hadd(H, LE + a, LE + b, 2 * lL * (dot3(Rah[a], Rah[b]) - dot3(prel, t)));
        }
        for (i = 0; i < 3; i++) {
          hadd(H, ic + i, ic + i, 2 * lL); hadd(H, LP + i, LP + i, 2 * lL); hadd(H, LP + i, ic + i, -2 * lL);
          for (a = 0; a < 3; a++) { hadd(H, LE + a, ic + i, -2 * lL * Rah[a][i]); hadd(H, LP + i, LE + a, 2 * lL * Rah[a][i]); }
        }
      }
    }
  }
  /* ---- friction pyramid gen:160-163 ---- */
  for (l = 0; l < 4; l++) {
    const int iff = LF + 3 * l; const double km = FRIC * mu;
    if (g) {
      g[rm.fric + l] = z[iff] - km * z[iff + 2];
      g[rm.fric + 4 + l] = -km * z[iff + 2] - z[iff];
      g[rm.fric + 8 + l] = z[iff + 1] - km * z[iff + 2];
      g[rm.fric + 12 + l] = -km * z[iff + 2] - z[iff + 1];
    }
    if (J) {
      JJ(rm.fric + l, iff) = 1; JJ(rm.fric + l, iff + 2) = -km;
      JJ(rm.fric + 4 + l, iff) = -1; JJ(rm.fric + 4 + l, iff + 2) = -km;
      JJ(rm.fric + 8 + l, iff + 1) = 1; JJ(rm.fric + 8 + l, iff + 2) = -km;
      JJ(rm.fric + 12 + l, iff + 1) = -1; JJ(rm.fric + 12 + l, iff + 2) = -km;
    }
  }
  /* ---- state boxes gen:166-169 ---- */
  for (i = 0; i < 6; i++) {
    if (g) { g[rm.box + i] = z[i]; g[rm.box + 6 + i] = z[i]; g[rm.box + 12 + i] = z[6 + i]; g[rm.box + 18 + i] = z[6 + i]; }
    if (J) { JJ(rm.box + i, i) = 1; JJ(rm.box + 6 + i, i) = 1; JJ(rm.box + 12 + i, 6 + i) = 1; JJ(rm.box + 18 + i, 6 + i) = 1; }
  }
}

/* ------------------------------------------------------------------------------------ */
/* structural patterns (what CasADi's symbolic Jacobian/Hessian of the UNSIMPLIFIED       */
/* expressions contain; verified == casadi_s4/casadi_s5 at N=20 in tests)                 */
/* ------------------------------------------------------------------------------------ */
static void stage_jac_struct(int last, unsigned char* S /* [104][60] */) {
  const rowmap rm = stage_rowmap(last);
  int i, j, l, a, q;
#define SS(r, c) S[(r) * LO_NLOC + (c)]
  memset(S, 0, LO_NROW * LO_NLOC);
  for (i = 0; i < 3; i++) {
    SS(i, LXN + LP + i) = 1; SS(i, LP + i) = 1; SS(i, LV + i) = 1;
    SS(3 + i, LXN + LE + i) = 1;
    for (j = 0; j < 3; j++) { SS(3 + i, LE + j) = 1; SS(3 + i, LW + j) = 1; }
    SS(6 + i, LXN + LV + i) = 1; SS(6 + i, LV + i) = 1;
    for (l = 0; l < 4; l++) SS(6 + i, LF + 3 * l + i) = 1;
    SS(9 + i, LXN + LW + i) = 1;
    for (j = 0; j < 3; j++) { SS(9 + i, LW + j) = 1; SS(9 + i, LP + j) = 1; SS(9 + i, LE + j) = 1; }
    for (j = 0; j < 12; j++) { SS(9 + i, LC + j) = 1; SS(9 + i, LF + j) = 1; }
  }
  SS(9, LE + 0) = 0;           /* first column of R has no roll dependence */
  for (l = 0; l < 4; l++) {
    const int rb = rm.leg0 + rm.leg_stride * l, ic = LC + 3 * l, iff = LF + 3 * l, icn = LCN + 3 * l;
    const int rk = rb + rm.kin;
    SS(12 + l, iff + 2) = 1;
    SS(rb, ic + 2) = 1; SS(rb + 1, iff + 2) = 1; SS(rb + 1, ic + 2) = 1;
    if (rm.slip) for (q = 0; q < 2; q++) for (i = 0; i < 3; i++) {
      SS(rb + 2 + 3 * q + i, iff + 2) = 1; SS(rb + 2 + 3 * q + i, icn + i) = 1; SS(rb + 2 + 3 * q + i, ic + i) = 1;
    }
    for (i = 0; i < 3; i++) {
      SS(rk + i, ic + i) = 1; SS(rk + i, LP + i) = 1;
      for (a = 0; a < 3; a++) SS(rk + i, LE + a) = 1;
      SS(rk + 3, ic + i) = 1; SS(rk + 3, LP + i) = 1; SS(rk + 3, LE + i) = 1;
    }
    SS(rk + 2, LE + 2) = 0;    /* (R*hip)_z with hip_z = 0 has no yaw dependence */
    SS(rm.fric + l, iff) = 1; SS(rm.fric + l, iff + 2) = 1;
    SS(rm.fric + 4 + l, iff) = 1; SS(rm.fric + 4 + l, iff + 2) = 1;
    SS(rm.fric + 8 + l, iff + 1) = 1; SS(rm.fric + 8 + l, iff + 2) = 1;
    SS(rm.fric + 12 + l, iff + 1) = 1; SS(rm.fric + 12 + l, iff + 2) = 1;
  }
  for (i = 0; i < 6; i++) { SS(rm.box + i, i) = 1; SS(rm.box + 6 + i, i) = 1; SS(rm.box + 12 + i, 6 + i) = 1; SS(rm.box + 18 + i, 6 + i) = 1; }
#undef SS
}
static void stage_hess_struct(int last, unsigned char* S /* [60][60], upper (a<=b) */) {
  int i, j, l, a;
#define HS(a_, b_) S[((a_) < (b_) ? (a_) : (b_)) * LO_NLOC + ((a_) < (b_) ? (b_) : (a_))]
  memset(S, 0, LO_NLOC * LO_NLOC);
  for (i = 0; i < 3; i++) {
    HS(LP + i, LP + i) = 1;
    for (a = 0; a < 3; a++) { HS(LP + i, LE + a) = 1; HS(LE + i, LE + a) = 1; HS(LE + a, LW + i) = 1; }
  }
  HS(LE + 0, LW + 0) = 0;      /* R(:,1) independent of roll */
  HS(LW + 0, LW + 1) = 1; HS(LW + 0, LW + 2) = 1; HS(LW + 1, LW + 2) = 1;
  for (l = 0; l < 4; l++) {
    const int ic = LC + 3 * l, iff = LF + 3 * l, icn = LCN + 3 * l;
    for (i = 0; i < 3; i++) {
      HS(ic + i, ic + i) = 1; HS(LP + i, ic + i) = 1;
      for (a = 0; a < 3; a++) { HS(LE + a, ic + i) = 1; HS(LE + a, iff + i) = 1; }
      for (j = 0; j < 3; j++) if (i != j) { HS(LP + i, iff + j) = 1; HS(ic + i, iff + j) = 1; }
      if (!last) HS(iff + 2, icn + i) = 1;
    }
    HS(ic + 2, iff + 2) = 1;
  }
#undef HS
}

/* boundary rows 0..35, gen:90-97 */
static void boundary_cols(int N, lo_int r, lo_int* col) {
  if (r < 12) *col = r;
  else if (r < 18) *col = 12 * (lo_int)N + (r - 12);
  else if (r < 24) *col = 12 * (lo_int)N + (r - 18);
  else if (r < 30) *col = 12 * (lo_int)N + 6 + (r - 24);
  else *col = 12 * (lo_int)N + 6 + (r - 30);
}

typedef struct { lo_int r, c; } rc_t;
static int rc_cmp(const void* a, const void* b) {
  const rc_t* x = (const rc_t*)a; const rc_t* y = (const rc_t*)b;
  if (x->c != y->c) return x->c < y->c ? -1 : 1;
  if (x->r != y->r) return x->r < y->r ? -1 : 1;
  return 0;
}
static void build_ccs(rc_t* e, lo_int n, lo_int ncol, lo_int* colind, lo_int* row) {
  lo_int i, c = 0;
  qsort(e, (size_t)n, sizeof(rc_t), rc_cmp);
  colind[0] = 0;
  for (i = 0; i < n; i++) {
    while (c < e[i].c) colind[++c] = i;
    row[i] = e[i].r;
  }
  while (c < ncol) colind[++c] = n;
}

void lo_pattern_jac(int N, lo_int* colind, lo_int* row) {
  const lo_int nnz = lo_nnz_jac(N);
  rc_t* e = (rc_t*)malloc(sizeof(rc_t) * (size_t)nnz);
  unsigned char S[LO_NROW * LO_NLOC];
  lo_int n = 0, r; int k, i, j;
  for (r = 0; r < 36; r++) { e[n].r = r; boundary_cols(N, r, &e[n].c); n++; }
  for (k = 0; k < N; k++) {
    const int last = (k == N - 1), nr = last ? 80 : 104;
    stage_jac_struct(last, S);
    for (i = 0; i < nr; i++) for (j = 0; j < LO_NLOC; j++) if (S[i * LO_NLOC + j]) {
      e[n].r = 36 + 104 * (lo_int)k + i; e[n].c = loc2glob(N, k, j); n++;
    }
  }
  build_ccs(e, n, lo_nx(N), colind, row);
  free(e);
}
void lo_pattern_hess(int N, lo_int* colind, lo_int* row) {
  const lo_int nnz = lo_nnz_hess(N);
  rc_t* e = (rc_t*)malloc(sizeof(rc_t) * (size_t)nnz);
  unsigned char S[LO_NLOC * LO_NLOC];
  lo_int n = 0; int k, i, j;
  for (k = 0; k < N; k++) {
    stage_hess_struct(k == N - 1, S);
    for (i = 0; i < LO_NLOC; i++) for (j = i; j < LO_NLOC; j++) if (S[i * LO_NLOC + j]) {
      lo_int gi = loc2glob(N, k, i), gj = loc2glob(N, k, j);
      e[n].r = gi < gj ? gi : gj; e[n].c = gi < gj ? gj : gi; n++;
    }
  }
  for (i = 0; i < 12; i++) { e[n].r = e[n].c = 12 * (lo_int)N + i; n++; }   /* terminal cost, gen:84-85 */
  build_ccs(e, n, lo_nx(N), colind, row);
  free(e);
}

/* position of (r,c) in a CCS pattern */
static lo_int ccs_find(const lo_int* colind, const lo_int* row, lo_int r, lo_int c) {
  lo_int lo = colind[c], hi = colind[c + 1] - 1;
  while (lo <= hi) { lo_int m = (lo + hi) / 2; if (row[m] == r) return m; if (row[m] < r) lo = m + 1; else hi = m - 1; }
  return -1;
}

/* ------------------------------------------------------------------------------------ */
/* NLP callbacks                                                                         */
/* ------------------------------------------------------------------------------------ */
void lo_nlp_grad_f(const lo_form* F, const double* x, const double* p, double* f, double* grad) {
  /* gen:83-87: cost = X_err'*diag(QN)*X_err, X_err = X(:,end)-Xref(:,end) */
  const int N = F->N; lo_poff o; int i; double s = 0;
  lo_param_offsets_form(F, &o);
  if (grad) memset(grad, 0, sizeof(double) * (size_t)lo_nx(N));
  for (i = 0; i < 12; i++) {
    double d = x[12 * N + i] - p[o.Xref + 12 * N + i];
    s += d * p[o.QN + i] * d;
    if (grad) grad[12 * N + i] = 2 * p[o.QN + i] * d;
  }
  if (F->run_cost) {
    int k;
    for (k = 0; k < N; k++) {
      double* gX = grad ? grad + 12 * k : NULL; double* gU = grad ? grad + 12 * (N + 1) + 24 * k : NULL;
      s += lo_run_cost_stage(F, x, p, k, gX, gU, gU ? gU + 12 : NULL);
    }
  }
  if (f) *f = s;
}
void lo_nlp_f(const lo_form* F, const double* x, const double* p, double* f) { lo_nlp_grad_f(F, x, p, f, NULL); }

static void boundary_g(const lo_form* F, const double* x, double* g) {
  const int N = F->N; int i;
  for (i = 0; i < 12; i++) g[i] = x[i];                               /* gen:90-91 */
  for (i = 0; i < 6; i++) {
    g[12 + i] = x[12 * N + i]; g[18 + i] = x[12 * N + i];            /* gen:94-95 */
    g[24 + i] = x[12 * N + 6 + i]; g[30 + i] = x[12 * N + 6 + i];    /* gen:96-97 */
  }
}
void lo_nlp_g(const lo_form* F, const double* x, const double* p, double* g) {
  int k;
  boundary_g(F, x, g);
  for (k = 0; k < F->N; k++) lo_stage_eval(F, k, x, p, NULL, g + 36 + 104 * k, NULL, NULL);
}

void lo_nlp_jac_g(const lo_form* F, const double* x, const double* p, double* g, double* jac) {
  const int N = F->N; const lo_int nx = lo_nx(N), nnz = lo_nnz_jac(N);
  lo_int* colind = (lo_int*)malloc(sizeof(lo_int) * (size_t)(nx + 1));
  lo_int* row = (lo_int*)malloc(sizeof(lo_int) * (size_t)nnz);
  double J[LO_NROW * LO_NLOC]; unsigned char S[LO_NROW * LO_NLOC];
  int k, i, j; lo_int r;
  lo_pattern_jac(N, colind, row);
  if (g) boundary_g(F, x, g);
  if (jac) {
    memset(jac, 0, sizeof(double) * (size_t)nnz);
    for (r = 0; r < 36; r++) { lo_int c; boundary_cols(N, r, &c); jac[ccs_find(colind, row, r, c)] = 1.0; }
  }
  for (k = 0; k < N; k++) {
    const int last = (k == N - 1), nr = last ? 80 : 104;
    lo_stage_eval(F, k, x, p, NULL, g ? g + 36 + 104 * k : NULL, jac ? J : NULL, NULL);
    if (!jac) continue;
    stage_jac_struct(last, S);
    for (i = 0; i < nr; i++) for (j = 0; j < LO_NLOC; j++) if (S[i * LO_NLOC + j])
      jac[ccs_find(colind, row, 36 + 104 * (lo_int)k + i, loc2glob(N, k, j))] = J[i * LO_NLOC + j];
  }
  free(colind); free(row);
}

void lo_nlp_hess_l(const lo_form* F, const double* x, const double* p, double lam_f,
                   const double* lam_g, double* hess) {
  const int N = F->N; const lo_int nx = lo_nx(N), nnz = lo_nnz_hess(N);
  lo_int* colind = (lo_int*)malloc(sizeof(lo_int) * (size_t)(nx + 1));
  lo_int* row = (lo_int*)malloc(sizeof(lo_int) * (size_t)nnz);
  double H[LO_NLOC * LO_NLOC]; unsigned char S[LO_NLOC * LO_NLOC];
  double lam[LO_NROW]; lo_poff o;
  int k, i, j;
  lo_param_offsets_form(F, &o);
  lo_pattern_hess(N, colind, row);
  memset(hess, 0, sizeof(double) * (size_t)nnz);
  for (k = 0; k < N; k++) {
    const int last = (k == N - 1), nr = last ? 80 : 104;
    for (i = 0; i < LO_NROW; i++) lam[i] = (i < nr && lam_g) ? lam_g[36 + 104 * k + i] : 0.0;
    lo_stage_eval(F, k, x, p, lam, NULL, NULL, H);
    stage_hess_struct(last, S);
    for (i = 0; i < LO_NLOC; i++) for (j = i; j < LO_NLOC; j++) if (S[i * LO_NLOC + j]) {
      lo_int gi = loc2glob(N, k, i), gj = loc2glob(N, k, j);
      hess[ccs_find(colind, row, gi < gj ? gi : gj, gi < gj ? gj : gi)] += H[i * LO_NLOC + j];
    }
  }
  for (i = 0; i < 12; i++) hess[ccs_find(colind, row, 12 * N + i, 12 * N + i)] += 2 * lam_f * p[o.QN + i];
  free(colind); free(row);
}

/* ---- Hessian of the Lagrangian WITH the running cost of the N=41 script (generate_quadruped_SRBM_CCC.m:81-89).  No generated C
 * of that script exists in the reference, so the CCS pattern is this project's: casadi_s4 (landingCtrller_IPOPT.c:63) plus the
 * diagonal entries the quadratic running cost adds and s4 lacks -- (omega, omega), (v, v), (f, f) of every stage:
 * 18 more per stage.  The other running-cost entries -- (pos, pos), (rpy, rpy), (pos_a, c_leg_a), (c, c) -- are already
 * structural nonzeros of s4.  Upper triangular, rows sorted inside a column (the new entries are diagonals: last in their column). */
lo_int lo_nnz_hess_rc(int N) { return lo_nnz_hess(N) + 18 * (lo_int)N; }
static int rc_new_diag(int N, lo_int c) {          /* is (c, c) one of the 18 N added diagonals? */
  const lo_int nX = 12 * (lo_int)(N + 1);
  if (c < 12 * (lo_int)N) return (c % 12) >= 6;                                  /* omega (6..8), v (9..11) of X_0..X_{N-1} */
  if (c >= nX) { const lo_int j = (c - nX) % 24; return j >= 12; }                  /* f */
  return 0;
}
void lo_pattern_hess_rc(int N, lo_int* colind, lo_int* row) {
  const lo_int nx = lo_nx(N), nnz = lo_nnz_hess(N);
  lo_int* ci = (lo_int*)malloc(sizeof(lo_int) * (size_t)(nx + 1));
  lo_int* ro = (lo_int*)malloc(sizeof(lo_int) * (size_t)nnz);
  lo_int c, i, n = 0;
  lo_pattern_hess(N, ci, ro);
  for (c = 0; c < nx; c++) {
    colind[c] = n;
    for (i = ci[c]; i < ci[c + 1]; i++) row[n++] = ro[i];
    if (rc_new_diag(N, c)) row[n++] = c;
  }
  colind[nx] = n;
  free(ci); free(ro);
}
void lo_nlp_hess_l_rc(const lo_form* F, const double* x, const double* p, double lam_f, const double* lam_g, double* hess) {
  const int N = F->N; const lo_int nx = lo_nx(N), nnz = lo_nnz_hess(N), nrc = lo_nnz_hess_rc(N);
  lo_int* ci = (lo_int*)malloc(sizeof(lo_int) * (size_t)(nx + 1)); lo_int* ro = (lo_int*)malloc(sizeof(lo_int) * (size_t)nrc);
  lo_int* c4 = (lo_int*)malloc(sizeof(lo_int) * (size_t)(nx + 1)); lo_int* r4 = (lo_int*)malloc(sizeof(lo_int) * (size_t)nnz);
  double* h4 = (double*)malloc(sizeof(double) * (size_t)nnz);
  lo_poff o; lo_int c, i; int k, l, a;
  (void)x;
  lo_param_offsets_form(F, &o);
  lo_pattern_hess_rc(N, ci, ro); lo_pattern_hess(N, c4, r4);
  lo_nlp_hess_l(F, x, p, lam_f, lam_g, h4);
  memset(hess, 0, sizeof(double) * (size_t)nrc);
  for (c = 0; c < nx; c++) for (i = c4[c]; i < c4[c + 1]; i++) hess[ccs_find(ci, ro, r4[i], c)] = h4[i];
  if (F->run_cost) for (k = 0; k < N; k++) {      /* second derivatives of dt_k (|X - Xref|^2_QX + sum_legs |pos + p_hip - c|^2_Qc + |f - f_ref|^2_Qf) */
    const double d2 = 2.0 * lam_f * p[o.dt + k];
    const lo_int X = 12 * (lo_int)k, U = 12 * (lo_int)(N + 1) + 24 * (lo_int)k;
    for (a = 0; a < 12; a++) hess[ccs_find(ci, ro, X + a, X + a)] += d2 * (rcQX(F, &o, p, a) + (a < 3 ? 4.0 * rcQc(F, &o, p, a) : 0.0));
    for (l = 0; l < 4; l++) for (a = 0; a < 3; a++) {
      hess[ccs_find(ci, ro, X + a, U + 3 * l + a)] += -d2 * rcQc(F, &o, p, a);
      hess[ccs_find(ci, ro, U + 3 * l + a, U + 3 * l + a)] += d2 * rcQc(F, &o, p, a);
      hess[ccs_find(ci, ro, U + 12 + 3 * l + a, U + 12 + 3 * l + a)] += d2 * rcQf(F, &o, p, a);
    }
  }
  free(ci); free(ro); free(c4); free(r4); free(h4);
}

void lo_nlp_grad(const lo_form* F, const double* x, const double* p, double lam_f,
                 const double* lam_g, double* f, double* g, double* grad_x, double* grad_p) {
  const int N = F->N; const lo_int nx = lo_nx(N), ng = lo_ng(N);
  lo_poff o; int k, i, j, l; lo_int r;
  double J[LO_NROW * LO_NLOC];
  double* gg = (double*)malloc(sizeof(double) * (size_t)ng);
  lo_param_offsets_form(F, &o);
  boundary_g(F, x, gg);
  if (grad_x) {
    double ff;
    lo_nlp_grad_f(F, x, p, &ff, grad_x);
    for (r = 0; r < nx; r++) grad_x[r] *= lam_f;
    for (r = 0; r < 36; r++) { lo_int c; boundary_cols(N, r, &c); grad_x[c] += lam_g[r]; }
  }
  if (grad_p) memset(grad_p, 0, sizeof(double) * (size_t)o.np);
  for (k = 0; k < N; k++) {
    const int last = (k == N - 1), nr = last ? 80 : 104;
    const rowmap rm = stage_rowmap(last);
    const double* lam = lam_g + 36 + 104 * k;
    double* gk = gg + 36 + 104 * k;
    lo_stage_eval(F, k, x, p, NULL, gk, J, NULL);
    if (grad_x) for (i = 0; i < nr; i++) for (j = 0; j < LO_NLOC; j++) {
      if (J[i * LO_NLOC + j] != 0.0) grad_x[loc2glob(N, k, j)] += lam[i] * J[i * LO_NLOC + j];
    }
    if (grad_p) {
      const double dt = p[o.dt + k], mass = p[o.mass], mu = p[o.mu];
      const double* Xk = x + 12 * k; const double* Xn = x + 12 * (k + 1);
      const double* Uk = x + 12 * (N + 1) + 24 * k;
      double s = 0, fs[3] = {0, 0, 0}, rate[12];
      (void)mu;
      /* rows are X+ - X - rate*dt  ->  rate = (X+ - X - g)/dt ; d g/d dt = -rate */
      {
        static const int xo[12] = {0, 1, 2, 3, 4, 5, 9, 10, 11, 6, 7, 8}; /* row -> state index */
        for (i = 0; i < 12; i++) { rate[i] = (Xn[xo[i]] - Xk[xo[i]] - gk[i]) / dt; s += lam[i] * (-rate[i]); }
      }
      grad_p[o.dt + k] += s;
      for (l = 0; l < 4; l++) for (i = 0; i < 3; i++) fs[i] += Uk[12 + 3 * l + i];
      for (i = 0; i < 3; i++) grad_p[o.mass] += lam[6 + i] * dt * fs[i] / (mass * mass);
      for (l = 0; l < 4; l++) {
        const double fz = Uk[12 + 3 * l + 2];
        grad_p[o.mu] += -FRIC * fz * (lam[rm.fric + l] + lam[rm.fric + 4 + l] + lam[rm.fric + 8 + l] + lam[rm.fric + 12 + l]);
      }
      {
        const double* w = Xk + 6;
        const double Ibi[3] = {p[o.Ib_inv], p[o.Ib_inv + 1], p[o.Ib_inv + 2]};
        /* omd_i = Ibi_i*(tau_b - n)_i ; rate[9+i] = omd_i */
        for (i = 0; i < 3; i++) grad_p[o.Ib_inv + i] += -dt * lam[9 + i] * rate[9 + i] / Ibi[i];
        /* n = w x (Ib.*w): dn_x/dIb = (0,-wy wz, wy wz), dn_y/dIb=(wz wx,0,-wx wz), dn_z/dIb=(-wx wy, wx wy,0) */
        grad_p[o.Ib + 0] += dt * (lam[10] * Ibi[1] * w[2] * w[0] - lam[11] * Ibi[2] * w[0] * w[1]);
        grad_p[o.Ib + 1] += dt * (-lam[9] * Ibi[0] * w[1] * w[2] + lam[11] * Ibi[2] * w[0] * w[1]);
        grad_p[o.Ib + 2] += dt * (lam[9] * Ibi[0] * w[1] * w[2] - lam[10] * Ibi[1] * w[0] * w[2]);
      }
    }
  }
  if (grad_p) for (i = 0; i < 12; i++) {
    double d = x[12 * N + i] - p[o.Xref + 12 * N + i];
    grad_p[o.Xref + 12 * N + i] += -2 * lam_f * p[o.QN + i] * d;
    grad_p[o.QN + i] += lam_f * d * d;
  }
  if (grad_p && F->run_cost) for (k = 0; k < N; k++) {   /* running cost: d/dXref_k and d/ddt_k; with run_cost 2 also d/dUref_k (force part), d/dQX, d/dQc, d/dQf */
    const double dt = p[o.dt + k];
    const double* X = x + 12 * k; const double* U = x + 12 * (N + 1) + 24 * k;
    for (i = 0; i < 12; i++) grad_p[o.Xref + 12 * k + i] += -2.0 * lam_f * dt * rcQX(F, &o, p, i) * (X[i] - p[o.Xref + 12 * k + i]);
    grad_p[o.dt + k] += lam_f * lo_run_cost_stage(F, x, p, k, NULL, NULL, NULL) / dt;
    if (F->run_cost == 2) {
      int l2, a2;
      for (i = 0; i < 12; i++) { const double e = X[i] - p[o.Xref + 12 * k + i]; grad_p[o.QX + i] += lam_f * dt * e * e; }
      for (l2 = 0; l2 < 4; l2++) for (a2 = 0; a2 < 3; a2++) {
        const double r = X[a2] + F->p_hip[3 * l2 + a2] - U[3 * l2 + a2], u = U[12 + 3 * l2 + a2] - p[o.Uref + 24 * k + 12 + 3 * l2 + a2];
        grad_p[o.Qc + a2] += lam_f * dt * r * r; grad_p[o.Qf + a2] += lam_f * dt * u * u;
        grad_p[o.Uref + 24 * k + 12 + 3 * l2 + a2] += -2.0 * lam_f * dt * p[o.Qf + a2] * u;
      }
    }
  }
  if (f) lo_nlp_f(F, x, p, f);
  if (g) memcpy(g, gg, sizeof(double) * (size_t)ng);
  free(gg);
}

/* ------------------------------------------------------------------------------------ */
/* bounds (optistack_internal.cpp:742-856 canonical forms)                 */
/* ------------------------------------------------------------------------------------ */
void lo_bounds(const lo_form* F, const double* p, double* lbg, double* ubg) {
  const int N = F->N; lo_poff o; int k, i, l;
  const double inf = INFINITY;
  lo_param_offsets_form(F, &o);
  for (i = 0; i < 6; i++) {
    lbg[i] = ubg[i] = p[o.q_init + i]; lbg[6 + i] = ubg[6 + i] = p[o.qd_init + i];
    lbg[12 + i] = p[o.q_term_min + i]; ubg[12 + i] = inf;
    lbg[18 + i] = -inf; ubg[18 + i] = p[o.q_term_max + i];
    lbg[24 + i] = p[o.qd_term_min + i]; ubg[24 + i] = inf;
    lbg[30 + i] = -inf; ubg[30 + i] = p[o.qd_term_max + i];
  }
  for (k = 0; k < N; k++) {
    const int last = (k == N - 1); const rowmap rm = stage_rowmap(last);
    double* lb = lbg + 36 + 104 * k; double* ub = ubg + 36 + 104 * k;
    for (i = 0; i < 12; i++) lb[i] = ub[i] = 0.0;
    for (l = 0; l < 4; l++) {
      const int rb = rm.leg0 + rm.leg_stride * l, rk = rb + rm.kin;
      lb[12 + l] = 0.0; ub[12 + l] = p[o.f_max];
      lb[rb] = 0.0; ub[rb] = inf;
      lb[rb + 1] = -inf; ub[rb + 1] = F->comp_eps;
      if (rm.slip) for (i = 0; i < 3; i++) { lb[rb + 2 + i] = -inf; ub[rb + 2 + i] = F->slip_eps; lb[rb + 5 + i] = -F->slip_eps; ub[rb + 5 + i] = inf; }
      lb[rk] = -F->kin_box[0]; ub[rk] = F->kin_box[0];
      lb[rk + 1] = -F->kin_box[1]; ub[rk + 1] = F->kin_box[1];
      lb[rk + 2] = -F->kin_box[2]; ub[rk + 2] = 0.0;
      lb[rk + 3] = -inf; ub[rk + 3] = p[o.l_leg_max] * p[o.l_leg_max];
    }
    for (i = 0; i < 16; i++) { lb[rm.fric + i] = -inf; ub[rm.fric + i] = 0.0; }
    for (i = 0; i < 6; i++) {
      lb[rm.box + i] = -inf; ub[rm.box + i] = p[o.q_max + i];
      lb[rm.box + 6 + i] = p[o.q_min + i]; ub[rm.box + 6 + i] = inf;
      lb[rm.box + 12 + i] = -inf; ub[rm.box + 12 + i] = p[o.qd_max + i];
      lb[rm.box + 18 + i] = p[o.qd_min + i]; ub[rm.box + 18 + i] = inf;
    }
  }
}

/* KKT residual, SURVEY 8(d): pr_inf = max_viol(g), du_inf = ||grad f + J^T lam||_inf,
 * compl = max |lam_i * min(g_i-lb_i, ub_i-g_i)| with sign convention lam>0 <-> upper bound
 * (for one-sided rows the distance to the only finite bound is used; equality rows give 0). */
void lo_kkt(const lo_form* F, const double* x, const double* p, const double* lam_g, double out[3]) {
  const int N = F->N; const lo_int nx = lo_nx(N), ng = lo_ng(N); lo_int i;
  double* g = (double*)malloc(sizeof(double) * (size_t)ng);
  double* lb = (double*)malloc(sizeof(double) * (size_t)ng);
  double* ub = (double*)malloc(sizeof(double) * (size_t)ng);
  double* gx = (double*)malloc(sizeof(double) * (size_t)nx);
  double pr = 0, du = 0, co = 0;
  lo_nlp_grad(F, x, p, 1.0, lam_g, NULL, g, gx, NULL);
  lo_bounds(F, p, lb, ub);
  for (i = 0; i < nx; i++) if (fabs(gx[i]) > du) du = fabs(gx[i]);
  for (i = 0; i < ng; i++) {
    double v = 0, dist;
    if (g[i] < lb[i]) v = lb[i] - g[i];
    if (g[i] > ub[i]) v = g[i] - ub[i];
    if (v > pr) pr = v;
    if (lb[i] == ub[i]) continue;
    if (lam_g[i] > 0) dist = ub[i] - g[i]; else dist = g[i] - lb[i];
    if (isinf(dist)) dist = (lam_g[i] == 0.0) ? 0.0 : INFINITY;  /* multiplier on an absent bound */
    if (fabs(lam_g[i] * dist) > co) co = fabs(lam_g[i] * dist);
  }
  out[0] = pr; out[1] = du; out[2] = co;
  free(g); free(lb); free(ub); free(gx);
}
