/*
 * landing_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, fp64, scalar) of the SRBM quadruped-landing NLP of
 * se-hwan/landing-controller, written from the MATLAB/Opti formulation
 *   optimizations/landing/generate_solver/generate_landingCtrller_IPOPT.m:41-170
 *   (+ generate_quadruped_SRBM_CCC.m:43-190 for the N=41 kin-box / running cost),
 *   utilities_general/dynamics-utilities/{rpyToRotMat.m:2,Binv.m:13-17},
 *   utilities_general/spatial_v2/3D/{rx.m:8-13,ry.m:8-13,rz.m:8-13},
 * N-generic, with analytic first and second derivatives.
 *
 * Pinning: at N=20 every output (f, g, grad_f, jac_g nz, hess_l nz, grad_gamma_x/p and
 * the two CCS sparsity patterns) is compared against the reference's own CasADi-generated
 * C (optimizations/landing/codegen_casadi/landingCtrller_IPOPT.c, compiled by
 * oracle/Makefile into oracle/_ref/liblanding_ref.so) in tests/test_oracle_vs_ref.py and
 * against committed fixtures generated from it (tests/golden/, tests/make_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this code.
 */
#ifndef LANDING_ORACLE_H
#define LANDING_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef long long lo_int;

/* Formulation knobs that differ between the reference's scripts (not part of p). */
typedef struct {
  int N;              /* number of intervals (reference "N" minus one)                     */
  double kin_box[3];  /* .15,.15,.30 generate_landingCtrller_IPOPT.m:149-151 ; CCC: .05,.05,.27 */
  double kin_z_off;   /* 0.05, generate_landingCtrller_IPOPT.m:155                          */
  double comp_eps;    /* 1e-3, :140 */
  double slip_eps;    /* 1e-2, :143-144 */
  /* running cost of the N=41 script (generate_quadruped_SRBM_CCC.m:81-89), off by default:
   *   sum_k dt_k ( |X_k - Xref_k|^2_QX + |pos_k + p_hip - c_k|^2_Qc (per leg) + |f_k - f_ref|^2_Qf (per leg) )
   * run_cost 1: QX, Qc, Qf, f_ref are the constants below and p is the IPOPT variant's; run_cost 2: the script's OWN parameter vector
   * (lo_param_offsets_form: Uref, QX, Qc, Qf are entries of p, grad_gamma_p has entries for them; the fields below are ignored). */
  int run_cost;
  double QX[12], Qc[3], Qf[3];
  double f_ref[3];    /* Uref(13:24,k) = f_ref per leg in the callers (test_loadCasadi_ws.m:68-72) */
  double p_hip[12];   /* CCC :76-79 */
} lo_form;

/* running cost of stage k; gX/gc/gf (12 each, may be NULL) receive its gradient (added to) */
double lo_run_cost_stage(const lo_form* F, const double* x, const double* p, int k, double* gX, double* gc, double* gf);

void lo_form_default(lo_form* F, int N);

/* sizes: nx=36N+12, ng=104N+12, np=13N+94 */
lo_int lo_nx(int N);
lo_int lo_ng(int N);
lo_int lo_np(int N);
lo_int lo_nnz_jac(int N);
lo_int lo_nnz_hess(int N);

/* offsets into p (generate_landingCtrller_IPOPT.m:51-75; Uref is inactive and dropped) */
typedef struct {
  int Xref, dt, q_min, q_max, qd_min, qd_max, q_init, qd_init, q_term_min, q_term_max,
      qd_term_min, qd_term_max, QN, mu, l_leg_max, f_max, mass, Ib, Ib_inv, np;
  int Uref, QX, Qc, Qf;      /* only with lo_form.run_cost == 2 (the N=41 script's own parameter vector); -1 otherwise */
} lo_poff;
void lo_param_offsets(int N, lo_poff* o);
void lo_param_offsets_form(const lo_form* F, lo_poff* o);
lo_int lo_np_form(const lo_form* F);
double lo_rc_weight(const lo_form* F, const double* p, int which, int i);   /* QX / Qc / Qf of the running cost: form constants or entries of p */      /* 13N+94, or 37N+112 with run_cost == 2 */

/* CCS patterns: colind[nx+1], row[nnz] (casadi mem.h:73-91 without the 2-int header) */
void lo_pattern_jac(int N, lo_int* colind, lo_int* row);
void lo_pattern_hess(int N, lo_int* colind, lo_int* row);

/* NLP callbacks (reference: landingCtrller_IPOPT.c nlp_f:10995, nlp_g:11161,
 * nlp_grad_f:52602, nlp_jac_g:94014, nlp_hess_l:53527, nlp_grad:22015). Outputs may be NULL. */
void lo_nlp_f(const lo_form* F, const double* x, const double* p, double* f);
void lo_nlp_grad_f(const lo_form* F, const double* x, const double* p, double* f, double* grad);
void lo_nlp_g(const lo_form* F, const double* x, const double* p, double* g);
void lo_nlp_jac_g(const lo_form* F, const double* x, const double* p, double* g, double* jac_nz);
void lo_nlp_hess_l(const lo_form* F, const double* x, const double* p, double lam_f,
                   const double* lam_g, double* hess_nz);
/* Lagrangian Hessian including the running cost of the N=41 script, in this project's extended pattern (casadi_s4 + 18 N diagonals) */
lo_int lo_nnz_hess_rc(int N);
void lo_pattern_hess_rc(int N, lo_int* colind, lo_int* row);
void lo_nlp_hess_l_rc(const lo_form* F, const double* x, const double* p, double lam_f, const double* lam_g, double* hess_nz);
void lo_nlp_grad(const lo_form* F, const double* x, const double* p, double lam_f,
                 const double* lam_g, double* f, double* g, double* grad_x, double* grad_p);

/* lbg/ubg from p, Opti canonicalisation (optistack_internal.cpp:742-856).
 * +-inf are returned as +-INFINITY. */
void lo_bounds(const lo_form* F, const double* p, double* lbg, double* ubg);

/* Reference-consistent KKT residual: out[0]=pr_inf, out[1]=du_inf, out[2]=compl */
void lo_kkt(const lo_form* F, const double* x, const double* p, const double* lam_g,
            double out[3]);

/* ---- stage-level API (used by the CPU solver port, oracle/landing_solver_cpu.c) ---- */
#define LO_NLOC 60   /* X_k(12) U_k(24) X_{k+1}(12) c_{k+1}(12) */
#define LO_NROW 104
/* residual rows of stage k (104; last stage 80), dense Jacobian J[row*60+loc] and, if lam
 * (multipliers of the stage's rows) is given, dense symmetric Hessian H[60*60] of lam^T g_k. */
void lo_stage_eval(const lo_form* F, int k, const double* x, const double* p,
                   const double* lam, double* g, double* J, double* H);
int lo_stage_rows(const lo_form* F, int k);

#ifdef __cplusplus
}
#endif
#endif
